// blosclz_kernel.h -- the BloscLZ stream codec (codec format 0) for gfx950, one wave per stream.
//
// The reference selects it with enums::codec::blosclz (compressed/enums.h:18-24 -> BLOSC_BLOSCLZ,
// blosc2/wrapper.h:74-119); c-blosc2 then calls blosclz_compress / blosclz_decompress per stream exactly where
// it calls LZ4 for enums::codec::lz4.  Bytes are those of BloscLZ 2.3.0 (oracle/blosclz.c states what that pin
// means); the decoder is format-defined.
//
// Encode.  blosclz_compress is a sequential scan: EVERY position reads its hash slot, writes its own position
// and tests the candidate (no skip acceleration), a 4-byte hit is only taken when the match is long enough, and
// two more slots are written at the end of a match.  A wave runs it as windows of 64 consecutive positions with
// the same read / write / read-back collision rule as the LZ4 windows (encode_kernel.h): the window is committed
// up to B = min(first accepted match, first head that shares a slot with another head of the window), lanes
// above B put their old slot value back.  "Accepted" needs the match length: every lane with a 4-byte hit
// compares the next 12 bytes in the same round trip, which decides every minimum-length rule of the codec.
// The encoder's entropy probe (get_csize: a dry run over the first n / 8 bytes, twice at level 9 to choose the
// "ipshift") is the same scan with a counting sink, so one loop serves the two or three passes of a stream.
// Found sequences are parked in lanes and written 64 at a time.  Every budget check of blosclz_compress is
// "op + k <= limit" with a left-hand side that never decreases, so the call succeeds iff size + 1 <= maxout
// (and maxout >= 66): `need` = max(66, size + 1) and nothing has to be checked per token.
//
// Decode.  Compressed bytes are parked at the end of the stream's LDS region and decoded in place; a literal
// run (<= 32 bytes) or a match header always sits inside one 64-byte register window, so a token costs one LDS
// round trip plus its copy.
#pragma once
#include "codec_types.h"
#include "wave.h"
#include "decode_kernel.h"

namespace cimg {

enum : int {
    BLZ_MAX_COPY = 32, BLZ_MAX_DISTANCE = 8191, BLZ_HASH_LOG = 12, BLZ_TAB_BYTES = (1 << BLZ_HASH_LOG) * 2,
    BLZ_MAX_INPUT = 65535,          // positions live in a uint16 table
};

// (blz_inplace_margin / blz_region_stride, the room for in-place decode, live in decode_kernel.h)
inline int blz_encode_lds_bytes(int stream_bytes) { return round16(stream_bytes) + 16 + BLZ_TAB_BYTES; }

CIMG_DEV uint32_t blz_hash(uint32_t v, int shift) { return (v * 2654435761u) >> shift; }
CIMG_DEV int blz_div255(int x) { return (int)(((uint64_t)(uint32_t)x * 0x80808081ull) >> 39); }

// number of equal bytes in[a + k] == in[b + k], k = 0 .. maxc - 1 (256 bytes per LDS round trip).  Loads are
// unguarded: the hash table follows the plane in LDS and lanes past maxc are cut by the min.
CIMG_DEV int blz_count(const uint8_t* in, int a, int b, int maxc, int n)
{
    int cnt = 0;
    for (int it = 0; it <= n / 256 + 1; ++it) {
        LV<int> len;
        LV<bool> stop;
        FOR_LANES(l) {
            const int k = cnt + 4 * l;
            const uint32_t x = lds_ld32u(in, a + k) ^ lds_ld32u(in, b + k);
            const int ln = imin(x ? (int)(__builtin_ctz(x) >> 3) : 4, imax(maxc - k, 0));
            len[l] = ln;
            stop[l] = ln < 4;
        }
        const uint64_t sm = ballot(stop);
        if (sm) { const int f = ctz64(sm); return cnt + 4 * f + readlane(len, f); }
        cnt += 256;
    }
    return imin(cnt, maxc);
}

// ---- stream layout helpers -----------------------------------------------------------------------------------------
// L literals occupy L + ceil(L / 32) bytes: a control byte (count - 1) in front of every group of <= 32
CIMG_DEV int blz_lit_bytes(int L) { return L ? L + ((L + 31) >> 5) : 0; }

// whole-wave copy of L literals in[from ..) to out[pos ..) in the control-byte layout
CIMG_DEV void blz_emit_literals(const uint8_t* in, int from, cimg_global_u8p out, int pos, int L)
{
    const int groups = (L + 31) >> 5;
    for (int g0 = 0; g0 < groups; g0 += 64) {
        FOR_LANES(l) {
            const int g = g0 + l;
            if (g < groups) {
                const int cnt = imin(32, L - 32 * g);
                out[pos + 33 * g] = (uint8_t)((cnt - 1) | ((pos + 33 * g) == 0 ? 0x20 : 0));
            }
        }
    }
#ifndef CIMG_EMULATE
#pragma unroll 1
#endif
    for (int c = 0; c < L; c += 64) {
        FOR_LANES(l) { const int j = c + l; if (j < L) out[pos + 1 + j + (j >> 5)] = in[from + j]; }
    }
}

// Writes the np parked sequences (lane k = k-th sequence: L literals from anchor, then a match of `len` at biased
// distance d) at out[op ..) and advances op.  false: the stream cannot fit cap any more.
CIMG_DEV bool blz_emit_pending(const uint8_t* in, cimg_global_u8p out, int cap, int& op, int np,
                               const LV<int>& P_anchor, const LV<int>& P_lit, const LV<int>& P_dist, const LV<int>& P_len)
{
    LV<int> size, start, lsz, ext;
    FOR_LANES(l) {
        const int L = P_lit[l], len = P_len[l];
        ext[l] = len >= 7 ? blz_div255(len - 7) + 1 : 0;
        lsz[l] = blz_lit_bytes(L);
        size[l] = l < np ? lsz[l] + (P_dist[l] >= BLZ_MAX_DISTANCE ? 4 : 2) + ext[l] : 0;
    }
    int total;
    wave_exscan(size, start, total);
    // at least one more literal and its control byte follow, and the encoder wants size + 1 <= cap
    if (op + total + 3 > cap) return false;
    LV<bool> shortlit, longlit, longm;
    FOR_LANES(l) {
        const bool act = l < np;
        const int L = P_lit[l];
        shortlit[l] = act && L > 0 && L <= 8;
        longlit[l] = act && L > 8;
        longm[l] = act && ext[l] > 1;
    }
    // match bytes
    FOR_LANES(l) {
        if (l < np) {
            const int len = P_len[l], d = P_dist[l];
            const bool far = d >= BLZ_MAX_DISTANCE;
            const int dd = far ? d - BLZ_MAX_DISTANCE : d;
            int q = op + start[l] + lsz[l];
            out[q++] = (uint8_t)(((len < 7 ? len : 7) << 5) + (far ? 31 : (dd >> 8)));
            if (len >= 7) {
                const int rem = len - 7, n255 = ext[l] - 1;
                out[q + n255] = (uint8_t)(rem - 255 * n255);
                q += n255 + 1;
            }
            if (far) { out[q] = 255; out[q + 1] = (uint8_t)(dd >> 8); out[q + 2] = (uint8_t)(dd & 255); }
            else out[q] = (uint8_t)(dd & 255);
        }
    }
    if (ballot(longm)) {
        // 255-bytes of long matches: one lane per sequence and one byte per round for the sequences with at most four of
        // them; a longer run (a match of 1282 bytes and more: 16 for a 4 KiB row of an image) would set the round count for
        // everybody and is written 64 bytes at a time by the whole wave instead (encode_kernel.h: emit_pending has the numbers)
        for (int r = 0; r < 4; ++r) {
            LV<bool> more;
            FOR_LANES(l) {
                const int n255 = ext[l] - 1 <= 4 ? ext[l] - 1 : 0;
                const bool w = longm[l] && r < n255;
                if (w) out[op + start[l] + lsz[l] + 1 + r] = 255;
                more[l] = longm[l] && r + 1 < n255;
            }
            if (!ballot(more)) break;
        }
        LV<bool> big;
        FOR_LANES(l) { big[l] = longm[l] && ext[l] - 1 > 4; }
        for (uint64_t m = ballot(big); m; m &= m - 1) {
            const int k = ctz64(m);
            const int at = op + readlane(start, k) + readlane(lsz, k) + 1, n255 = readlane(ext, k) - 1;
            for (int c = 0; c < n255; c += 64) {
                FOR_LANES_W(l) { if (c + l < n255) out[at + c + l] = 255; }
            }
        }
    }
    if (ballot(shortlit)) {
        FOR_LANES(l) { if (shortlit[l]) out[op + start[l]] = (uint8_t)((P_lit[l] - 1) | ((op + start[l]) == 0 ? 0x20 : 0)); }
        for (int r = 0; r < 8; ++r) {
            LV<bool> more;
            FOR_LANES(l) {
                if (shortlit[l] && r < P_lit[l]) out[op + start[l] + 1 + r] = in[P_anchor[l] + r];
                more[l] = shortlit[l] && r + 1 < P_lit[l];
            }
            if (!ballot(more)) break;
        }
    }
    uint64_t lm = ballot(longlit);
    while (lm) {
        const int k = ctz64(lm);
        lm &= lm - 1;
        blz_emit_literals(in, readlane(P_anchor, k), out, op + readlane(start, k), readlane(P_lit, k));
    }
    op += total;
    return true;
}

// Bit-exact blosclz_compress(clevel, in, n, out, cap) by one wave.  in: LDS plane (16 readable bytes past the end),
// tab: BLZ_TAB_BYTES of LDS.  Returns bytes written, 0 if the codec gives up / the result does not fit, < 0 if a loop
// guard tripped.  need_out = smallest cap that still succeeds.
CIMG_DEV int blosclz_encode_body(const uint8_t* in, uint8_t* tab, int n, uint8_t* out_generic, int cap, int clevel, int& need_out)
{
    cimg_global_u8p out = CIMG_AS_GLOBAL(out_generic);
    cimg_lds_vu16p tab16 = CIMG_AS_LDS_VU16(tab);
    need_out = 0;
    if (n < 16 || cap < 66 || clevel < 1 || clevel > 9) return 0;
    const int hashlog = clevel == 1 ? BLZ_HASH_LOG - 2 : (clevel == 2 ? BLZ_HASH_LOG - 1 : BLZ_HASH_LOG);
    const int minlen_tab = clevel <= 2 ? 12 : 14 - clevel;           // {-, 12, 12, 11, 10, 9, 8, 7, 6, (5)}
    const int maxlen = n >> 3;
    // passes: level 9 probes with ipshift 3 and 4, the other levels with 4 only; then the real scan
    int pass = clevel == 9 ? 0 : 1;
    int csize3 = 0, csize4 = 0, ipshift = 4;

    LV<int> P_anchor, P_lit, P_dist, P_len;
    FOR_LANES(l) { P_anchor[l] = 0; P_lit[l] = 0; P_dist[l] = 0; P_len[l] = 0; }
    int np = 0, op = 0;
    int anchor = 0;                                                   // start of the literals not yet accounted for

    // ONE way out of both loops (`break` with `ending` set): a return inside a loop makes the compiler dispatch every
    // iteration on an exit selector (encode_kernel.h has the measurement)
    int ending = 1;                                                   // 1: go on; 0: give up (stored raw); < 0: a loop guard tripped
    bool saw_hit = false;                                             // a probe met four equal bytes somewhere (a candidate that compared equal)
    for (; pass < 3; ++pass) {
        const bool probe = pass < 2;
        if (pass == 2) {
            // entropy probe verdict (blosclz_compress: "discard probes with small compression ratios")
            const int cs = clevel == 9 ? imin(csize3, csize4) : csize4;
            ipshift = (clevel == 9 && csize3 < csize4) ? 3 : 4;
            const double cratio = (double)maxlen / (double)cs;
            const double thr = clevel <= 4 ? 2.0 : (clevel == 5 ? 1.8 : (clevel == 6 ? 1.6 : (clevel == 7 ? 1.4 : (clevel == 8 ? 1.2 : 1.1))));
            if (cratio < thr) { ending = 0; break; }
        }
        const int shift = probe ? 32 - BLZ_HASH_LOG : 32 - hashlog;
        const int ips = probe ? (pass == 0 ? 3 : 4) : ipshift;
        const int minlen = probe ? 3 : (clevel == 9 ? ipshift : minlen_tab);
        const int nlim = probe ? maxlen : n;
        const int ip_bound = nlim - 1, ip_limit = nlim - 12;
        {
            const u128 z = {0, 0, 0, 0};
            for (int u0 = 0; u0 < BLZ_TAB_BYTES / 16; u0 += 64) { FOR_LANES(l) { st128a(tab + 16 * (u0 + l), z); } }
        }
        int ip = probe ? 0 : 4;
        anchor = 0;
        int oc = 5, copyc = 4;                                        // the probe's counters ("4 literals already copied")
        int guard = 0;
        while (ip < ip_limit) {
            if (++guard > n + 2) { ending = -1; break; }              // a window consumes at least one position
            const int nv = imin(64, ip_limit - ip);
            LV<int> pos;
            LV<bool> valid;
            LV<uint32_t> v, h;
            FOR_LANES(l) {
                pos[l] = ip + l;
                valid[l] = l < nv;
                v[l] = lds_ld32u(in, l < nv ? ip + l : 0);
                h[l] = blz_hash(v[l], shift);
            }
            LV<uint32_t> ph, pv;
            lane_prev(h, ph);
            lane_prev(v, pv);
            LV<bool> head, cont;
            LV<uint32_t> old;
            FOR_LANES(l) {
                cont[l] = valid[l] && l > 0 && h[l] == ph[l];
                head[l] = valid[l] && !cont[l];
                old[l] = tab16[h[l]];
            }
            FOR_LANES_W(l) { if (head[l]) tab16[h[l]] = (uint16_t)pos[l]; }
            LV<bool> loser, hit;
            LV<int> cand;
            FOR_LANES(l) {
                const uint32_t rb = tab16[h[l]];
                const uint32_t mvh = lds_ld32u(in, (int)old[l]);
                const uint32_t mv = head[l] ? mvh : pv[l];
                loser[l] = head[l] && rb != (uint32_t)pos[l];
                cand[l] = head[l] ? (int)old[l] : pos[l] - 1;
                hit[l] = valid[l] && mv == v[l] && cand[l] != pos[l];      // distance 0: position 0 against the empty table
            }
            uint64_t involved = ballot(loser);
            if (involved) {
                FOR_LANES_W(l) { if (loser[l]) tab16[h[l]] = (uint16_t)pos[l]; }
                LV<bool> inv;
                FOR_LANES(l) { inv[l] = head[l] && (loser[l] || tab16[h[l]] != (uint16_t)pos[l]); }
                involved = ballot(inv);
            }
            const int k1 = ctz64(involved);
            const int limit = imin(nv - 1, k1);
            const uint64_t below = limit >= 63 ? ~0ull : ((1ull << (limit + 1)) - 1);
            const uint64_t hits = ballot(hit) & below;
            int m = -1;
            if (hits) {
                saw_hit = true;
                // a 4-byte hit is a match only if it is long enough: the next 12 bytes decide every rule
                LV<bool> acc;
                FOR_LANES(l) {
                    const int a = hit[l] ? pos[l] + 4 : 0, b = hit[l] ? cand[l] + 4 : 0;
                    const uint32_t x0 = lds_ld32u(in, a) ^ lds_ld32u(in, b);
                    const uint32_t x1 = lds_ld32u(in, a + 4) ^ lds_ld32u(in, b + 4);
                    const uint32_t x2 = lds_ld32u(in, a + 8) ^ lds_ld32u(in, b + 8);
                    const int kk = x0 ? (int)(__builtin_ctz(x0) >> 3) : (x1 ? 4 + (int)(__builtin_ctz(x1) >> 3) : (x2 ? 8 + (int)(__builtin_ctz(x2) >> 3) : 12));
                    const int room = ip_bound - (pos[l] + 4);
                    const int lenlb = 4 + imin(kk + 1, room) - ips;            // exact below 13, a lower bound from there
                    const bool far = pos[l] - cand[l] - 1 >= BLZ_MAX_DISTANCE;
                    const bool ok = probe ? lenlb >= (far ? 4 : 3) : (lenlb >= minlen && !(lenlb <= 5 && far));
                    acc[l] = hit[l] && ok;
                }
                const uint64_t am = ballot(acc) & below;
                if (am) m = ctz64(am);
            }
            const int B = m >= 0 ? m : limit;
            const uint64_t headmask = ballot(head), contmask = ballot(cont);
            const uint64_t above = B >= 63 ? 0ull : (~0ull << (B + 1));
            if (headmask & above) {
                FOR_LANES_W(l) { if (head[l] && l > B) tab16[h[l]] = (uint16_t)old[l]; }
                FOR_LANES_W(l) { if (head[l] && l == B) tab16[h[l]] = (uint16_t)pos[l]; }
            }
            if (contmask & ~above) {
                FOR_LANES_W(l) {
                    if (cont[l] && l <= B && (l == B || !((contmask >> ((l + 1) & 63)) & 1) || l == 63)) tab16[h[l]] = (uint16_t)pos[l];
                }
            }
            if (m < 0) { ip += B + 1; continue; }
            // ---- a match at lane m ------------------------------------------------------------------------------
            const int mpos = readlane(pos, m), mref = readlane(cand, m);
            const int room = ip_bound - (mpos + 4);
            const int k = blz_count(in, mpos + 4, mref + 4, room, n);
            const int nip = mpos + 4 + (k < room ? k + 1 : room) - ips;
            const int len = nip - mpos, dist = mpos - mref - 1, L = mpos - anchor;
            if (probe) {
                const int t = copyc + L;
                oc += L + (t >> 5);
                if ((t & 31) == 0) oc--;
                copyc = 0;
                oc += (len >= 7 ? blz_div255(len - 7) + 1 : 0) + (dist >= BLZ_MAX_DISTANCE ? 4 : 2) + 1;
            } else {
                const int slot = np;
                FOR_LANES(l) { if (l == slot) { P_anchor[l] = anchor; P_lit[l] = L; P_dist[l] = dist; P_len[l] = len; } }
                if (++np == 64) {
                    if (!blz_emit_pending(in, out, cap, op, np, P_anchor, P_lit, P_dist, P_len)) { ending = 0; break; }
                    np = 0;
                }
            }
            {   // "update the hash at match boundary": positions nip and nip + 1 (the second from three bytes)
                LV<uint32_t> s2;
                FOR_LANES(l) { s2[l] = lds_ld32u(in, nip); }
                FOR_LANES_W(l) { tab16[blz_hash(s2[l], shift)] = (uint16_t)nip; }
                FOR_LANES_W(l) { tab16[blz_hash(s2[l] >> 8, shift)] = (uint16_t)(nip + 1); }
            }
            ip = nip + 2;
            anchor = ip;
        }
        if (ending <= 0) break;
        if (probe) {
            // the dry run has no left-over loop: it stops at ip_limit, the literals it counted are those below it
            const int Lc = imax(ip_limit - anchor, 0);
            const int t = copyc + Lc;
            oc += Lc + (t >> 5);
            if ((t & 31) == 0) oc--;
            if (pass == 0) csize3 = oc; else csize4 = oc;
            // The second probe of level 9 differs from the first only in what it does WITH a hit (the position it goes on from, the
            // length rule).  A first probe that never met four equal bytes -- the noisy low planes of an image -- is the second
            // one too: same table, same positions, same count.
            if (pass == 0 && !saw_hit) { csize4 = csize3; pass = 1; }
        }
    }
    if (ending <= 0) return ending;
    // ---- real pass: flush the parked sequences, then the left-over literals (at least one: the scan stops 12 bytes early)
    if (np && !blz_emit_pending(in, out, cap, op, np, P_anchor, P_lit, P_dist, P_len)) return 0;
    const int Lf = n - anchor;
    const int size = op + blz_lit_bytes(Lf);
    if (size + 1 > cap) return 0;
    blz_emit_literals(in, anchor, out, op, Lf);
    need_out = imax(66, size + 1);
    return size;
}

// ---- decode -------------------------------------------------------------------------------------------------------
// blosclz_decompress(in, csize, out, n) by one wave, in place inside LDS: compressed bytes occupy [cs, cs + csize),
// output goes to [base, base + n).  Returns 0 or ERR_DATA (c-blosc2 treats any result != n as an error).  Every LDS
// index is clamped to lds_limit, so a damaged stream gives garbage + an error, never an out-of-range access.
// Checks follow the library one to one (oracle/blosclz.c), including "a far match that ends the input is not copied".
// Round 4: tokens are taken a 64-byte window at a time, as the LZ4 decoder takes its sequences (decode_kernel.h, "batch path").
// Until then every token was a round trip of its own -- a thousand cycles each, and a byte plane of an image is a hundred to a
// thousand tokens (a control byte below 32: a run of up to 32 literals; from 32 on: a match -- three bits of length, possibly
// length bytes, an offset byte, possibly two more for a far offset).  Every lane assumes a token starts at its byte and parses
// it from five bytes (ONE round trip: all five addresses are known up front); a scalar walk follows the real chain; a prefix sum
// places the tokens' output; the literal runs of the whole window go out in one store (they ARE the window's bytes); matches that
// read nothing this window writes are copied sixteen at a time, four lanes each, the rest in order.  What a window cannot take
// -- the first token of a stream (its control byte is a literal count whatever its top bits say), a length that needs more
// than one length byte, a literal run that leaves the window, the last tokens of the stream -- goes through the token-by-token
// loop below, which is the round-2 decoder.  Same checks, same results on damaged streams (tests/emu runs both forms).
CIMG_DEV int blosclz_decode_wave(uint8_t* lds, int base, int n, int cs, int csize, int lds_limit)
{
    if (csize <= 0) return ERR_DATA;
    const int iend = cs + csize, oend = base + n;
    const int clampmax = lds_limit - 1;
    int ip = cs, op = base;                                // ip: at the control byte of the next token
    LV<uint32_t> w;                                        // token-by-token loop: lane l holds input byte wbase + l
    int wbase = -4096;
    bool first = true, bad = false;
    for (int guard = 0; guard <= csize + 1; ++guard) {
        if (ip >= iend) break;
#ifndef CIMG_BLZ_NO_BATCH
        if (!first && iend - ip >= 8) {
            // ---- batch: every token of the next 64 input bytes that the window can take ------------------------------------------
            LV<uint32_t> c0;
            LV<int> lit_l, walk_l, len_l, off_l, ml_l;
            LV<bool> good;
            enum : int { WALK_BAD = 0x2000 };
            FOR_LANES(l) {
                const int at = ip + l;
                c0[l] = lds[imin(at, clampmax)];
                const uint32_t b1 = lds[imin(at + 1, clampmax)], b2 = lds[imin(at + 2, clampmax)];
                const uint32_t b3 = lds[imin(at + 3, clampmax)], b4 = lds[imin(at + 4, clampmax)];
                const uint32_t c = c0[l];
                const bool is_lit = c < 32;
                const int lf = (int)(c >> 5);
                const bool ext = lf == 7;
                const uint32_t code = ext ? b2 : b1;
                const bool far = (code == 255) & ((c & 31u) == 31u);
                const uint32_t f1 = ext ? b3 : b2, f2 = ext ? b4 : b3;
                const int dist = far ? BLZ_MAX_DISTANCE + (int)((f1 << 8) + f2) + 1 : (int)((c & 31u) << 8) + (int)code + 1;
                const int hdr = 2 + (ext ? 1 : 0) + (far ? 2 : 0);
                lit_l[l] = is_lit ? (int)c + 1 : 0;
                ml_l[l] = is_lit ? 0 : lf + 2 + (ext ? (int)b1 : 0);
                off_l[l] = is_lit ? 0 : dist;
                const int nxt = l + (is_lit ? 1 + lit_l[l] : hdr);
                len_l[l] = lit_l[l] + ml_l[l];
                // (one length byte only; a following token exists: the last tokens of a stream go the slow way, with its checks)
                good[l] = (is_lit | !ext | (b1 < 255)) & (ip + nxt < iend);
                // a literal run that leaves the window is copied LDS -> LDS, for the LAST token of a batch only (flag + 1024)
                walk_l[l] = good[l] ? nxt + ((is_lit & (nxt > 64)) ? 1024 : 0) : (int)WALK_BAD;
            }
            uint64_t tokens = 0;
            int s = 0, t_ = 0;
#define CIMG_WALK_STEP t_ = readlane(walk_l, s); if (t_ > 63) break; tokens |= 1ull << s; s = t_;
            for (int rnd = 0; rnd < 4; ++rnd) {
                CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP
            }
#undef CIMG_WALK_STEP
            if (t_ > 63 && !(t_ & WALK_BAD)) { tokens |= 1ull << s; s = t_; }
            int biglast = 0;
            if (s >= 1024) { s -= 1024; biglast = 1; }
            LV<int> tlen, opos;
            LV<bool> istok;
            FOR_LANES(l) { istok[l] = (tokens >> l) & 1; tlen[l] = istok[l] ? len_l[l] : 0; }
            int acc;
            wave_exscan(tlen, opos, acc);
            if (acc > oend - op) {                                 // cut the batch at the first token that does not fit
                const int room = oend - op;
                LV<bool> over;
                FOR_LANES(l) { over[l] = istok[l] & (opos[l] + tlen[l] > room); }
                const int f = ctz64(ballot(over));
                tokens &= (1ull << f) - 1;
                s = f;
                biglast = 0;
                acc = readlane(opos, f);
                FOR_LANES(l) { istok[l] = (tokens >> l) & 1; }
            }
            if (tokens) {
                // literals: lane j belongs to the last token at or before j - 1 (a run's bytes follow its control byte)
                const int tlast = 63 - (int)__builtin_clzll(tokens);
                const int litlim = biglast ? tlast : imin(s, 64);
                LV<int> owner;
                LV<bool> is_lit;
                FOR_LANES(l) {
                    const uint64_t below = tokens & ((1ull << l) - 1);
                    owner[l] = below ? 63 - (int)__builtin_clzll(below) : 0;
                    is_lit[l] = below != 0;
                }
                LV<int> own_lit, own_pos;
                lane_gather(lit_l, owner, own_lit);
                lane_gather(opos, owner, own_pos);
                FOR_LANES_W(l) {
                    const int k = l - owner[l] - 1;
                    if (is_lit[l] & (k < own_lit[l]) & (l < litlim)) lds[op + own_pos[l] + k] = (uint8_t)c0[l];
                }
                if (biglast) lds_copy_bytes(lds, op + readlane(opos, tlast), ip + tlast + 1, readlane(lit_l, tlast));
                // matches
                LV<int> dstv, srcv;
                LV<bool> badv, hasm;
                FOR_LANES(l) {
                    dstv[l] = op + opos[l];                        // (a match token has no literals of its own)
                    srcv[l] = dstv[l] - off_l[l];
                    hasm[l] = istok[l] & (ml_l[l] > 0);
                    badv[l] = hasm[l] & (srcv[l] < base);
                }
                if (ballot(badv)) return ERR_DATA;
                LV<bool> par;
                FOR_LANES(l) {
                    const int off = off_l[l], ml = ml_l[l];
                    par[l] = hasm[l] & (ml <= 64) & ((off >= ml) | (off == 1)) & (srcv[l] + (off == 1 ? 1 : ml) <= op);
                }
                uint64_t parmask = ballot(par);
                if (popc64(parmask) < 3) parmask = 0;
                if (parmask) {
                    const int P = popc64(parmask);
                    LV<int> rank, d0, d1, D0, D1;
                    FOR_LANES(l) {
                        rank[l] = par[l] ? lane_rank(parmask, l) : 63;
                        d0[l] = dstv[l] | (ml_l[l] << 18);
                        d1[l] = srcv[l] | (off_l[l] == 1 ? 1 << 18 : 0);
                    }
                    lane_scatter(d0, rank, D0);
                    lane_scatter(d1, rank, D1);
                    for (int g = 0; g < P; g += 16) {
                        LV<int> who, e0, e1;
                        FOR_LANES(l) { who[l] = g + (l >> 2); }
                        lane_gather(D0, who, e0);
                        lane_gather(D1, who, e1);
                        LV<u128> wv;
                        FOR_LANES(l) {
                            const bool act = who[l] < P;
                            const bool f = (e1[l] >> 18) & 1;
                            const int sa = act ? (e1[l] & 0x3FFFF) + (f ? 0 : (l & 3) * 16) : base;
                            const int a = sa & ~3;
                            const uint32_t sh = (uint32_t)sa & 3u;
                            const uint32_t q0 = *reinterpret_cast<const uint32_t*>(lds + a);
                            const uint32_t q1 = *reinterpret_cast<const uint32_t*>(lds + a + 4);
                            const uint32_t q2 = *reinterpret_cast<const uint32_t*>(lds + a + 8);
                            const uint32_t q3 = *reinterpret_cast<const uint32_t*>(lds + a + 12);
                            const uint32_t q4 = *reinterpret_cast<const uint32_t*>(lds + a + 16);
                            const uint32_t x0 = alignbyte(q1, q0, sh);
                            const uint32_t fb = (x0 & 0xFF) * 0x01010101u;
                            wv[l].x = f ? fb : x0;
                            wv[l].y = f ? fb : alignbyte(q2, q1, sh);
                            wv[l].z = f ? fb : alignbyte(q3, q2, sh);
                            wv[l].w = f ? fb : alignbyte(q4, q3, sh);
                        }
                        FOR_LANES_W(l) {
                            const int rem = who[l] < P ? ((e0[l] >> 18) & 0x7F) - (l & 3) * 16 : 0;
                            uint8_t* d = lds + (e0[l] & 0x3FFFF) + (l & 3) * 16;
                            const uint32_t v[4] = {wv[l].x, wv[l].y, wv[l].z, wv[l].w};
                            CIMG_UNROLL
                            for (int j = 0; j < 4; j++) { if (rem >= 4 * j + 4) lds_st32u(d + 4 * j, v[j]); }
                            const int t = rem > 0 ? (rem > 16 ? 16 : rem) & ~3 : 0;
                            const uint32_t last = t < 16 ? v[(t >> 2) & 3] : 0;
                            CIMG_UNROLL
                            for (int k = 0; k < 3; k++) { if (rem > t + k && t + k < 16) d[t + k] = (uint8_t)(last >> (8 * k)); }
                        }
                    }
                }
                uint64_t todo = ballot(hasm) & ~parmask;
                while (todo) {
                    const int t = ctz64(todo);
                    todo &= todo - 1;
                    const int dst = readlane(dstv, t), src = readlane(srcv, t), ml = readlane(ml_l, t);
                    const int offset = dst - src;
                    if (ml <= 64) {
                        LV<uint32_t> mv;
                        if (offset >= ml) { FOR_LANES(l) { mv[l] = lds[src + (l < ml ? l : 0)]; } }
                        else if (offset == 1) { FOR_LANES(l) { mv[l] = lds[src]; } }
                        else {
                            const float inv = fast_rcp((float)offset);
                            FOR_LANES(l) { mv[l] = lds[src + small_mod(l < ml ? l : 0, offset, inv)]; }
                        }
                        FOR_LANES_W(l) { if (l < ml) lds[dst + l] = (uint8_t)mv[l]; }
                    } else {
                        lds_copy_match(lds, dst, src, ml);
                    }
                }
                ip += s;
                op += acc;
                wbase = -4096;                                     // (the token-by-token window is stale)
                continue;
            }
        }
#endif
        // ---- one token the slow way (the round-2 decoder): lane l holds input byte wbase + l; a literal run (<= 32 bytes) or a match
        // header (<= 5 bytes) that starts in the first 28 bytes of the window is served from these registers
        int k = ip - wbase;
        if (k > 27 || k < 0) {
            wbase = ip; k = 0;
            FOR_LANES(l) { w[l] = lds[imin(wbase + l, clampmax)]; }
        }
        uint32_t ctrl = readlane(w, k);
        if (first) { ctrl &= 31u; first = false; }
        k++;                                                   // k: the byte behind the control byte
        if (ctrl >= 32) {
            int len = (int)(ctrl >> 5) - 1;
            int ofs = (int)(ctrl & 31u) << 8;
            if (len == 6) {
                // length bytes: added up to and including the first that is not 255; each needs a byte behind it
                for (;;) {
                    LV<bool> stop;
                    FOR_LANES(l) { stop[l] = (l >= k) & (w[l] != 255); }
                    const int f = ctz64(ballot(stop));
                    if (f < 64) {
                        if (wbase + f + 1 >= iend) { bad = true; break; }
                        len += 255 * (f - k) + (int)readlane(w, f);
                        k = f + 1;
                        break;
                    }
                    if (wbase + 64 >= iend) { bad = true; break; }
                    len += 255 * (64 - k);
                    wbase += 64; k = 0;
                    FOR_LANES(l) { w[l] = lds[imin(wbase + l, clampmax)]; }
                }
                if (bad) break;
                if (k > 56) { wbase += k; k = 0; FOR_LANES(l) { w[l] = lds[imin(wbase + l, clampmax)]; } }
            } else if (wbase + k + 1 >= iend) { bad = true; break; }
            const int code = (int)readlane(w, k);
            k++;
            len += 3;
            int ref = op - ofs - code;
            if (code == 255 && ofs == (31 << 8)) {
                if (wbase + k + 1 >= iend) { bad = true; break; }
                ofs = ((int)readlane(w, k) << 8) + (int)readlane(w, k + 1);
                k += 2;
                ref = op - ofs - BLZ_MAX_DISTANCE;
            }
            if (len > oend - op || ref - 1 < base) { bad = true; break; }
            ip = wbase + k;
            ref--;
            if (len <= 64) {
                const int offset = op - ref;
                LV<uint32_t> mv;
                if (offset >= len) {
                    FOR_LANES(l) { mv[l] = lds[ref + (l < len ? l : 0)]; }
                } else if (offset == 1) {
                    FOR_LANES(l) { mv[l] = lds[ref]; }
                } else {
                    const float inv = fast_rcp((float)offset);
                    FOR_LANES(l) { mv[l] = lds[ref + small_mod(l < len ? l : 0, offset, inv)]; }
                }
                FOR_LANES_W(l) { if (l < len) lds[op + l] = (uint8_t)mv[l]; }
            } else {
                lds_copy_match(lds, op, ref, len);
            }
            op += len;
        } else {
            const int cnt = (int)ctrl + 1;                   // <= 32: the run is in the window
            ip = wbase + k;
            if (cnt > oend - op || ip + cnt > iend) { bad = true; break; }
            FOR_LANES_W(l) { if (l >= k && l < k + cnt) lds[op + l - k] = (uint8_t)w[l]; }
            op += cnt; ip += cnt;
        }
    }
    return (!bad && op == oend) ? 0 : ERR_DATA;
}

}  // namespace cimg
