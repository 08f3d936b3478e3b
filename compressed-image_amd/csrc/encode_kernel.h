// encode_kernel.h -- blosc2 chunk encode for gfx950: shuffle + run detection + LZ4 per stream.
//
// Replaces what the reference reaches through blosc2_compress_ctx (blosc2/wrapper.h:139,172, called
// per 4 MiB chunk from schunk.h:85-104).  One 256-thread workgroup per 32 KiB block of the batch:
//
//   phase A  four waves load the block with 16-byte coalesced reads and scatter it into LDS as
//            `typesize` byte planes (the byte shuffle; v_perm transposes for typesize 2 / 4).
//   barrier
//   phase B  wave s owns plane s: run check, then a *bit-exact* LZ4_compress_fast of the plane with
//            the byU16 hash table (8192 x u16) in LDS.  The match search is inherently sequential
//            (every probe reads and writes the table), so it is run as 64-probe windows:
//              - lane l takes probe l of the skip schedule (positions known in advance),
//              - all lanes read their table slot, write their position, read it back; a lane whose
//                read-back differs shares its slot with another lane of the window,
//              - the window is committed up to B = min(first matching lane, first lane involved
//                in a slot collision): lanes <= B see exactly the table a sequential scan would
//                show them, lanes > B put their old slot value back, lane B re-writes its own,
//              - a match at lane B is extended (lane-parallel compare, 256 B per step) and emitted
//                with lane-parallel literal copies; the "probe right after the match" of the
//                sequential algorithm rides as lane 0 of the next window.
//            Output goes to the block's scratch slot; a per-stream record (kind, size, need) is
//            left for the layout kernel.  `need` is the smallest output budget under which LZ4
//            still succeeds; it lets the layout kernel re-apply blosc2's running-destsize rule
//            without re-encoding (oracle/chunk.c: orc_blosc2_compress_2phase is the CPU twin).
#pragma once
#include "codec_types.h"
#include "wave.h"
#include "decode_kernel.h"   // round16, find_chunk, byte_perm, wave copies

namespace cimg {

struct EncodeArgs {
    const ChunkDesc* descs;
    int32_t nchunks;
    CodecParams p;
    const uint8_t* raw;       // pixels at raw + desc.raw_off
    uint8_t* scratch;         // block b owns scratch + b * p.slot_bytes
    StreamRec* recs;          // block b, stream s -> recs[b * p.streams_per_block + s]
    int32_t lds_bytes;
};

enum : int { LZ4_HASH_BYTES = 16384, LZ4_MAX_INPUT_U16 = 65536 + 11 - 1 };

CIMG_HD int plane_stride(int neblock) { return round16(neblock) + 16; }
inline int encode_lds_bytes(int blocksize, int typesize, bool split)
{
    const int ns = split ? typesize : 1;
    const int ne = blocksize / ns;
    // unsplit blocks still need room for the whole shuffled block
    return ns * (round16(ne) + 16) + ns * LZ4_HASH_BYTES + 32;
}

CIMG_DEV uint32_t lz4_hash(uint32_t v) { return (v * 2654435761u) >> 19; }
// sum_{x=0}^{n-1} (x >> 6)
CIMG_DEV int skip_prefix(int n) { const int q = n >> 6, r = n & 63; return 32 * q * (q - 1) + q * r; }
// offset of probe t of a search from the search start (probe 0 sits on the start)
CIMG_DEV int probe_offset(int t, int s64) { return t <= 0 ? 0 : 1 + skip_prefix(s64 + t - 1) - skip_prefix(s64); }

// write `count` bytes of an LZ4 length extension (count-1 times 255, then `last`) at out[pos..)
CIMG_DEV void emit_len_ext(uint8_t* out, int pos, int rem)
{
    const int n255 = rem / 255, last = rem - 255 * n255;
    for (int c = 0; c <= n255; c += 64) {
        FOR_LANES(l) { if (c + l <= n255) out[pos + c + l] = (uint8_t)(c + l < n255 ? 255 : last); }
    }
}

CIMG_DEV void emit_literals(const uint8_t* in, int from, uint8_t* out, int pos, int count)
{
    for (int c = 0; c < count; c += 64) {
        FOR_LANES(l) { if (c + l < count) out[pos + c + l] = in[from + c + l]; }
    }
}

// Bit-exact LZ4_compress_fast(in, out, n, cap, accel) in limited-output mode, byU16 table, by one wave.
// in: LDS plane (padded by >= 8 readable bytes), tab: 16 KiB LDS.  Returns bytes written, 0 if the
// result does not fit cap.  *need = smallest cap that still succeeds.
CIMG_DEV int lz4_encode_wave(const uint8_t* in, uint8_t* tab, int n, uint8_t* out, int cap, int accel, int& need_out)
{
    uint16_t* tab16 = reinterpret_cast<uint16_t*>(tab);
    {
        const u128 z = {0, 0, 0, 0};
        for (int u0 = 0; u0 < LZ4_HASH_BYTES / 16; u0 += 64) {
            FOR_LANES(l) { st128a(tab + 16 * (u0 + l), z); }
        }
    }
    const int mflimit_p1 = n - 11, matchlimit = n - 5;
    const int s64 = accel << 6;
    int anchor = 0, op = 0, need = 0;

    if (n >= 13) {
        {   // first byte
            LV<uint32_t> v0;
            FOR_LANES(l) { v0[l] = lds_ld32u(in, 0); }
            FOR_LANES_W(l) { if (l == 0) tab16[lz4_hash(v0[l])] = 0; }
        }
        int sstart = 1;     // search start position
        int t0 = 0;         // probes of this search already committed
        bool pre = false;   // lane 0 = the probe right after a match (position sstart - 1)
        for (;;) {
            // ---- lay the window out ------------------------------------------------------------------
            LV<int> pos;
            LV<bool> valid;
            LV<uint32_t> v, h, old, rb, mv;
            LV<uint32_t> back;           // dword at (sstart - 3): the "put(ip - 2)" refill after a match
            FOR_LANES(l) {
                if (pre && l == 0) {
                    pos[l] = sstart - 1;
                    valid[l] = true;
                } else {
                    const int t = t0 + l - (pre ? 1 : 0);
                    pos[l] = sstart + probe_offset(t, s64);
                    valid[l] = sstart + probe_offset(t + 1, s64) <= mflimit_p1;
                }
                v[l] = valid[l] ? lds_ld32u(in, pos[l]) : 0u;
                h[l] = lz4_hash(v[l]);
                back[l] = pre ? lds_ld32u(in, sstart - 3) : 0u;
            }
            const uint64_t vmask = ballot(valid);
            const int nv = popc64(vmask);                 // valid lanes are a prefix
            if (nv == 0) break;                           // -> last literals
            if (pre) {
                FOR_LANES_W(l) { if (l == 0) tab16[lz4_hash(back[l])] = (uint16_t)(sstart - 3); }
            }
            FOR_LANES(l) { old[l] = valid[l] ? tab16[h[l]] : 0u; }
            FOR_LANES_W(l) { if (valid[l]) tab16[h[l]] = (uint16_t)pos[l]; }
            LV<bool> loser, hit;
            FOR_LANES(l) {
                rb[l] = valid[l] ? tab16[h[l]] : 0u;
                mv[l] = valid[l] ? lds_ld32u(in, (int)old[l]) : 0u;
                loser[l] = valid[l] && rb[l] != (uint32_t)pos[l];
                hit[l] = valid[l] && mv[l] == v[l];
            }
            uint64_t involved = ballot(loser);
            if (involved) {
                // lanes that lost a slot write again; a winner whose slot changes has company too
                FOR_LANES_W(l) { if (loser[l]) tab16[h[l]] = (uint16_t)pos[l]; }
                LV<bool> inv;
                FOR_LANES(l) { inv[l] = valid[l] && (loser[l] || tab16[h[l]] != (uint16_t)pos[l]); }
                involved = ballot(inv);
            }
            const int k1 = ctz64(involved);                           // 64 if no collision
            const int limit = imin(nv - 1, k1);
            const uint64_t hits = ballot(hit) & (limit >= 63 ? ~0ull : ((1ull << (limit + 1)) - 1));
            const int m = hits ? ctz64(hits) : -1;
            const int B = m >= 0 ? m : limit;
            if (B < nv - 1) {
                FOR_LANES_W(l) { if (valid[l] && l > B) tab16[h[l]] = (uint16_t)old[l]; }
                FOR_LANES_W(l) { if (l == B) tab16[h[l]] = (uint16_t)pos[l]; }
            }
            if (m < 0) {
                if (B + 1 < 64 && B + 1 >= nv) break;             // the next probe would pass mflimit
                t0 += B + 1 - (pre ? 1 : 0);
                pre = false;
                continue;
            }

            // ---- a match at lane m ------------------------------------------------------------------------
            int ip = readlane(pos, m);
            int cand = (int)readlane(old, m);
            const bool zero_lit = pre && m == 0;
            if (!zero_lit) {
                // extend backwards while the bytes before both positions agree
                int room = imin(ip - anchor, cand);
                while (room > 0) {
                    LV<bool> eq;
                    FOR_LANES(l) { eq[l] = l < room && in[ip - 1 - l] == in[cand - 1 - l]; }
                    const int run = ctz64(~ballot(eq));
                    ip -= run; cand -= run; room -= run;
                    if (run < 64) break;
                }
            }
            const int lit = zero_lit ? 0 : ip - anchor;
            // forward match length beyond the 4 verified bytes
            int mcode = 0;
            {
                const int maxc = matchlimit - (ip + 4);
                for (;;) {
                    LV<int> len;
                    LV<bool> stop;
                    FOR_LANES(l) {
                        const int k = mcode + 4 * l;
                        int vb = maxc - k;
                        vb = vb < 0 ? 0 : (vb > 4 ? 4 : vb);
                        int ln = 0;
                        if (vb > 0) {
                            const uint32_t x = lds_ld32u(in, ip + 4 + k) ^ lds_ld32u(in, cand + 4 + k);
                            ln = x ? (int)(__builtin_ctz(x) >> 3) : 4;
                            if (ln > vb) ln = vb;
                        }
                        len[l] = ln;
                        stop[l] = ln < 4;
                    }
                    const uint64_t sm = ballot(stop);
                    if (sm) {
                        const int f = ctz64(sm);
                        mcode += 4 * f + readlane(len, f);
                        break;
                    }
                    mcode += 256;
                }
            }
            // ---- budget checks (limited-output rules) -------------------------------------------------------
            const int tok = op;
            int q = op + 1;
            const int lhs1 = q + lit + 8 + lit / 255;
            if (lhs1 > cap) return 0;
            need = imax(need, lhs1);
            if (lit >= 15) q += (lit - 15) / 255 + 1;
            const int litpos = q;
            q += lit;
            const int offpos = q;
            q += 2;
            const int lhs2 = q + 6 + (mcode + 240) / 255;
            if (lhs2 > cap) return 0;
            need = imax(need, lhs2);
            // ---- emit ----------------------------------------------------------------------------------------
            {
                const int off = ip - cand;
                const uint32_t token = (uint32_t)((lit >= 15 ? 15 : lit) << 4) | (uint32_t)(mcode >= 15 ? 15 : mcode);
                FOR_LANES(l) {
                    if (l == 0) out[tok] = (uint8_t)token;
                    if (l == 1) out[offpos] = (uint8_t)(off & 0xFF);
                    if (l == 2) out[offpos + 1] = (uint8_t)(off >> 8);
                }
                if (lit >= 15) emit_len_ext(out, tok + 1, lit - 15);
                emit_literals(in, anchor, out, litpos, lit);
                if (mcode >= 15) { emit_len_ext(out, q, mcode - 15); q += (mcode - 15) / 255 + 1; }
            }
            op = q;
            ip += mcode + 4;
            anchor = ip;
            if (ip >= mflimit_p1) break;
            sstart = ip + 1;
            t0 = 0;
            pre = true;
        }
    }
    // ---- last literals ------------------------------------------------------------------------------------
    {
        const int run = n - anchor;
        const int lhs = op + run + 1 + (run + 240) / 255;
        if (lhs > cap) return 0;
        need = imax(need, lhs);
        const uint32_t token = (uint32_t)((run >= 15 ? 15 : run) << 4);
        FOR_LANES(l) { if (l == 0) out[op] = (uint8_t)token; }
        op++;
        if (run >= 15) { emit_len_ext(out, op, run - 15); op += (run - 15) / 255 + 1; }
        emit_literals(in, anchor, out, op, run);
        op += run;
    }
    need_out = need;
    return op;
}

// true if all n bytes of the LDS plane equal its first byte
CIMG_DEV bool plane_is_run(const uint8_t* in, int n, uint32_t& value)
{
    LV<uint32_t> first;
    FOR_LANES(l) { first[l] = in[0]; }
    value = readlane(first, 0);
    const uint32_t w = value * 0x01010101u;
    const int words = n >> 2;
    for (int c = 0; c < words; c += 64) {
        LV<bool> bad;
        FOR_LANES(l) { bad[l] = c + l < words && *reinterpret_cast<const uint32_t*>(in + 4 * (c + l)) != w; }
        if (ballot(bad)) return false;
    }
    LV<bool> bad;
    FOR_LANES(l) { bad[l] = 4 * words + l < n && in[4 * words + l] != (uint8_t)value; }
    return ballot(bad) == 0;
}

CIMG_DEV void wave_copy_l2g(const uint8_t* lds, int off, uint8_t* g, int nbytes)
{
    const int units = nbytes >> 4;
    for (int u0 = 0; u0 < units; u0 += 64) {
        FOR_LANES(l) { if (u0 + l < units) st128u(g + 16 * (u0 + l), ld128a(lds + off + 16 * (u0 + l))); }
    }
    const int done = units << 4;
    FOR_LANES(l) { if (done + l < nbytes) g[done + l] = lds[off + done + l]; }
}

struct EncodeBlock {
    const EncodeArgs& a;
    uint8_t* lds;
    int b;
    int chunk, j, bsize, ns, neblock, ps, ts, tab0;
    bool active;
    const uint8_t* src;

    CIMG_DEV EncodeBlock(const EncodeArgs& a_, uint8_t* lds_, int b_) : a(a_), lds(lds_), b(b_) {}

    // phase A: global -> LDS byte planes (the shuffle filter)
    CIMG_DEV void phase_a(int wave)
    {
        chunk = find_chunk(a.descs, a.nchunks, b);
        const ChunkDesc& d = a.descs[chunk];
        j = b - d.blk0;
        active = !d.memcpyed;
        if (!active) return;
        ts = a.p.typesize;
        src = a.raw + d.raw_off + (int64_t)j * d.blocksize;
        const bool leftover_blk = (j == d.nblocks - 1 && d.leftover);
        bsize = leftover_blk ? d.leftover : d.blocksize;
        ns = (d.split && !leftover_blk) ? ts : 1;
        neblock = bsize / ns;
        ps = plane_stride(neblock);
        tab0 = ns * ps;
        const int units = bsize >> 4;
        const int tid0 = wave * 64;
        const bool shuf = a.p.filter == FILTER_SHUFFLE && ts > 1;
        const int ne = bsize / ts;
        // LDS address of plane p: split blocks keep planes ps apart, unsplit blocks keep them packed
        #define CIMG_PLANE(p) (ns > 1 ? (p) * ps : (p) * ne)
        if (shuf && ts == 2 && (ns > 1 || (ne & 7) == 0)) {
            for (int u0 = tid0; u0 < units; u0 += 256) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        const u128 x = ld128u(src + 16 * u);
                        // even bytes -> plane 0, odd bytes -> plane 1
                        const uint32_t e0 = byte_perm(x.y, x.x, 0x06040200u), o0 = byte_perm(x.y, x.x, 0x07050301u);
                        const uint32_t e1 = byte_perm(x.w, x.z, 0x06040200u), o1 = byte_perm(x.w, x.z, 0x07050301u);
                        uint32_t* d0 = reinterpret_cast<uint32_t*>(lds + CIMG_PLANE(0) + 8 * u);
                        uint32_t* d1 = reinterpret_cast<uint32_t*>(lds + CIMG_PLANE(1) + 8 * u);
                        d0[0] = e0; d0[1] = e1;
                        d1[0] = o0; d1[1] = o1;
                    }
                }
            }
        } else if (shuf && ts == 4 && (ns > 1 || (ne & 3) == 0)) {
            for (int u0 = tid0; u0 < units; u0 += 256) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        const u128 x = ld128u(src + 16 * u);      // 4 elements of 4 bytes
                        const uint32_t t0 = byte_perm(x.y, x.x, 0x05010400u), t1 = byte_perm(x.y, x.x, 0x07030602u);
                        const uint32_t v0 = byte_perm(x.w, x.z, 0x05010400u), v1 = byte_perm(x.w, x.z, 0x07030602u);
                        *reinterpret_cast<uint32_t*>(lds + CIMG_PLANE(0) + 4 * u) = byte_perm(v0, t0, 0x05040100u);
                        *reinterpret_cast<uint32_t*>(lds + CIMG_PLANE(1) + 4 * u) = byte_perm(v0, t0, 0x07060302u);
                        *reinterpret_cast<uint32_t*>(lds + CIMG_PLANE(2) + 4 * u) = byte_perm(v1, t1, 0x05040100u);
                        *reinterpret_cast<uint32_t*>(lds + CIMG_PLANE(3) + 4 * u) = byte_perm(v1, t1, 0x07060302u);
                    }
                }
            }
        } else if (!shuf) {
            // no filter (or typesize 1): planes are consecutive slices of the block
            for (int u0 = tid0; u0 < units; u0 += 256) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        const int k = 16 * u;
                        if (ns > 1 && (neblock & 15)) {
                            for (int i = 0; i < 16; i++) lds[((k + i) / neblock) * ps + (k + i) % neblock] = src[k + i];
                        } else {
                            st128a(lds + (ns > 1 ? (k / neblock) * ps + k % neblock : k), ld128u(src + k));
                        }
                    }
                }
            }
        } else {
            for (int u0 = tid0; u0 < units; u0 += 256) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        for (int i = 0; i < 16; i++) {
                            const int k = 16 * u + i;
                            if (k < ne * ts) lds[CIMG_PLANE(k % ts) + k / ts] = src[k]; else lds[k] = src[k];
                        }
                    }
                }
            }
        }
        if (wave == 0) {
            const int done = units << 4;
            FOR_LANES(l) {
                const int k = done + l;
                if (k < bsize) {
                    if (shuf) { if (k < ne * ts) lds[CIMG_PLANE(k % ts) + k / ts] = src[k]; else lds[k] = src[k]; }
                    else lds[ns > 1 ? (k / neblock) * ps + k % neblock : k] = src[k];
                }
            }
        }
        #undef CIMG_PLANE
    }

    // phase B: per-stream run check + LZ4
    CIMG_DEV void phase_b(int wave)
    {
        if (!active) return;
        uint8_t* slot = a.scratch + (int64_t)b * a.p.slot_bytes;
        for (int s = wave; s < ns; s += 4) {
            const uint8_t* in = lds + (ns > 1 ? s * ps : 0);
            uint8_t* tab = lds + tab0 + s * LZ4_HASH_BYTES;
            uint8_t* out = slot + (int64_t)s * neblock;
            StreamRec r;
            r.kind = REC_RAW; r.value = 0; r.csize = neblock; r.need = 0;
            uint32_t value;
            if (plane_is_run(in, neblock, value)) {
                r.kind = REC_RUN; r.value = (int32_t)value; r.csize = 0;
            } else {
                int need = 0;
                const int cb = lz4_encode_wave(in, tab, neblock, out, neblock, a.p.accel, need);
                if (cb > 0 && cb < neblock) {
                    r.kind = REC_LZ4; r.csize = cb; r.need = need;
                } else {
                    wave_copy_l2g(lds, ns > 1 ? s * ps : 0, out, neblock);
                }
            }
            StreamRec* dst = a.recs + (int64_t)b * a.p.streams_per_block + s;
            FOR_LANES(l) { if (l == 0) *dst = r; }
        }
    }
};

}  // namespace cimg
