// zstd_decode.h -- a Zstandard FRAME decoder (RFC 8878), the read side of blosc2's codec format 4.
//
// The reference offers enums::codec::zstd (enums.h:18-24, wrapper.h:74-119); c-blosc2 then stores every stream of a block
// as one complete zstd frame (ZSTD_compressCCtx, no dictionary, no checksum).  Compression with zstd is NOT built here (no
// zstd encoder can be pinned byte-exactly in this image) -- but chunks the reference wrote with it must
// still be readable, and decoding is defined by the format alone.  This file is that decoder: plain scalar code, written
// from the format description, with every table in a caller-provided work area (LDS on the device).  It is the slow path
// by design: one lane's worth of work per stream (all lanes of the wave execute it redundantly, uniformly), no attempt at
// the wave-parallel entropy decoding a fast path would need.  Pinned by frames libzstd 1.4.8 produced at levels 1 - 19
// (tests/golden/zstd_kat.npz, tests/golden/make_zstd_golden.py).
//
// Limits (anything else is ERR_CODEC_SUPPORT / ERR_DATA, never an out-of-range access): regenerated size of a frame <= the
// caller's capacity, no dictionary, window = the frame itself.  The output buffer doubles as the literal buffer (zstd_block).
#pragma once
#include "codec_types.h"
#include "wave.h"
#include <cstring>

namespace cimg {

enum : int { ZSTD_HUF_LOG_MAX = 11, ZSTD_FSE_LOG_MAX = 9 };

struct ZstdFseEntry { uint8_t sym, nb; uint16_t base; };      // (read as one little-endian dword in the sequence loop)
static_assert(sizeof(ZstdFseEntry) == 4, "one dword per entry");

struct ZstdWork {
    uint16_t huf[1 << ZSTD_HUF_LOG_MAX];            // symbol | code length << 8: one load per decoded literal
    ZstdFseEntry ll[1 << ZSTD_FSE_LOG_MAX], ml[1 << ZSTD_FSE_LOG_MAX], of[1 << ZSTD_FSE_LOG_MAX], wt[64];
    int16_t freq[256];
    uint16_t sdesc[256];
    uint8_t weights[256];
    int32_t ll_log, ml_log, of_log, huf_log, have_huf, have_tables;
    int32_t rank_count[ZSTD_HUF_LOG_MAX + 2], rank_idx[ZSTD_HUF_LOG_MAX + 2];
    int32_t tail;            // 1: the frame lies in global memory -- the section being decoded is copied to `stage` when it fits there
    int32_t stage_cap;       // bytes at stage
    uint8_t* stage;          // 16-byte aligned
};

// ---- bit readers --------------------------------------------------------------------------------------------------
// bits [off, off + n) of src (LSB-first within bytes), n <= 32; bytes outside [0, size) read as zero
// (the _lane form is what a lane computes for ITS OWN stream; zstd_bits is the wave-uniform use of it)
CIMG_DEV uint32_t zstd_bits_lane(const uint8_t* src, int size, int64_t off, int n)
{
    if (n <= 0) return 0;
    uint64_t acc = 0;
    const int64_t b0 = off >> 3;
    if (b0 >= 0 && b0 + 8 <= size) {                   // the usual case: eight bytes in one (unaligned) load
        memcpy(&acc, src + b0, 8);
        return (uint32_t)((acc >> (off & 7)) & ((1ull << n) - 1));
    }
    for (int k = 0; k < 5; k++) {
        const int64_t b = b0 + k;
        const uint64_t v = (b >= 0 && b < size) ? src[b] : 0;
        acc |= v << (8 * k);
    }
    return (uint32_t)((acc >> (off & 7)) & ((1ull << n) - 1));
}
CIMG_DEV uint32_t zstd_bits(const uint8_t* src, int size, int64_t off, int n) { return uni(zstd_bits_lane(src, size, off, n)); }
// n bits below bit position `top` of a backward stream (top itself is not changed); bits below position 0 read as zero
CIMG_DEV uint32_t zstd_rbits_lane(const uint8_t* src, int size, int top, int n)
{
    const int off = top - n;
    if (n <= 0) return 0;
    if (off >= 0) return zstd_bits_lane(src, size, off, n);
    const int miss = -off;
    if (miss >= n) return 0;
    return zstd_bits_lane(src, size, 0, n - miss) << miss;
}
// backward stream: *off is the bit position just above the next bits; bits below position 0 read as zero
CIMG_DEV uint32_t zstd_rbits(const uint8_t* src, int size, int64_t* off, int n)
{
    *off -= n;
    if (n <= 0) return 0;
    if (*off >= 0) return zstd_bits(src, size, *off, n);
    const int64_t miss = -*off;                       // bits that lie below the start of the stream
    if (miss >= n) return 0;
    return zstd_bits(src, size, 0, (int)(n - miss)) << miss;
}
// The backward reader of the entropy loops: `off` as above, with the 64 bits around it kept in a register -- one 8-byte load per
// ~ 30 bits consumed instead of one per read.  Near the start of the stream (and for streams shorter than 8 bytes) it falls
// back on zstd_rbits, which knows about the zero bits below position 0.
struct ZstdBack {
    const uint8_t* src;
    int size;
    int64_t off, lo;                                   // cont holds bits [lo, lo + 64); lo < 0: nothing loaded yet
    uint64_t cont;
    CIMG_DEV void init(const uint8_t* s, int n, int64_t o) { src = s; size = n; off = o; lo = -1; cont = 0; }
    CIMG_DEV uint32_t get(int n)
    {
        if (n <= 0) return 0;
        const int64_t t = off - n;
        if (size >= 8 && t >= 0) {
            if (lo < 0 || t < lo) {
                const int64_t b = ((off + 7) >> 3) - 8;
                lo = 8 * (b > 0 ? b : 0);
                uint64_t c;
                memcpy(&c, src + (lo >> 3), 8);
                cont = (uint64_t)uni((uint32_t)c) | ((uint64_t)uni((uint32_t)(c >> 32)) << 32);
            }
            off = t;
            return (uint32_t)((cont >> (t - lo)) & ((1ull << n) - 1));
        }
        return zstd_rbits(src, size, &off, n);
    }
};

// a byte every lane reads from the same address: wave-uniform, and said so (the decoder's control flow then stays scalar)
CIMG_DEV int zstd_u8(const uint8_t* p, int i) { return (int)uni((uint32_t)p[i]); }
CIMG_HD int zstd_highbit(uint32_t v) { int r = -1; while (v) { v >>= 1; r++; } return r; }

// ---- FSE ------------------------------------------------------------------------------------------------------------
// table description at src[0..size): fills w->freq, returns bytes consumed (< 0: error); *log_out = accuracy log
CIMG_DEV int zstd_fse_read_header(const uint8_t* src, int size, int max_log, int max_sym, ZstdWork* w, int* log_out, int* nsym_out)
{
    int64_t off = 0;
    const int log = 5 + (int)zstd_bits(src, size, off, 4);
    off += 4;
    if (log > max_log) return ERR_DATA;
    int remaining = 1 << log, sym = 0;
    while (remaining > 0 && sym <= max_sym) {
        const int bits = zstd_highbit((uint32_t)remaining + 1) + 1;
        int val = (int)zstd_bits(src, size, off, bits);
        off += bits;
        const int lower = (1 << (bits - 1)) - 1;
        const int threshold = (1 << bits) - 1 - (remaining + 1);
        if ((val & lower) < threshold) { off -= 1; val &= lower; }
        else if (val > lower) val -= threshold;
        const int proba = val - 1;
        remaining -= proba < 0 ? -proba : proba;
        w->freq[sym++] = (int16_t)proba;
        if (proba == 0) {
            int rep = (int)zstd_bits(src, size, off, 2);
            off += 2;
            for (;;) {
                for (int i = 0; i < rep && sym <= max_sym; i++) w->freq[sym++] = 0;
                if (rep != 3) break;
                rep = (int)zstd_bits(src, size, off, 2);
                off += 2;
            }
        }
        if (off > (int64_t)size * 8 + 16) return ERR_DATA;
    }
    if (remaining != 0 || sym > max_sym + 1) return ERR_DATA;
    *log_out = log;
    *nsym_out = sym;
    const int used = (int)((off + 7) >> 3);
    return used > size ? ERR_DATA : used;
}

CIMG_DEV int zstd_fse_build(ZstdFseEntry* t, int log, int nsym, ZstdWork* w)
{
    const int size = 1 << log;
    int high = size;
    for (int s = 0; s < nsym; s++)
        if (uni((int)w->freq[s]) == -1) { t[--high].sym = (uint8_t)s; w->sdesc[s] = 1; }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++) {
        const int fr = uni((int)w->freq[s]);
        if (fr <= 0) continue;
        w->sdesc[s] = (uint16_t)fr;
        for (int i = 0; i < fr; i++) {
            t[pos].sym = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos >= high);
        }
    }
    if (pos != 0) return ERR_DATA;
    for (int i = 0; i < size; i++) {
        const int s = uni((int)t[i].sym);
        const int next = uni((int)w->sdesc[s]);
        w->sdesc[s] = (uint16_t)(next + 1);
        const int nb = log - zstd_highbit((uint32_t)next);
        t[i].nb = (uint8_t)nb;
        t[i].base = (uint16_t)((next << nb) - size);
    }
    return 0;
}

CIMG_DEV void zstd_fse_rle(ZstdFseEntry* t, int sym) { t[0].sym = (uint8_t)sym; t[0].nb = 0; t[0].base = 0; }

// predefined distributions (RFC 8878 section 3.1.1.3.2.2)
CIMG_HD int zstd_default_freq(int which, int i)
{
    // which: 0 literal lengths (36 symbols, log 6), 1 offsets (29, log 5), 2 match lengths (53, log 6)
    if (which == 0) {
        if (i == 0) return 4;
        if (i == 1 || i == 25) return 3;
        if ((i >= 13 && i <= 15) || (i >= 27 && i <= 31)) return 1;
        if (i >= 32) return -1;
        return 2;
    }
    if (which == 1) {
        if (i >= 24) return -1;
        if (i >= 6 && i <= 8) return 2;
        return 1;
    }
    if (i == 0) return 1;
    if (i == 1) return 4;
    if (i == 2) return 3;
    if (i >= 3 && i <= 8) return 2;
    if (i >= 46) return -1;
    return 1;
}

CIMG_DEV int zstd_ll_base(int c) { return c < 16 ? c : c < 20 ? 16 + 2 * (c - 16) : c < 22 ? 24 + 4 * (c - 20) : c < 24 ? 32 + 8 * (c - 22) : c == 24 ? 48 : 64 << (c - 25); }
CIMG_DEV int zstd_ll_bits(int c) { return c < 16 ? 0 : c < 20 ? 1 : c < 22 ? 2 : c < 24 ? 3 : c == 24 ? 4 : c - 19; }
CIMG_DEV int zstd_ml_base(int c)
{
    if (c < 32) return c + 3;
    if (c < 36) return 35 + 2 * (c - 32);
    if (c < 38) return 43 + 4 * (c - 36);
    if (c < 40) return 51 + 8 * (c - 38);
    if (c < 42) return 67 + 16 * (c - 40);
    if (c == 42) return 99;
    return (128 << (c - 43)) + 3;
}
CIMG_DEV int zstd_ml_bits(int c) { return c < 32 ? 0 : c < 36 ? 1 : c < 38 ? 2 : c < 40 ? 3 : c < 42 ? 4 : c == 42 ? 5 : c - 36; }

// ---- Huffman ----------------------------------------------------------------------------------------------------------
// tree description at src: fills the decoding table, returns bytes consumed
CIMG_DEV int zstd_huf_read_tree(const uint8_t* src, int size, ZstdWork* w)
{
    if (size < 1) return ERR_DATA;
    const int hb = zstd_u8(src, 0);
    int n = 0, used;
    if (hb >= 128) {
        n = hb - 127;
        used = 1 + (n + 1) / 2;
        if (used > size) return ERR_DATA;
        for (int i = 0; i < n; i++) w->weights[i] = (i & 1) ? (zstd_u8(src, 1 + i / 2) & 15) : (zstd_u8(src, 1 + i / 2) >> 4);
    } else {
        if (hb == 0 || 1 + hb > size) return ERR_DATA;
        const uint8_t* f = src + 1;
        int log = 0, nsym = 0;
        const int h = zstd_fse_read_header(f, hb, 6, 12, w, &log, &nsym);
        if (h < 0) return h;
        int rc = zstd_fse_build(w->wt, log, nsym, w);
        if (rc < 0) return rc;
        const uint8_t* bs = f + h;
        const int bl = hb - h;
        if (bl < 1 || zstd_u8(bs, bl - 1) == 0) return ERR_DATA;
        ZstdBack br;
        br.init(bs, bl, (int64_t)bl * 8 - (8 - zstd_highbit(zstd_u8(bs, bl - 1))));
        const int mask = (1 << log) - 1;
        int s1 = (int)br.get(log), s2 = (int)br.get(log);
        for (;;) {
            if (n >= 254) return ERR_DATA;
            w->weights[n++] = w->wt[s1 & mask].sym;
            s1 = uni((int)w->wt[s1 & mask].base) + (int)br.get(uni((int)w->wt[s1 & mask].nb));
            if (br.off < 0) { w->weights[n++] = w->wt[s2 & mask].sym; break; }
            w->weights[n++] = w->wt[s2 & mask].sym;
            s2 = uni((int)w->wt[s2 & mask].base) + (int)br.get(uni((int)w->wt[s2 & mask].nb));
            if (br.off < 0) { w->weights[n++] = w->wt[s1 & mask].sym; break; }
        }
        used = 1 + hb;
    }
    // the last weight is implied: the sum of 2^(w-1) is a power of two
    uint32_t sum = 0;
    for (int i = 0; i < n; i++) { const int wt = uni((int)w->weights[i]); if (wt > ZSTD_HUF_LOG_MAX) return ERR_DATA; if (wt) sum += 1u << (wt - 1); }
    if (sum == 0) return ERR_DATA;
    const int maxbits = zstd_highbit(sum) + 1;
    if (maxbits > ZSTD_HUF_LOG_MAX) return ERR_DATA;
    const uint32_t left = (1u << maxbits) - sum;
    if (left & (left - 1)) return ERR_DATA;
    w->weights[n++] = (uint8_t)(zstd_highbit(left) + 1);
    // code lengths -> table: longest codes first, symbols in ascending order within a length
    int32_t* rank_count = w->rank_count;               // (in the work area, not on the stack: a private array is scratch memory on the device)
    int32_t* rank_idx = w->rank_idx;
    for (int i = 0; i <= ZSTD_HUF_LOG_MAX + 1; i++) rank_count[i] = 0;
    for (int i = 0; i < n; i++) { const int wt = uni((int)w->weights[i]); const int b = wt ? maxbits + 1 - wt : 0; w->weights[i] = (uint8_t)b; rank_count[b] = uni((int)rank_count[b]) + 1; }
    rank_idx[maxbits] = 0;
    for (int i = maxbits; i >= 1; i--) rank_idx[i - 1] = uni((int)rank_idx[i]) + uni((int)rank_count[i]) * (1 << (maxbits - i));
    if (uni((int)rank_idx[0]) != (1 << maxbits)) return ERR_DATA;
    for (int i = 0; i < n; i++) {
        const int b = uni((int)w->weights[i]);
        if (!b) continue;
        const int len = 1 << (maxbits - b), at = uni((int)rank_idx[b]);
        // (the 2^(maxbits - b) entries of a code: 64 per step)
        for (int k0 = 0; k0 < len; k0 += 64) { FOR_LANES_W(l) { if (k0 + l < len) w->huf[at + k0 + l] = (uint16_t)(i | (b << 8)); } }
        rank_idx[b] = at + len;
    }
    w->huf_log = maxbits;
    w->have_huf = 1;
    return used;
}

CIMG_DEV int zstd_huf_stream(const uint8_t* src, int size, uint8_t* out, int count, const ZstdWork* w)
{
    if (size < 1 || zstd_u8(src, size - 1) == 0) return ERR_DATA;
    const int log = w->huf_log, mask = (1 << log) - 1;
    ZstdBack br;
    br.init(src, size, (int64_t)size * 8 - (8 - zstd_highbit(zstd_u8(src, size - 1))));
    int state = (int)br.get(log);
    int n = 0;
    while (br.off > -log) {
        if (n >= count) return ERR_DATA;
        const uint32_t e = uni((uint32_t)w->huf[state & mask]);
        out[n++] = (uint8_t)e;
        const int nb = (int)(e >> 8);
        state = ((state << nb) + (int)br.get(nb)) & mask;
    }
    return (br.off == -log && n == count) ? 0 : ERR_DATA;
}

// The four streams of a literals section on FOUR LANES: lane k < 4 runs the loop of zstd_huf_stream for stream k -- its own bit
// position, state and output quarter in its own registers, table reads and literal stores at its own LDS addresses.  One round
// of the loop costs what one literal costs the scalar form (measured: 460 cycles, a dependent LDS read and some thirty
// instructions) and yields four.  The checks are those of the scalar form, per lane.
CIMG_DEV int zstd_huf_stream4(const uint8_t* ls, int z0, int z1, int z2, int z3, uint8_t* out, int per, int last, const ZstdWork* w)
{
    const int log = w->huf_log, mask = (1 << log) - 1;
    LV<int> base, size, count, obase, top, state, n;
    LV<bool> act, bad;
    FOR_LANES(l) {
        const int k = l & 3;
        base[l] = k == 0 ? 0 : k == 1 ? z0 : k == 2 ? z0 + z1 : z0 + z1 + z2;
        size[l] = k == 0 ? z0 : k == 1 ? z1 : k == 2 ? z2 : z3;
        count[l] = k == 3 ? last : per;
        obase[l] = k * per;
        act[l] = l < 4;
        n[l] = 0;
        const int lastb = size[l] >= 1 ? (int)ls[base[l] + size[l] - 1] : 0;
        bad[l] = act[l] & (lastb == 0);
        top[l] = size[l] * 8 - (8 - zstd_highbit((uint32_t)(lastb | 1)));     // (| 1: keep the arithmetic defined for a bad stream)
        state[l] = (int)zstd_rbits_lane(ls + base[l], size[l], top[l], log);
        top[l] -= log;
    }
    if (ballot(bad)) return ERR_DATA;
    for (int guard = 0; guard <= (per > last ? per : last) + 1; ++guard) {
        LV<bool> go;
        FOR_LANES(l) { go[l] = act[l] & (top[l] > -log); }
        if (!ballot(go)) break;
        FOR_LANES(l) { bad[l] = go[l] & (n[l] >= count[l]); }
        if (ballot(bad)) return ERR_DATA;
        FOR_LANES_W(l) {
            if (go[l]) {
                const uint32_t e = w->huf[state[l] & mask];
                out[obase[l] + n[l]] = (uint8_t)e;
                const int nb = (int)(e >> 8);
                state[l] = ((state[l] << nb) + (int)zstd_rbits_lane(ls + base[l], size[l], top[l], nb)) & mask;
                top[l] -= nb;
                n[l] += 1;
            }
        }
    }
    FOR_LANES(l) { bad[l] = act[l] & ((top[l] != -log) | (n[l] != count[l])); }
    return ballot(bad) ? ERR_DATA : 0;
}

// ---- byte movers: the only lane-parallel part of the decoder (64 bytes per step) -------------------------------------------
// non-overlapping copy
CIMG_DEV void zstd_copy(uint8_t* dst, const uint8_t* src, int n)
{
    for (int k0 = 0; k0 < n; k0 += 64) { FOR_LANES_W(l) { if (k0 + l < n) dst[k0 + l] = src[k0 + l]; } }
}
CIMG_DEV void zstd_fill(uint8_t* dst, uint8_t v, int n)
{
    for (int k0 = 0; k0 < n; k0 += 64) { FOR_LANES_W(l) { if (k0 + l < n) dst[k0 + l] = v; } }
}
// dst[k] = dst[k - offset], k = 0 .. n-1 in order: with offset < 64 the source is the repeating pattern in front of dst
CIMG_DEV void zstd_match(uint8_t* dst, int offset, int n)
{
    if (offset >= 64 || offset >= n) {                 // (n <= offset: nothing written here is read here)
        // (a step reads [k0 - offset, k0 + 64 - offset), all of it in front of the 64 bytes it writes)
        for (int k0 = 0; k0 < n; k0 += 64) { FOR_LANES_W(l) { if (k0 + l < n) dst[k0 + l] = dst[k0 + l - offset]; } }
        return;
    }
    const uint8_t* pat = dst - offset;
    for (int k0 = 0; k0 < n; k0 += 64) { FOR_LANES_W(l) { if (k0 + l < n) dst[k0 + l] = pat[(k0 + l) % offset]; } }
}

// any alignment on either side, 16 bytes per lane and four loads in flight (dst and src do not overlap)
CIMG_DEV void zstd_stage(uint8_t* dst, const uint8_t* src, int n)
{
    const int head = imin(n, (int)((16 - ((uintptr_t)dst & 15)) & 15));
    FOR_LANES_W(l) { if (l < head) dst[l] = src[l]; }
    dst += head; src += head; n -= head;
    const int units = n >> 4;
    for (int u0 = 0; u0 < units; u0 += 256) {
        LV<u128> t0, t1, t2, t3;
        FOR_LANES(l) {
            const int u = u0 + l;
            if (u < units) t0[l] = ld128u(src + 16 * u);
            if (u + 64 < units) t1[l] = ld128u(src + 16 * (u + 64));
            if (u + 128 < units) t2[l] = ld128u(src + 16 * (u + 128));
            if (u + 192 < units) t3[l] = ld128u(src + 16 * (u + 192));
        }
        FOR_LANES_W(l) {
            const int u = u0 + l;
            if (u < units) st128a(dst + 16 * u, t0[l]);
            if (u + 64 < units) st128a(dst + 16 * (u + 64), t1[l]);
            if (u + 128 < units) st128a(dst + 16 * (u + 128), t2[l]);
            if (u + 192 < units) st128a(dst + 16 * (u + 192), t3[l]);
        }
    }
    const int done = units << 4;
    FOR_LANES_W(l) { if (done + l < n) dst[done + l] = src[done + l]; }
}

// ---- one compressed block ---------------------------------------------------------------------------------------------
struct ZstdFrameState { int r0, r1, r2; };      // the three repeat offsets

CIMG_DEV int zstd_block(const uint8_t* src, int size, uint8_t* dst, int dpos, int dcap, ZstdWork* w, ZstdFrameState* fs)
{
    if (size < 1) return ERR_DATA;
    // ---- literals section
    const int ltype = zstd_u8(src, 0) & 3, sf = (zstd_u8(src, 0) >> 2) & 3;
    int regen, comp = 0, hdr, streams = 1;
    if (ltype < 2) {
        if (sf == 0 || sf == 2) { hdr = 1; regen = zstd_u8(src, 0) >> 3; }
        else if (sf == 1) { hdr = 2; if (size < 2) return ERR_DATA; regen = (zstd_u8(src, 0) >> 4) | (zstd_u8(src, 1) << 4); }
        else { hdr = 3; if (size < 3) return ERR_DATA; regen = (zstd_u8(src, 0) >> 4) | (zstd_u8(src, 1) << 4) | (zstd_u8(src, 2) << 12); }
    } else {
        if (size < 5) { if (size < 3 || sf >= 2) return ERR_DATA; }
        const uint64_t v = (uint64_t)zstd_u8(src, 0) | ((uint64_t)zstd_u8(src, 1) << 8) | ((uint64_t)zstd_u8(src, 2) << 16) | ((uint64_t)(size > 3 ? zstd_u8(src, 3) : 0) << 24) | ((uint64_t)(size > 4 ? zstd_u8(src, 4) : 0) << 32);
        if (sf == 0 || sf == 1) { hdr = 3; regen = (int)((v >> 4) & 0x3FF); comp = (int)((v >> 14) & 0x3FF); streams = sf == 0 ? 1 : 4; }
        else if (sf == 2) { hdr = 4; regen = (int)((v >> 4) & 0x3FFF); comp = (int)((v >> 18) & 0x3FFF); streams = 4; }
        else { hdr = 5; regen = (int)((v >> 4) & 0x3FFFF); comp = (int)((v >> 22) & 0x3FFFF); streams = 4; }
    }
    // The literals of a block have no buffer of their own: they are regenerated at the END of the output, dst[dcap - regen, dcap).
    // Sequence by sequence they move down to where they belong, and what is written never passes what is still unread: the
    // output stands at dpos + (literals consumed) + (bytes matched), the unread literals begin at dcap - regen + (literals
    // consumed), and dpos + regen + (all the block's matches) <= dcap in any frame that fits dst.  (A frame that does not fit
    // fails the bounds checks below, possibly after garbling literals of its own -- all of it inside dst.)  Copies run in
    // ascending 64-byte steps, load before store, which is safe for a source AT OR ABOVE its destination.
    if (regen > dcap - dpos) return ERR_DATA;
    uint8_t* const litbuf = dst + (dcap - regen);
    int pos = hdr;
    const uint8_t* lit = litbuf;
    // A frame in global memory (w->tail): every bit read and every literal run would wait for a load from there, once per
    // sequence.  Raw literals are brought over in one piece; the coded literals, then the sequences, go through w->stage when
    // they fit it (they are read where they lie when not).
    const int stage_cap = w->tail ? w->stage_cap : 0;
    if (ltype == 0) {
        if (pos + regen > size) return ERR_DATA;
        if (w->tail) zstd_stage(litbuf, src + pos, regen);
        else lit = src + pos;                              // raw literals of a frame in LDS are used where they lie
        pos += regen;
    } else if (ltype == 1) {
        if (pos + 1 > size) return ERR_DATA;
        zstd_fill(litbuf, zstd_u8(src, pos), regen);
        pos += 1;
    } else {
        if (pos + comp > size) return ERR_DATA;
        const uint8_t* ls = src + pos;
        int lsz = comp;
        if (comp <= stage_cap) { zstd_stage(w->stage, ls, comp); ls = w->stage; }
        if (ltype == 2) {
            const int t = zstd_huf_read_tree(ls, lsz, w);
            if (t < 0) return t;
            ls += t; lsz -= t;
        } else if (!w->have_huf) return ERR_DATA;
        if (streams == 1) {
            const int rc = zstd_huf_stream(ls, lsz, litbuf, regen, w);
            if (rc < 0) return rc;
        } else {
            if (lsz < 6) return ERR_DATA;
            const int s1 = zstd_u8(ls, 0) | (zstd_u8(ls, 1) << 8), s2 = zstd_u8(ls, 2) | (zstd_u8(ls, 3) << 8), s3 = zstd_u8(ls, 4) | (zstd_u8(ls, 5) << 8);
            const int s4 = lsz - 6 - s1 - s2 - s3;
            if (s4 < 1) return ERR_DATA;
            const int per = (regen + 3) / 4;
            if (3 * per > regen) return ERR_DATA;
            if (s1 < 1 || s2 < 1 || s3 < 1) return ERR_DATA;
            const int rc = zstd_huf_stream4(ls + 6, s1, s2, s3, s4, litbuf, per, regen - 3 * per, w);
            if (rc < 0) return rc;
        }
        pos += comp;
    }
    // ---- sequences section
    if (pos >= size) return ERR_DATA;
    int nseq = zstd_u8(src, pos++);
    if (nseq >= 128) {
        if (nseq == 255) { if (pos + 2 > size) return ERR_DATA; nseq = zstd_u8(src, pos) + (zstd_u8(src, pos + 1) << 8) + 0x7F00; pos += 2; }
        else { if (pos + 1 > size) return ERR_DATA; nseq = ((nseq - 128) << 8) + zstd_u8(src, pos++); }
    }
    int lpos = 0;                                          // literals consumed
    if (nseq > 0) {
        if (pos >= size) return ERR_DATA;
        const int modes = zstd_u8(src, pos++);
        if (modes & 3) return ERR_DATA;
        for (int k = 0; k < 3; k++) {
            const int mode = (modes >> (6 - 2 * k)) & 3;
            ZstdFseEntry* t = k == 0 ? w->ll : k == 1 ? w->of : w->ml;
            int* lg = k == 0 ? &w->ll_log : k == 1 ? &w->of_log : &w->ml_log;
            const int maxsym = k == 0 ? 35 : k == 1 ? 31 : 52, maxlog = k == 0 ? 9 : k == 1 ? 8 : 9;
            if (mode == 0) {
                const int nsym = k == 0 ? 36 : k == 1 ? 29 : 53;
                for (int i = 0; i < nsym; i++) w->freq[i] = (int16_t)zstd_default_freq(k, i);
                *lg = k == 1 ? 5 : 6;
                const int rc = zstd_fse_build(t, *lg, nsym, w);
                if (rc < 0) return rc;
            } else if (mode == 1) {
                if (pos >= size) return ERR_DATA;
                if (zstd_u8(src, pos) > maxsym) return ERR_DATA;
                zstd_fse_rle(t, zstd_u8(src, pos++));
                *lg = 0;
            } else if (mode == 2) {
                int nsym = 0;
                const int h = zstd_fse_read_header(src + pos, size - pos, maxlog, maxsym, w, lg, &nsym);
                if (h < 0) return h;
                const int rc = zstd_fse_build(t, *lg, nsym, w);
                if (rc < 0) return rc;
                pos += h;
            } else if (!(w->have_tables & (1 << k))) return ERR_DATA;
            w->have_tables |= 1 << k;
        }
        const uint8_t* bs = src + pos;
        const int bl = size - pos;
        if (bl >= 1 && bl <= stage_cap) { zstd_stage(w->stage, bs, bl); bs = w->stage; }
        if (bl < 1 || zstd_u8(bs, bl - 1) == 0) return ERR_DATA;
        ZstdBack br;
        br.init(bs, bl, (int64_t)bl * 8 - (8 - zstd_highbit(zstd_u8(bs, bl - 1))));
        const int llm = (1 << w->ll_log) - 1, ofm = (1 << w->of_log) - 1, mlm = (1 << w->ml_log) - 1;
        int sl = (int)br.get(w->ll_log), so = (int)br.get(w->of_log), sm = (int)br.get(w->ml_log);
        int r0 = fs->r0, r1 = fs->r1, r2 = fs->r2;
        for (int i = 0; i < nseq; i++) {
            // one 32-bit load per entry, and every lane loads the same three: say so, and the loop's control flow and addresses
            // stay on the scalar unit
            uint32_t pl, po, pm;
            memcpy(&pl, &w->ll[sl & llm], 4); memcpy(&po, &w->of[so & ofm], 4); memcpy(&pm, &w->ml[sm & mlm], 4);
            pl = uni(pl); po = uni(po); pm = uni(pm);
            ZstdFseEntry el, eo, em;
            el.sym = (uint8_t)pl; el.nb = (uint8_t)(pl >> 8); el.base = (uint16_t)(pl >> 16);
            eo.sym = (uint8_t)po; eo.nb = (uint8_t)(po >> 8); eo.base = (uint16_t)(po >> 16);
            em.sym = (uint8_t)pm; em.nb = (uint8_t)(pm >> 8); em.base = (uint16_t)(pm >> 16);
            if (eo.sym > 31 || el.sym > 35 || em.sym > 52) return ERR_DATA;
            // The extra bits of offset, match length and literal length lie one behind the other in the stream (first read =
            // highest bits), and so do the three state updates: one read each when they fit 32 bits (they nearly always do)
            // instead of three.
            const int xa = eo.sym, xb = zstd_ml_bits(em.sym), xc = zstd_ll_bits(el.sym);
            uint32_t xo, xm, xl;
            if (xa + xb + xc <= 32) {
                const uint32_t V = br.get(xa + xb + xc);
                xl = V & ((1u << xc) - 1);
                xm = (V >> xc) & ((1u << xb) - 1);
                xo = xa ? V >> (xb + xc) : 0;
            } else { xo = br.get(xa); xm = br.get(xb); xl = br.get(xc); }
            const uint32_t ov = (1u << eo.sym) + xo;
            const int mlen = zstd_ml_base(em.sym) + (int)xm;
            const int llen = zstd_ll_base(el.sym) + (int)xl;
            if (i + 1 < nseq) {
                const uint32_t V = br.get(el.nb + em.nb + eo.nb);              // <= 9 + 9 + 8 bits
                so = eo.base + (int)(V & ((1u << eo.nb) - 1));
                sm = em.base + (int)((V >> eo.nb) & ((1u << em.nb) - 1));
                sl = el.base + (int)(V >> (eo.nb + em.nb));
            }
            if (br.off < 0) return ERR_DATA;
            // (three named scalars, no indexing by idx: a dynamically indexed private array is scratch memory on the device)
            int offset;
            if (ov > 3) { offset = (int)(ov - 3); r2 = r1; r1 = r0; r0 = offset; }
            else {
                const int idx = (int)ov + (llen == 0 ? 1 : 0);
                if (idx == 1) offset = r0;
                else if (idx == 2) { offset = r1; r1 = r0; r0 = offset; }
                else { offset = idx == 3 ? r2 : r0 - 1; r2 = r1; r1 = r0; r0 = offset; }
            }
            if (llen > regen - lpos || llen > dcap - dpos) return ERR_DATA;
            zstd_copy(dst + dpos, lit + lpos, llen);
            dpos += llen; lpos += llen;
            if (offset <= 0 || offset > dpos || mlen > dcap - dpos) return ERR_DATA;
            zstd_match(dst + dpos, offset, mlen);
            dpos += mlen;
        }
        if (br.off != 0) return ERR_DATA;
        fs->r0 = r0; fs->r1 = r1; fs->r2 = r2;
    }
    const int rest = regen - lpos;
    if (rest > dcap - dpos) return ERR_DATA;
    zstd_copy(dst + dpos, lit + lpos, rest);
    return dpos + rest;
}

// One frame at src[0, size) -> dst[0, cap).  Returns the regenerated size or a negative blosc2 error code.
CIMG_DEV int zstd_decode_frame(const uint8_t* src, int size, uint8_t* dst, int cap, ZstdWork* w)
{
    if (size < 6) return ERR_DATA;
    if (!(zstd_u8(src, 0) == 0x28 && zstd_u8(src, 1) == 0xB5 && zstd_u8(src, 2) == 0x2F && zstd_u8(src, 3) == 0xFD)) return ERR_DATA;
    const int fhd = zstd_u8(src, 4);
    const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did = fhd & 3;
    if (fhd & 0x08) return ERR_DATA;
    if (did) return ERR_CODEC_SUPPORT;                      // dictionaries: blosc2 does not use them
    int pos = 5 + (single ? 0 : 1);
    const int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : fcs_flag == 1 ? 2 : fcs_flag == 2 ? 4 : 8;
    if (pos + fcs_bytes > size) return ERR_DATA;
    int64_t fcs = -1;
    if (fcs_bytes) {
        fcs = 0;
        for (int k = 0; k < fcs_bytes && k < 8; k++) fcs |= (int64_t)zstd_u8(src, pos + k) << (8 * k);
        if (fcs_bytes == 2) fcs += 256;
        if (fcs > cap) return ERR_WRITE_BUFFER;
    }
    pos += fcs_bytes;
    w->have_huf = 0; w->have_tables = 0;
    ZstdFrameState fs;
    fs.r0 = 1; fs.r1 = 4; fs.r2 = 8;
    int dpos = 0;
    for (int guard = 0; guard <= size; ++guard) {
        if (pos + 3 > size) return ERR_DATA;
        const int bh = zstd_u8(src, pos) | (zstd_u8(src, pos + 1) << 8) | (zstd_u8(src, pos + 2) << 16);
        pos += 3;
        const int last = bh & 1, type = (bh >> 1) & 3, bsz = bh >> 3;
        if (type == 0) {
            if (pos + bsz > size || bsz > cap - dpos) return ERR_DATA;
            zstd_copy(dst + dpos, src + pos, bsz);
            dpos += bsz; pos += bsz;
        } else if (type == 1) {
            if (pos + 1 > size || bsz > cap - dpos) return ERR_DATA;
            zstd_fill(dst + dpos, zstd_u8(src, pos), bsz);
            dpos += bsz; pos += 1;
        } else if (type == 2) {
            if (pos + bsz > size) return ERR_DATA;
            const int r = zstd_block(src + pos, bsz, dst, dpos, cap, w, &fs);
            if (r < 0) return r;
            dpos = r; pos += bsz;
        } else return ERR_DATA;
        if (last) {
            if (checksum) pos += 4;
            if (pos > size) return ERR_DATA;
            if (fcs >= 0 && fcs != dpos) return ERR_DATA;
            return dpos;
        }
    }
    return ERR_DATA;
}

}  // namespace cimg
