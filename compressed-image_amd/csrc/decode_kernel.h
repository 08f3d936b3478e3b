// decode_kernel.h -- blosc2 chunk decode for gfx950, one 256-thread workgroup per block.
//
// Replaces what the reference reaches through blosc2_decompress_ctx (blosc2/wrapper.h:246, called
// from schunk.h:164-180 per chunk, serially, on one CPU thread).  Here a whole batch of chunks is
// decoded by one launch: workgroup b handles block b of the batch-wide block numbering.
//
//   phase A  every wave reads the chunk header + bstarts[j] (wave-uniform loads), walks the
//            per-stream int32 csize words of its block and stages its streams in LDS:
//              run / zero stream  -> fill
//              raw stream         -> 16-byte global loads -> LDS
//              LZ4 stream         -> compressed bytes are parked at the END of the stream's LDS
//                                    region and decoded *in place* towards the front by one wave
//                                    (every lane parses "a token starts at my byte" for the next 64
//                                    input bytes, a scalar walk follows the real chain, literals of the
//                                    whole batch go out in one store, matches run in order)
//   barrier
//   phase B  all four waves undo the byte shuffle straight out of LDS (v_perm byte transposes for
//            typesize 2 and 4) and store the pixels with 16-byte coalesced writes.
//
// HBM traffic per block: compressed bytes read once, pixels written once (the algorithmic bytes).
// Blocks with at most one LZ4-coded plane are normally taken by the lean launch in front of this kernel
// (decode_lean_kernel.h); `done[b] == gen` marks them and this kernel skips them.
// Memcpyed and special-zero chunks skip LDS.  Written in the wave.h vocabulary; see wave.h for the
// host-emulation build used by tests/emu.
#pragma once
#include "codec_types.h"
#include "wave.h"
#include <type_traits>

namespace cimg {

#ifdef CIMG_EMULATE
extern long g_emu_dec_par, g_emu_dec_serial, g_emu_dec_batches;   // test-side statistics only
extern long g_emu_d2[8];                                          // lz4_decode_wave2: windows, tokens, flushes, rounds, parallel matches, serial matches, scalar sequences, empty windows
#endif

struct DecodeArgs {
    const ChunkDesc* descs;
    int32_t nchunks;
    const uint8_t* comp;      // compressed chunks live at comp + desc.comp_off
    uint8_t* raw;             // pixels go to raw + desc.raw_off
    int32_t* status;          // per chunk: 0 ok, <0 blosc2 error code
    int32_t lds_bytes;        // dynamic LDS size the launch provides
    uint64_t* dbg;            // diagnostics only: per-workgroup time stamps (nullptr in production)
    int32_t uniform_nblocks;  // > 0: every chunk has this many blocks (chunk = block / uniform_nblocks)
    uint32_t* done;           // optional, per block: == gen when cimg_decode_lean already wrote the block's pixels
    uint32_t gen;
    uint32_t* skipped;        // optional (lean launch): one word per wave of the launch (zeroed by the host), the blocks that wave left to the general kernel
    int32_t total_blocks;     // blocks of the batch (the persistent lean launch strides over them)
    // cimg_decode_blocks: workgroup k decodes block blk_first + k * blk_step (0, 1: every block); blk_step == 0: the LAST block of
    // chunk k -- the leftover blocks the lean kernel never takes, one workgroup per chunk.  cimg_decode_lean: blk_first != 0 says
    // that launch exists, and leftover blocks are then not counted in `skipped`.
    int32_t blk_first, blk_step;
    // cimg_decode_lean: how two waves that share a SIMD are kept from running their chains in lockstep (engine.hip: lean_tune;
    // decode_lean_kernel.h: run) -- bits 0-7: the younger wave of a SIMD starts n x 64 cycles late; bit 8: from its second block on
    // the younger wave runs at raised priority
    int32_t tune;
    // the zstd read path's two launches (zstd_walk_kernel.h): block blk_first + k has slot k of `zplan`; zcap bytes of records and of
    // literals a slot; zarea = the plane area of the replay launch (a block larger than that is nobody's).  cimg_decode_zstd with
    // zplan set reads exactly the blocks whose plan was marked as not fitting.
    uint8_t* zplan;
    int64_t zplan_stride;
    int32_t zcap, zarea;
    int32_t zlanes;           // > 0: walkers leave sequences to cimg_zstd_seq (zstd_seq_kernel.h), which gives this many lanes of a wave a block each
    int32_t zblocks;          // blocks of the group (cimg_zstd_seq: lane l of workgroup g has block g * zlanes + l of them)
};

CIMG_HD int round16(int x) { return (x + 15) & ~15; }
// room between the end of a stream's output and the end of its parked compressed bytes that keeps
// the in-place write pointer behind the read pointer for every valid LZ4 block
CIMG_HD int inplace_margin(int n) { return round16(n / 255) + 64; }
CIMG_HD int region_stride(int neblock) { return round16(neblock) + inplace_margin(neblock); }
// the same for a BloscLZ stream (blosclz_kernel.h): one control byte per 32 literals
CIMG_HD int blz_inplace_margin(int n) { return round16(n / 32 + 1) + 64; }
CIMG_HD int blz_region_stride(int neblock) { return round16(neblock) + blz_inplace_margin(neblock); }
// LDS of one block workgroup; sized for the codec with the larger margin (the host does not look at the codec bits)
inline int decode_lds_bytes(int blocksize, int typesize)
{
    int a = blz_region_stride(blocksize);
    if (typesize >= 1 && typesize <= MAX_STREAMS) {
        const int ne = blocksize / typesize;
        const int b = typesize * blz_region_stride(ne);
        if (b > a) a = b;
    }
    return a + 32;
}

// A chunk descriptor read by every lane from the same address is wave-uniform, but the compiler cannot
// know that (vector loads land in VGPRs) and would treat everything derived from it -- block geometry,
// early exits, loop bounds -- as divergent, wrapping uniform control flow in exec-mask bookkeeping.
// Passing each field through v_readfirstlane pins it to SGPRs.
CIMG_DEV int64_t uni64(int64_t x)
{
    const uint32_t lo = uni((uint32_t)(uint64_t)x), hi = uni((uint32_t)((uint64_t)x >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
CIMG_DEV ChunkDesc uniform_desc(const ChunkDesc* p)
{
    ChunkDesc d = *p;
    d.raw_off = uni64(d.raw_off); d.comp_off = uni64(d.comp_off);
    d.nbytes = uni(d.nbytes); d.destsize = uni(d.destsize); d.blocksize = uni(d.blocksize); d.nblocks = uni(d.nblocks);
    d.leftover = uni(d.leftover); d.blk0 = uni(d.blk0); d.flags = uni(d.flags); d.split = uni(d.split);
    d.memcpyed = uni(d.memcpyed); d.nstreams = uni(d.nstreams); d.assemble = uni(d.assemble);
    return d;
}

// find the chunk that owns batch-wide block index b (descs are ordered by blk0)
CIMG_DEV int find_chunk(const ChunkDesc* descs, int nchunks, int b, int uniform_nblocks = 0)
{
    if (uniform_nblocks > 0) return b / uniform_nblocks;
    int lo = 0, hi = nchunks - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (uni(descs[mid].blk0) <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// ---- wave-cooperative copies ---------------------------------------------------------------------
// global -> LDS, any source alignment, LDS offset 16-byte aligned
CIMG_DEV void wave_copy_g2l(const uint8_t* g, uint8_t* lds, int off, int nbytes)
{
    const int units = nbytes >> 4;
    constexpr int DEPTH = 8;                      // loads in flight per lane before the first LDS store
    int u0 = 0;
    for (; u0 + 64 * DEPTH <= units; u0 += 64 * DEPTH) {
        LV<u128> t[DEPTH];
        CIMG_UNROLL
        for (int k = 0; k < DEPTH; k++) { FOR_LANES(l) { t[k][l] = ld128u(g + 16 * (u0 + 64 * k + l)); } }
        CIMG_UNROLL
        for (int k = 0; k < DEPTH; k++) { FOR_LANES(l) { st128a(lds + off + 16 * (u0 + 64 * k + l), t[k][l]); } }
    }
    for (; u0 < units; u0 += 256) {
        LV<u128> t0, t1, t2, t3;
        FOR_LANES(l) {
            const int u = u0 + l;
            if (u < units) t0[l] = ld128u(g + 16 * u);
            if (u + 64 < units) t1[l] = ld128u(g + 16 * (u + 64));
            if (u + 128 < units) t2[l] = ld128u(g + 16 * (u + 128));
            if (u + 192 < units) t3[l] = ld128u(g + 16 * (u + 192));
        }
        FOR_LANES(l) {
            const int u = u0 + l;
            if (u < units) st128a(lds + off + 16 * u, t0[l]);
            if (u + 64 < units) st128a(lds + off + 16 * (u + 64), t1[l]);
            if (u + 128 < units) st128a(lds + off + 16 * (u + 128), t2[l]);
            if (u + 192 < units) st128a(lds + off + 16 * (u + 192), t3[l]);
        }
    }
    const int done = units << 4;
    FOR_LANES(l) { if (done + l < nbytes) lds[off + done + l] = g[done + l]; }
}

CIMG_DEV void wave_fill_lds(uint8_t* lds, int off, int nbytes, uint32_t byte)
{
    const uint32_t w = byte * 0x01010101u;
    const u128 q = {w, w, w, w};
    const int units = round16(nbytes) >> 4;          // regions are padded to 16
    for (int u0 = 0; u0 < units; u0 += 64) {
        FOR_LANES(l) { if (u0 + l < units) st128a(lds + off + 16 * (u0 + l), q); }
    }
}

// global -> global, used for memcpyed chunks; 256 threads = 4 waves, wave w takes every 4th KiB.
// DEPTH 16-byte loads per lane are in flight before the first store (a copy with one is pure HBM latency): four where four waves
// share a block, sixteen -- a whole 16 KiB plane in one round trip -- where ONE wave copies its streams inside the encode launch
// (a wave alone copies 17 KiB in five round trips of ~2 us at depth four: the tail of the launch).
template <int DEPTH = 4>
CIMG_DEV void wave_copy_g2g(const uint8_t* src, uint8_t* dst, int nbytes, int wave, int nwaves)
{
    const int units = nbytes >> 4;
    const int stride = nwaves * 64;
    int u0 = wave * 64;
    for (; u0 + (DEPTH - 1) * stride + 64 <= units; u0 += DEPTH * stride) {
        LV<u128> t[DEPTH];
        CIMG_UNROLL
        for (int k = 0; k < DEPTH; k++) { FOR_LANES(l) { t[k][l] = ld128u(src + 16 * (u0 + k * stride + l)); } }
        CIMG_UNROLL
        for (int k = 0; k < DEPTH; k++) { FOR_LANES(l) { st128u(dst + 16 * (u0 + k * stride + l), t[k][l]); } }
    }
    for (; u0 < units; u0 += stride) {
        LV<u128> t;
        FOR_LANES(l) { if (u0 + l < units) t[l] = ld128u(src + 16 * (u0 + l)); }
        FOR_LANES(l) { if (u0 + l < units) st128u(dst + 16 * (u0 + l), t[l]); }
    }
    if (wave == 0) {
        const int done = units << 4;
        FOR_LANES(l) { if (done + l < nbytes) dst[done + l] = src[done + l]; }
    }
}

CIMG_DEV void wave_fill_global(uint8_t* dst, int nbytes, uint32_t byte, int wave, int nwaves)
{
    const uint32_t w = byte * 0x01010101u;
    const u128 q = {w, w, w, w};
    const int units = nbytes >> 4;
    for (int u0 = wave * 64; u0 < units; u0 += nwaves * 64) {
        FOR_LANES(l) { if (u0 + l < units) st128u(dst + 16 * (u0 + l), q); }
    }
    if (wave == 0) {
        const int done = units << 4;
        FOR_LANES(l) { if (done + l < nbytes) dst[done + l] = (uint8_t)byte; }
    }
}

// ---- LZ4 block decode by one wave, in place inside LDS ----------------------------------------------
// Compressed bytes occupy [cs, cs + csize); output is written to [base, base + n).  Every LDS index is
// clamped to lds_limit so a corrupt stream can produce garbage but never an out-of-range access.
//
// The token stream is read through a 256-byte register window (lane l holds the aligned dword at
// wbase + 4l); the next 8 input bytes are pulled into a scalar with three v_readlane, so a whole short
// sequence header (token, <= 5 literals, offset) is parsed with scalar shifts and its literals are
// written straight from the scalar -- no LDS read on that path.  Matches with offset 1 (runs) are
// fills; offsets < 64 replicate a pattern with one read; long copies move 256 bytes per step.
CIMG_DEV void lds_copy_bytes(uint8_t* lds, int dst, int src, int len)
{
    // forward copy, dst < src or dst - src >= 64 (each 64-byte step reads before it writes)
    for (int c = 0; c < len; c += 64) {
        LV<uint32_t> t;
        FOR_LANES(l) { if (c + l < len) t[l] = lds[src + c + l]; }
        FOR_LANES_W(l) { if (c + l < len) lds[dst + c + l] = (uint8_t)t[l]; }
    }
}

// dst - src >= 256 or dst < src - 256 (no overlap inside a 256-byte step)
CIMG_DEV void lds_copy_wide(uint8_t* lds, int dst, int src, int len)
{
    const int head = imin((4 - (dst & 3)) & 3, len);
    if (head) {
        LV<uint32_t> t;
        FOR_LANES(l) { if (l < head) t[l] = lds[src + l]; }
        FOR_LANES_W(l) { if (l < head) lds[dst + l] = (uint8_t)t[l]; }
    }
    int done = head;
    while (len - done >= 4) {
        const int words = imin((len - done) >> 2, 64);
        LV<uint32_t> t;
        FOR_LANES(l) { if (l < words) t[l] = lds_ld32u(lds, src + done + 4 * l); }
        FOR_LANES_W(l) { if (l < words) *reinterpret_cast<uint32_t*>(lds + dst + done + 4 * l) = t[l]; }
        done += 4 * words;
    }
    if (done < len) {
        const int tail = len - done;
        LV<uint32_t> t;
        FOR_LANES(l) { if (l < tail) t[l] = lds[src + done + l]; }
        FOR_LANES_W(l) { if (l < tail) lds[dst + done + l] = (uint8_t)t[l]; }
    }
}

CIMG_DEV void lds_fill_bytes(uint8_t* lds, int dst, int len, uint32_t byte)
{
    for (int c = 0; c < len; c += 64) {
        FOR_LANES_W(l) { if (c + l < len) lds[dst + c + l] = (uint8_t)byte; }
    }
}

// dst - src >= 1024 (no overlap inside a 1 KiB step): 16 bytes per lane, aligned 128-bit stores
CIMG_DEV void lds_copy_huge(uint8_t* lds, int dst, int src, int len)
{
    const int head = imin((16 - (dst & 15)) & 15, len);
    if (head) {
        LV<uint32_t> t;
        FOR_LANES(l) { t[l] = lds[src + (l < head ? l : 0)]; }
        FOR_LANES_W(l) { if (l < head) lds[dst + l] = (uint8_t)t[l]; }
    }
    int done = head;
    while (len - done >= 16) {
        const int units = imin((len - done) >> 4, 64);
        LV<u128> t;
        FOR_LANES(l) {
            const int sa = src + done + 16 * (l < units ? l : 0);
            const int a = sa & ~3;
            const uint32_t sh = (uint32_t)sa & 3u;
            const uint32_t w0 = *reinterpret_cast<const uint32_t*>(lds + a);
            const uint32_t w1 = *reinterpret_cast<const uint32_t*>(lds + a + 4);
            const uint32_t w2 = *reinterpret_cast<const uint32_t*>(lds + a + 8);
            const uint32_t w3 = *reinterpret_cast<const uint32_t*>(lds + a + 12);
            const uint32_t w4 = *reinterpret_cast<const uint32_t*>(lds + a + 16);
            t[l].x = alignbyte(w1, w0, sh);
            t[l].y = alignbyte(w2, w1, sh);
            t[l].z = alignbyte(w3, w2, sh);
            t[l].w = alignbyte(w4, w3, sh);
        }
        FOR_LANES_W(l) { if (l < units) st128a(lds + dst + done + 16 * l, t[l]); }
        done += 16 * units;
    }
    if (done < len) {
        const int tail = len - done;
        LV<uint32_t> t;
        FOR_LANES(l) { t[l] = lds[src + done + (l < tail ? l : 0)]; }
        FOR_LANES_W(l) { if (l < tail) lds[dst + done + l] = (uint8_t)t[l]; }
    }
}

// t mod m for 0 <= t < 128, 1 <= m < 64 (float reciprocal + one fix-up; exact in this range)
CIMG_DEV int small_mod(int t, int m, float inv_m)
{
    int q = (int)((float)t * inv_m);
    int r = t - q * m;
    if (r < 0) r += m;
    if (r >= m) r -= m;
    return r;
}

// a match longer than 64 bytes, any offset >= 1
CIMG_DEV void lds_copy_match(uint8_t* lds, int dst, int src, int ml)
{
    const int offset = dst - src;
    if (offset >= 1024 && ml >= 512) {
        lds_copy_huge(lds, dst, src, ml);
    } else if (offset >= 256) {
        lds_copy_wide(lds, dst, src, ml);
    } else if (offset >= 64) {
        lds_copy_bytes(lds, dst, src, ml);
    } else if (offset == 1) {
        LV<uint32_t> t;
        FOR_LANES(l) { t[l] = lds[src]; }
        lds_fill_bytes(lds, dst, ml, readlane(t, 0));
    } else {
        // overlapping match: byte t of the match equals pattern byte t mod offset
        const int period = offset * ((63 + offset) / offset);     // smallest multiple of offset >= 64
        const float inv = fast_rcp((float)offset);
        for (int c = 0; c < ml; c += 64) {
            LV<uint32_t> t;
            FOR_LANES(l) {
                if (c + l < ml) t[l] = (c == 0) ? lds[src + small_mod(l, offset, inv)] : lds[dst + c + l - period];
            }
            FOR_LANES_W(l) { if (c + l < ml) lds[dst + c + l] = (uint8_t)t[l]; }
        }
    }
}

CIMG_DEV int lz4_decode_wave(uint8_t* lds, int base, int n, int cs, int csize, int lds_limit, uint64_t* dbg = nullptr, int item = 0)
{
    CIMG_PROF_DECL;
    (void)dbg; (void)item;
    int ip = cs;
    const int iend = cs + csize;
    int op = base;
    const int oend = base + n;
    int wbase = -4096;
    LV<uint32_t> win;
    const int clampmax = (lds_limit - 4) & ~3;

    // a match copy of <= 64 bytes whose source bytes have been requested but not yet stored: the next
    // sequence's header is parsed (scalar work on the register window) while the LDS read is in flight
    LV<uint32_t> pend;
    int pend_dst = 0, pend_len = 0;
#define CIMG_RETIRE()                                                                            \
    do {                                                                                         \
        if (pend_len) {                                                                          \
            FOR_LANES_W(l) { if (l < pend_len) lds[pend_dst + l] = (uint8_t)pend[l]; }           \
            pend_len = 0;                                                                        \
        }                                                                                        \
    } while (0)

    // 8 input bytes starting at `at` as a scalar (bytes past the window / the stream are garbage the
    // callers never use: every use is guarded by iend).  A reload reads LDS *behind* any pending store
    // of output (ip stays ahead of op), so it does not need the pending copy retired first.
#define CIMG_FETCH8(dst, at)                                                                     \
    do {                                                                                         \
        const int at_ = (at);                                                                    \
        if (at_ - wbase > 256 - 12 || at_ < wbase) {                                             \
            wbase = at_ & ~3;                                                                    \
            FOR_LANES(l) { win[l] = *reinterpret_cast<const uint32_t*>(lds + imin(wbase + 4 * l, clampmax)); } \
        }                                                                                        \
        const int i_ = (at_ - wbase) >> 2;                                                       \
        const uint64_t d0_ = readlane(win, i_), d1_ = readlane(win, i_ + 1), d2_ = readlane(win, i_ + 2); \
        const int sh_ = (at_ & 3) * 8;                                                           \
        dst = ((d0_ | (d1_ << 32)) >> sh_) | (sh_ ? (d2_ << (64 - sh_)) : 0);                    \
    } while (0)

    // LZ4 length extension: bytes are added up to and including the first one that is not 255.  64 of them
    // are looked at per LDS round trip (a 12 KiB match carries 48).
#define CIMG_LENEXT(acc)                                                                         \
    do {                                                                                         \
        for (;;) {                                                                               \
            if (ip >= iend) return ERR_DATA;                                                     \
            LV<uint32_t> eb_;                                                                    \
            LV<bool> stop_;                                                                      \
            FOR_LANES(l) {                                                                       \
                eb_[l] = lds[imin(ip + l, clampmax)];                                            \
                stop_[l] = (eb_[l] != 255) | (ip + l >= iend);                                   \
            }                                                                                    \
            const int f_ = ctz64(ballot(stop_));                                                 \
            if (f_ < 64) {                                                                       \
                if (ip + f_ >= iend) return ERR_DATA;                                            \
                acc += 255 * f_ + (int)readlane(eb_, f_);                                        \
                ip += f_ + 1;                                                                    \
                break;                                                                           \
            }                                                                                    \
            acc += 255 * 64;                                                                     \
            ip += 64;                                                                            \
        }                                                                                        \
    } while (0)

    for (;;) {
        if (ip >= iend) return ERR_DATA;
        // ---- batch path: parse every "simple" sequence of the next 64 input bytes at once ------------------
        // Every lane pretends a token starts at its byte and works out that sequence's header; a short
        // scalar walk then follows the real token chain through those per-lane answers.  Literals of the
        // whole batch are stored with one instruction, matches run in order from three v_readlane each.
        if (iend - ip >= 24) {
            CIMG_RETIRE();
            CIMG_PROF_LAP(0);                                   // scalar-path work since the last batch
            LV<uint32_t> tb, o0, o1, ex;
            LV<int> lit_l, lsrc_l, walk_l, len_l, off_l, ml_l;
            LV<bool> good, litok;
            FOR_LANES(l) {
                const int at = ip + l;
                tb[l] = lds[imin(at, clampmax)];
                const uint32_t lb = lds[imin(at + 1, clampmax)];     // literal-length byte, meaningful when the nibble is 15
                const int litn = (int)(tb[l] >> 4);
                const bool lext = litn == 15;
                lit_l[l] = lext ? 15 + (int)lb : litn;
                litok[l] = !lext | (lb < 255);
                lsrc_l[l] = l + (lext ? 2 : 1);                      // window offset of the first literal
                const int hp = imin(ip + lsrc_l[l] + lit_l[l], clampmax);
                o0[l] = lds[hp];
                o1[l] = lds[hp + 1];
                ex[l] = lds[hp + 2];
            }
            FOR_LANES(l) {
                const int mln = (int)(tb[l] & 15);
                const bool has_ext = mln == 15;
                off_l[l] = (int)(o0[l] | (o1[l] << 8));
                ml_l[l] = mln + 4 + (has_ext ? (int)ex[l] : 0);
                const int lend = lsrc_l[l] + lit_l[l];                // window offset just past the literals
                const int nxt = lend + 2 + (has_ext ? 1 : 0);
                // literals that do not sit inside the window are copied LDS -> LDS, which is done for the LAST
                // token of a batch only: such a token ends the chain walk (flagged by +1024)
                walk_l[l] = nxt + ((lit_l[l] >= 15) | (lend > 64) ? 1024 : 0);
                len_l[l] = lit_l[l] + ml_l[l];
                // lengths need no more bytes, a following token exists
                good[l] = litok[l] & (!has_ext | (ex[l] < 255)) & (ip + nxt < iend) & (off_l[l] != 0);
            }
            // what the walk reads: the window offset of the next token, or "this is no token the batch can take" (WALK_BAD)
            enum : int { WALK_BAD = 0x2000 };
            FOR_LANES(l) { walk_l[l] = good[l] ? walk_l[l] : (int)WALK_BAD; }
            CIMG_PROF_LAP(1);                                   // per-lane header parse
            // the real token chain: a scalar walk over the per-lane "next token" answers
            uint64_t tokens = 0;
            int s = 0;
            // Straight-line scalar code, unrolled eight times so that the tests are forward branches that are NOT taken while the
            // chain goes on (a rolled loop pays a taken backward branch, 20 cycles, per token).  24 steps cover the 22 tokens a window
            // can hold.  A step is ONE v_readlane and one compare on the way to the next: the answer of a lane says at once whether
            // the token counts (round 3's first form tested a bit of a ballot mask first: eight instructions a token, six of them
            // on the dependent path; this one has three there).  (Round 2 walked dense windows by pointer doubling over the LDS
            // crossbar: ~60 instructions but six dependent crossbar round trips -- the same time on the tiled family, 4 % worse on
            // the natural family.)
            int t_ = 0;                                         // what lane s answered (> 63: the walk ends at s)
#define CIMG_WALK_STEP t_ = readlane(walk_l, s); if (t_ > 63) break; tokens |= 1ull << s; s = t_;
            for (int rnd = 0; rnd < 3; ++rnd) {
                CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP
            }
#undef CIMG_WALK_STEP
            // the token the walk ended at: it counts unless it is none the batch can take; its answer says where the batch ends
            // (after 24 steps without an end, s is simply the next token, not taken yet)
            if (t_ > 63 && !(t_ & WALK_BAD)) { tokens |= 1ull << s; s = t_; }
            int biglast = 0;
            if (s >= 1024) { s -= 1024; biglast = 1; }
            CIMG_PROF_LAP(2);                                   // token chain walk
            // output position of every token: prefix sum of the sequence lengths over the token lanes
            LV<int> tlen, opos;
            LV<bool> istok;
            FOR_LANES(l) {
                istok[l] = (tokens >> l) & 1;
                tlen[l] = istok[l] ? len_l[l] : 0;
            }
            int acc;
            wave_exscan(tlen, opos, acc);
            if (acc > oend - op) {                                 // cut the batch at the first sequence that does not fit
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
                // literals: lane j belongs to the last token at or before j - 1
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
                    if (is_lit[l] & (k < own_lit[l]) & (l < litlim)) lds[op + own_pos[l] + k] = (uint8_t)tb[l];
                }
                if (biglast) lds_copy_bytes(lds, op + readlane(opos, tlast), ip + readlane(lsrc_l, tlast), readlane(lit_l, tlast));
                CIMG_PROF_LAP(3);                               // positions + literals
                // matches, in order; destination and source of every match are worked out for all tokens at once
                LV<int> dstv, srcv;
                LV<bool> badv;
                FOR_LANES(l) {
                    dstv[l] = op + opos[l] + lit_l[l];
                    srcv[l] = dstv[l] - off_l[l];
                    badv[l] = istok[l] & (srcv[l] < base);
                }
                if (ballot(badv)) return ERR_DATA;
                // Matches that read nothing this batch writes -- the source lies in front of the batch or inside the token's
                // own literals -- and do not overlap themselves (or are byte fills, offset 1) are independent of each other:
                // ONE LANE PER MATCH copies them all at once, eight bytes per step.  In image data that is nearly every
                // match of a batch; whatever is left runs in order below, after them (it may read what they wrote, they
                // never read what it writes).
                LV<bool> par;
                FOR_LANES(l) {
                    const int off = off_l[l], ml = ml_l[l];
                    par[l] = istok[l] & (ml <= 64) & ((off >= ml) | (off == 1))
                             & ((srcv[l] + (off == 1 ? 1 : ml) <= op) | (off <= lit_l[l]));
                }
                uint64_t parmask = ballot(par);
                if (popc64(parmask) < 3) parmask = 0;            // one or two: the in-order loop below is cheaper (latency of the gathers)
                if (parmask) {
                    // their descriptors are made dense (lane k = k-th such match), then SIXTEEN MATCHES PER STEP: four lanes
                    // per match, sixteen bytes per lane -- one store instruction moves a byte of every lane of sixteen matches
                    const int P = popc64(parmask);                       // <= 22: a sequence is at least three bytes
                    LV<int> rank, d0, d1, D0, D1;
                    FOR_LANES(l) {
                        rank[l] = par[l] ? lane_rank(parmask, l) : 63;   // lane 63 is never a dense slot
                        d0[l] = dstv[l] | (ml_l[l] << 18);               // LDS offsets are below 2^18
                        d1[l] = srcv[l] | (off_l[l] == 1 ? 1 << 18 : 0);
                    }
                    lane_scatter(d0, rank, D0);
                    lane_scatter(d1, rank, D1);
                    for (int g = 0; g < P; g += 16) {
                        // sixteen matches per step: four lanes per match, sixteen bytes per lane
                        LV<int> who, e0, e1;
                        FOR_LANES(l) { who[l] = g + (l >> 2); }
                        lane_gather(D0, who, e0);
                        lane_gather(D1, who, e1);
                        LV<u128> w;
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
                            w[l].x = f ? fb : x0;
                            w[l].y = f ? fb : alignbyte(q2, q1, sh);
                            w[l].z = f ? fb : alignbyte(q3, q2, sh);
                            w[l].w = f ? fb : alignbyte(q4, q3, sh);
                        }
                        FOR_LANES_W(l) {
                            const int rem = who[l] < P ? ((e0[l] >> 18) & 0x7F) - (l & 3) * 16 : 0;
                            uint8_t* d = lds + (e0[l] & 0x3FFFF) + (l & 3) * 16;
                            const uint32_t v[4] = {w[l].x, w[l].y, w[l].z, w[l].w};
                            // whole dwords with (unaligned) 32-bit stores, then the last 0..3 bytes: 7 LDS instructions instead of
                            // 16 byte stores (the byte stores were a third of the whole chain: 12 of 36 us on configs[1])
                            CIMG_UNROLL
                            for (int j = 0; j < 4; j++) { if (rem >= 4 * j + 4) lds_st32u(d + 4 * j, v[j]); }
                            const int t = rem > 0 ? (rem > 16 ? 16 : rem) & ~3 : 0;
                            const uint32_t last = t < 16 ? v[(t >> 2) & 3] : 0;
                            CIMG_UNROLL
                            for (int k = 0; k < 3; k++) { if (rem > t + k && t + k < 16) d[t + k] = (uint8_t)(last >> (8 * k)); }
                        }
                    }
                }
                CIMG_PROF_LAP(7);                               // lane-parallel matches
                uint64_t todo = tokens & ~parmask;
#ifdef CIMG_EMULATE
                g_emu_dec_par += popc64(parmask); g_emu_dec_serial += popc64(todo); g_emu_dec_batches++;
#endif
                while (todo) {
                    const int t = ctz64(todo);
                    todo &= todo - 1;
                    CIMG_PROF_COUNT(0);
                    const int dst = readlane(dstv, t);
                    const int src = readlane(srcv, t);
                    const int ml = readlane(ml_l, t);
                    const int offset = dst - src;
                    if (ml <= 64) {
                        LV<uint32_t> mv;
                        if (offset >= ml) {
                            FOR_LANES(l) { mv[l] = lds[src + (l < ml ? l : 0)]; }
                        } else if (offset == 1) {
                            FOR_LANES(l) { mv[l] = lds[src]; }
                        } else {
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
                CIMG_PROF_LAP(4); CIMG_PROF_COUNT(1);           // matches of the batch
                continue;
            }
        }
        uint64_t q;
        CIMG_FETCH8(q, ip);
        const uint32_t token = (uint32_t)(q & 0xFF);
        int lit = (int)(token >> 4);
        int ml = (int)(token & 15);
        int offset;
        const bool ext = ml == 15;
        if ((lit <= 4 || (lit == 5 && !ext)) && iend - ip >= 8) {
            // ---- short header: token, literals, offset (and one length byte) are all inside q ---------
            const int hdr = 1 + lit + 2;
            offset = (int)((q >> (8 * (1 + lit))) & 0xFFFF);
            int extra = 0;
            if (ext) {
                extra = (int)((q >> (8 * hdr)) & 0xFF);
                if (extra == 255) {
                    // long match: more length bytes follow
                    ip += hdr + 1;
                    ml += 255;
                    CIMG_LENEXT(ml);
                    extra = -1;
                } else {
                    ml += extra;
                }
            }
            if (extra >= 0) ip += hdr + (ext ? 1 : 0);
            if (lit > oend - op) return ERR_DATA;
            CIMG_RETIRE();
            if (lit) {
                const uint64_t lits = q >> 8;
                FOR_LANES_W(l) { if (l < lit) lds[op + l] = (uint8_t)(lits >> (8 * (l & 7))); }
                op += lit;
            }
        } else {
            // ---- general header ----------------------------------------------------------------------------
            CIMG_RETIRE();
            ip++;
            if (lit == 15) {
                CIMG_LENEXT(lit);
            }
            if (lit > iend - ip || lit > oend - op) return ERR_DATA;
            if (lit > 0) {
                if (lit >= 512) lds_copy_wide(lds, op, ip, lit); else lds_copy_bytes(lds, op, ip, lit);   // op < ip - 32: forward copy is safe
                ip += lit;
                op += lit;
            }
            if (ip == iend) break;                              // a block ends with literals
            if (iend - ip < 2) return ERR_DATA;
            uint64_t e;
            CIMG_FETCH8(e, ip);
            offset = (int)(e & 0xFFFF);
            ip += 2;
            if (ext) {
                CIMG_LENEXT(ml);
            }
        }
        if (ip > iend) return ERR_DATA;
        if (offset == 0 || offset > op - base) return ERR_DATA;
        ml += 4;
        if (ml > oend - op) return ERR_DATA;
        CIMG_PROF_COUNT(2);
        if (ml > 64) CIMG_PROF_COUNT(3);
        CIMG_PROF_LAP(5);                                       // scalar-path header
        const int src = op - offset;
        if (ml <= 64) {
            // request the source bytes now, store them after the next header has been parsed
            if (offset >= ml) {
                FOR_LANES(l) { pend[l] = lds[src + (l < ml ? l : 0)]; }
            } else if (offset == 1) {
                FOR_LANES(l) { pend[l] = lds[src]; }
            } else {
                const float inv = fast_rcp((float)offset);
                FOR_LANES(l) { pend[l] = lds[src + small_mod(l < ml ? l : 0, offset, inv)]; }
            }
            pend_dst = op;
            pend_len = ml;
        } else {
            lds_copy_match(lds, op, src, ml);
        }
        op += ml;
        CIMG_PROF_LAP(6);                                       // scalar-path match copy
    }
    CIMG_RETIRE();
    CIMG_PROF_LAP(0);
    CIMG_PROF_STORE(dbg, item);
#undef CIMG_FETCH8
#undef CIMG_LENEXT
#undef CIMG_RETIRE
    return op == oend ? 0 : ERR_DATA;
}

// ---- a batch of up to 63 tokens, executed at once (shared by lz4_decode_wave2 and the zstd executor) --------------------------------
// Token k < cnt: lit[k] literal bytes at offset lsrc[k], then a match of ml[k] bytes at distance off[k] (ml 0: none).  All
// offsets are relative to `lds` (MP: a generic pointer, or an address-space-3 one where the compiler cannot see that it is LDS);
// [base, oend) is the output, clampmax the last dword that may be READ.  op: where the batch's output starts (moved behind it).
// A batch that would overrun oend is executed up to the first token that does not fit: cut_out = its index (-1: none).
// Returns 0, or ERR_DATA for a match that reaches in front of `base`.
#define LZ_LD32(p) (*(LZ_WP)(p))
#define LZ_ST32(p, v) do { const uint32_t v_ = (v); __builtin_memcpy((p), &v_, 4); } while (0)
template <class MP, bool CLAMP = false>
CIMG_DEV int lz_batch_execute(MP lds, int base, int oend, int clampmax, int& op, int cnt_in, const LV<int>& lit_in, const LV<int>& ml_in,
                              const LV<int>& off_in, const LV<int>& lsrc_in, int& cut_out)
{
    using LZ_WP = typename std::conditional<std::is_same<MP, uint8_t*>::value, const uint32_t*, cimg_lds_cu32p>::type;
cut_out = -1;
int cnt = cnt_in;
if (cnt == 0) return 0;
LV<int> lit, ml, off, len, opos;
LV<bool> act;
FOR_LANES(l) {
    act[l] = l < cnt;
    lit[l] = act[l] ? lit_in[l] : 0;
    ml[l] = act[l] ? ml_in[l] : 0;
    off[l] = act[l] ? off_in[l] : 1;
    len[l] = lit[l] + ml[l];
}
    int acc;
    wave_exscan(len, opos, acc);
    if (acc > oend - op) {                             // cut at the first token that does not fit: the scalar path meets it next
        const int room = oend - op;
        LV<bool> over;
        FOR_LANES(l) { over[l] = act[l] & (opos[l] + len[l] > room); }
        const int f = ctz64(ballot(over));
        cut_out = f;
        cnt = f;
        acc = readlane(opos, f);
        FOR_LANES(l) { act[l] = l < cnt; if (!act[l]) { lit[l] = 0; ml[l] = 0; off[l] = 1; } }
        if (cnt == 0) return 0;
    }
    LV<int> P, D, S;
    LV<bool> badv;
    FOR_LANES(l) {
        P[l] = act[l] ? op + opos[l] : 0x7FFFFFFF;     // (lanes behind the batch: beyond every position the searches ask for)
        D[l] = op + opos[l] + lit[l];
        S[l] = D[l] - off[l];
        badv[l] = act[l] & (S[l] < base);
    }
    if (ballot(badv)) return ERR_DATA;
    // ---- literals: the bytes behind the token byte (and its length byte, from 15 literals on).  Every lane fetches the first 16
    // bytes of its run; runs longer than that are finished one after the other, in token order, by the whole wave; then every
    // lane stores its first bytes.  (The output of a token may lie on the compressed bytes of the tokens in front of it -- never on
    // those behind: all first pieces are in registers before anything is written, and the long runs go in order.)
    {
        LV<u128> v;
        LV<int> lsrc;
        LV<bool> longer;
        FOR_LANES(l) {
            lsrc[l] = lsrc_in[l];
            longer[l] = act[l] & (lit[l] > 16);
            const int sa = lsrc[l] > 0 ? lsrc[l] : 0;            // (the dword addresses are clamped, not the position: a run may END at the last readable byte)
            const int a = sa & ~3;
            const uint32_t sh = (uint32_t)sa & 3u;
            const uint32_t q0 = LZ_LD32(lds + imin(a, clampmax));
            const uint32_t q1 = LZ_LD32(lds + imin(a + 4, clampmax));
            const uint32_t q2 = LZ_LD32(lds + imin(a + 8, clampmax));
            const uint32_t q3 = LZ_LD32(lds + imin(a + 12, clampmax));
            const uint32_t q4 = LZ_LD32(lds + imin(a + 16, clampmax));
            v[l].x = alignbyte(q1, q0, sh); v[l].y = alignbyte(q2, q1, sh); v[l].z = alignbyte(q3, q2, sh); v[l].w = alignbyte(q4, q3, sh);
        }
        uint64_t todo = ballot(longer);
        while (todo) {
            const int t = ctz64(todo);
            todo &= todo - 1;
            lds_copy_bytes((uint8_t*)lds, readlane(P, t) + 16, readlane(lsrc, t) + 16, readlane(lit, t) - 16);
        }
        FOR_LANES_W(l) {
            const int nlit = act[l] ? imin(lit[l], 16) : 0;
            MP d = lds + (act[l] ? P[l] : base);
            const uint32_t w4[4] = {v[l].x, v[l].y, v[l].z, v[l].w};
            CIMG_UNROLL
            for (int j = 0; j < 4; j++) { if (nlit >= 4 * j + 4) LZ_ST32(d + 4 * j, w4[j]); }
            const int t = nlit & ~3;
            const uint32_t last = t < 16 ? w4[(t >> 2) & 3] : 0;
            CIMG_UNROLL
            for (int k = 0; k < 3; k++) { if (nlit > t + k && t + k < 16) d[t + k] = (uint8_t)(last >> (8 * k)); }
        }
    }
    // ---- matches.  Which batch tokens does a match's source touch?  need = the bytes it reads before writing (a fill reads
    // one, a match that overlaps itself its first `offset`); cntP(x) = tokens whose output starts at or below x.
    LV<bool> hasm, ovl;
    LV<int> lo_i, hi_i;
    {
        LV<int> ca, cb, xa, xb;
        FOR_LANES(l) {
            hasm[l] = act[l] & (ml[l] > 0);
            ovl[l] = hasm[l] & (((off[l] < ml[l]) & (off[l] != 1)) | (ml[l] > 64));      // copied by the whole wave, in its turn
            const int need = off[l] == 1 ? 1 : (off[l] < ml[l] ? off[l] : ml[l]);
            xa[l] = S[l]; xb[l] = S[l] + need - 1;
            ca[l] = 0; cb[l] = 0;
        }
        CIMG_UNROLL
        for (int k = 5; k >= 0; --k) {
            LV<int> ta, tb, pa, pb;
            FOR_LANES(l) { ta[l] = ca[l] + (1 << k) - 1; tb[l] = cb[l] + (1 << k) - 1; }     // index of the (count + 2^k)-th token
            lane_gather(P, ta, pa);
            lane_gather(P, tb, pb);
            FOR_LANES(l) {
                if (ta[l] < 64 && pa[l] <= xa[l]) ca[l] += 1 << k;
                if (tb[l] < 64 && pb[l] <= xb[l]) cb[l] += 1 << k;
            }
        }
        // tokens ja = ca - 1 .. jb = cb - 1 produce the source bytes (ca == 0: the source begins in front of the batch); the match
        // waits for the MATCHES of those below itself (their literals are in place; its own literals too)
        FOR_LANES(l) {
            lo_i[l] = ca[l] > 0 ? ca[l] - 1 : 0;
            hi_i[l] = imin(cb[l] - 1, l - 1);
        }
    }
    uint64_t done = ~ballot(hasm);
#ifdef CIMG_EMULATE
    g_emu_d2[2]++;
#endif
    for (int round = 0; round < 128 && ~done; ++round) {
        LV<bool> ready;
        FOR_LANES(l) {
            const uint64_t upto = hi_i[l] >= 63 ? ~0ull : ((1ull << (hi_i[l] + 1)) - 1);
            const uint64_t from = lo_i[l] >= 64 ? 0ull : (~0ull << lo_i[l]);
            const uint64_t deps = hi_i[l] >= lo_i[l] ? (upto & from) : 0ull;
            ready[l] = !((done >> l) & 1) & ((deps & ~done) == 0) & !ovl[l];
        }
        const uint64_t rmask = ballot(ready);
        // (one or two ready matches: the wave-wide copy of the FIRST pending match below is cheaper than the dense-descriptor
        // machinery -- that match is always executable: everything it depends on lies below it and is done)
#ifdef CIMG_EMULATE
        g_emu_d2[3]++; if (popc64(rmask) >= 3) g_emu_d2[4] += popc64(rmask); else g_emu_d2[5]++;
#endif
        if (popc64(rmask) >= 3) {
            const int R = popc64(rmask);
            LV<int> rank, d0, d1, D0, D1;
            FOR_LANES(l) {
                rank[l] = ready[l] ? lane_rank(rmask, l) : 63;   // (a batch has at most 63 tokens: slot 63 is nobody's)
                d0[l] = D[l] | ((ready[l] ? ml[l] : 0) << 18);            // (a ready match has at most 64 bytes)
                d1[l] = S[l] | (off[l] == 1 ? 1 << 18 : 0);
            }
            lane_scatter(d0, rank, D0);
            lane_scatter(d1, rank, D1);
            for (int g = 0; g < R; g += 16) {
                LV<int> who, e0, e1;
                FOR_LANES(l) { who[l] = g + (l >> 2); }
                lane_gather(D0, who, e0);
                lane_gather(D1, who, e1);
                LV<u128> w;
                FOR_LANES(l) {
                    const bool on = who[l] < R;
                    const bool f = (e1[l] >> 18) & 1;
                    const int sa = on ? (e1[l] & 0x3FFFF) + (f ? 0 : (l & 3) * 16) : base;
                    const int a = sa & ~3;
                    const uint32_t sh = (uint32_t)sa & 3u;
                    // (CLAMP: the output may end where readable memory ends -- the zstd executor in the host tests; the LZ4 planes
                    // have their in-place margin behind them)
                    const uint32_t q0 = LZ_LD32(lds + (CLAMP ? imin(a, clampmax) : a));
                    const uint32_t q1 = LZ_LD32(lds + (CLAMP ? imin(a + 4, clampmax) : a + 4));
                    const uint32_t q2 = LZ_LD32(lds + (CLAMP ? imin(a + 8, clampmax) : a + 8));
                    const uint32_t q3 = LZ_LD32(lds + (CLAMP ? imin(a + 12, clampmax) : a + 12));
                    const uint32_t q4 = LZ_LD32(lds + (CLAMP ? imin(a + 16, clampmax) : a + 16));
                    const uint32_t x0 = alignbyte(q1, q0, sh);
                    const uint32_t fb = (x0 & 0xFF) * 0x01010101u;
                    w[l].x = f ? fb : x0;
                    w[l].y = f ? fb : alignbyte(q2, q1, sh);
                    w[l].z = f ? fb : alignbyte(q3, q2, sh);
                    w[l].w = f ? fb : alignbyte(q4, q3, sh);
                }
                FOR_LANES_W(l) {
                    const int rem = who[l] < R ? ((e0[l] >> 18) & 0x7F) - (l & 3) * 16 : 0;
                    MP d = lds + (e0[l] & 0x3FFFF) + (l & 3) * 16;
                    const uint32_t v4[4] = {w[l].x, w[l].y, w[l].z, w[l].w};
                    CIMG_UNROLL
                    for (int j = 0; j < 4; j++) { if (rem >= 4 * j + 4) LZ_ST32(d + 4 * j, v4[j]); }
                    const int t = rem > 0 ? (rem > 16 ? 16 : rem) & ~3 : 0;
                    const uint32_t last = t < 16 ? v4[(t >> 2) & 3] : 0;
                    CIMG_UNROLL
                    for (int k = 0; k < 3; k++) { if (rem > t + k && t + k < 16) d[t + k] = (uint8_t)(last >> (8 * k)); }
                }
            }
            done |= rmask;
        } else {
            // the first match not done: everything it depends on lies below it and is done
            const int t = ctz64(~done);
            const int dst = readlane(D, t), src = readlane(S, t), mlen = readlane(ml, t);
            const int offset = dst - src;
            if (mlen <= 64) {
                LV<uint32_t> mv;
                if (offset >= mlen) { FOR_LANES(l) { mv[l] = lds[src + (l < mlen ? l : 0)]; } }
                else if (offset == 1) { FOR_LANES(l) { mv[l] = lds[src]; } }
                else {
                    const float inv = fast_rcp((float)offset);
                    FOR_LANES(l) { mv[l] = lds[src + small_mod(l < mlen ? l : 0, offset, inv)]; }
                }
                FOR_LANES_W(l) { if (l < mlen) lds[dst + l] = (uint8_t)mv[l]; }
            } else if (CLAMP) {
                // (byte-exact: nothing is read outside [src, dst + mlen) -- a step of 64 reads in front of what it writes when the
                // offset is at least 64 or the match does not overlap itself; else the pattern in front of dst is repeated)
                if (offset >= 64 || offset >= mlen) {
                    for (int k0 = 0; k0 < mlen; k0 += 64) {
                        LV<uint32_t> mv;
                        FOR_LANES(l) { mv[l] = lds[src + k0 + (k0 + l < mlen ? l : 0)]; }
                        FOR_LANES_W(l) { if (k0 + l < mlen) lds[dst + k0 + l] = (uint8_t)mv[l]; }
                    }
                } else {
                    for (int k0 = 0; k0 < mlen; k0 += 64) {
                        LV<uint32_t> mv;
                        FOR_LANES(l) { mv[l] = lds[src + (k0 + l) % offset]; }
                        FOR_LANES_W(l) { if (k0 + l < mlen) lds[dst + k0 + l] = (uint8_t)mv[l]; }
                    }
                }
            } else {
                lds_copy_match((uint8_t*)lds, dst, src, mlen);
            }
            done |= 1ull << t;
        }
    }
    op += acc;
    return 0;
}


#undef LZ_LD32
#undef LZ_ST32

// ---- LZ4 block decode, second form (round 4): tokens first, bytes 63 at a time -------------------------------------------------
// lz4_decode_wave above does everything window by window: parse 64 input bytes, walk the token chain, place the tokens' output,
// store literals, copy matches -- and a window of image data holds about thirteen tokens, so the fixed cost of every one of those
// steps (prefix sum, owner search, dense descriptors, crossbar gathers: 2300 of a window's 4000 cycles) is paid per thirteen tokens.
// Here a window only FINDS tokens (parse + walk); they are appended, a lane each, to a batch of up to 63 that is executed in one
// go: ONE prefix sum, the literal runs lane-parallel (16 bytes fetched per lane from the compressed bytes, which still lie ahead of
// the output: in-place decode keeps the write pointer behind the read pointer, and parsing ahead only reads), the matches in
// dependency ROUNDS: for every match the range of batch tokens whose output its source touches is found once (two binary searches
// over the tokens' output positions, crossbar gathers), a match is ready when all of those are done, and all ready matches of a round
// are copied sixteen a step, four lanes each.  Tokens a batch cannot take (literal runs of 15 bytes or more, matches above 64 bytes,
// the end of the stream) flush the batch and go through the scalar path of the first form.  Same results, also on damaged streams.
CIMG_DEV int lz4_decode_wave2(uint8_t* lds, int base, int n, int cs, int csize, int lds_limit)
{
    int ip = cs;
    const int iend = cs + csize;
    int op = base;
    const int oend = base + n;
    int wbase = -4096;
    LV<uint32_t> win;
    const int clampmax = (lds_limit - 4) & ~3;
    LV<uint32_t> pend;
    int pend_dst = 0, pend_len = 0;
#define CIMG_RETIRE()                                                                            \
    do {                                                                                         \
        if (pend_len) {                                                                          \
            FOR_LANES_W(l) { if (l < pend_len) lds[pend_dst + l] = (uint8_t)pend[l]; }           \
            pend_len = 0;                                                                        \
        }                                                                                        \
    } while (0)
#define CIMG_FETCH8(dst, at)                                                                     \
    do {                                                                                         \
        const int at_ = (at);                                                                    \
        if (at_ - wbase > 256 - 12 || at_ < wbase) {                                             \
            wbase = at_ & ~3;                                                                    \
            FOR_LANES(l) { win[l] = *reinterpret_cast<const uint32_t*>(lds + imin(wbase + 4 * l, clampmax)); } \
        }                                                                                        \
        const int i_ = (at_ - wbase) >> 2;                                                       \
        const uint64_t d0_ = readlane(win, i_), d1_ = readlane(win, i_ + 1), d2_ = readlane(win, i_ + 2); \
        const int sh_ = (at_ & 3) * 8;                                                           \
        dst = ((d0_ | (d1_ << 32)) >> sh_) | (sh_ ? (d2_ << (64 - sh_)) : 0);                    \
    } while (0)
#define CIMG_LENEXT(acc)                                                                         \
    do {                                                                                         \
        for (;;) {                                                                               \
            if (ip >= iend) return ERR_DATA;                                                     \
            LV<uint32_t> eb_;                                                                    \
            LV<bool> stop_;                                                                      \
            FOR_LANES(l) {                                                                       \
                eb_[l] = lds[imin(ip + l, clampmax)];                                            \
                stop_[l] = (eb_[l] != 255) | (ip + l >= iend);                                   \
            }                                                                                    \
            const int f_ = ctz64(ballot(stop_));                                                 \
            if (f_ < 64) {                                                                       \
                if (ip + f_ >= iend) return ERR_DATA;                                            \
                acc += 255 * f_ + (int)readlane(eb_, f_);                                        \
                ip += f_ + 1;                                                                    \
                break;                                                                           \
            }                                                                                    \
            acc += 255 * 64;                                                                     \
            ip += 64;                                                                            \
        }                                                                                        \
    } while (0)

    // the batch: lane k < cnt is the k-th token found and not yet executed
    LV<int> TA, TB, TP;                                    // literal count | match length << 8; offset; position of the token byte
    FOR_LANES(l) { TA[l] = 0; TB[l] = 0; TP[l] = 0; }
    int cnt = 0;
    bool force_scalar = false;

    // execute the batch: returns 0 or ERR_DATA
    // execute the batch (lz_batch_execute); a batch cut short puts the read position back on the token that did not fit
    auto flush = [&]() -> int {
        if (cnt == 0) return 0;
        LV<int> lit, ml, lsrc;
        FOR_LANES(l) {
            lit[l] = TA[l] & 0xFFFF;
            ml[l] = TA[l] >> 16;
            lsrc[l] = TP[l] + 1 + (lit[l] >= 15 ? 1 : 0);
        }
        int cut = -1;
        const int rc = lz_batch_execute<uint8_t*>(lds, base, oend, clampmax, op, cnt, lit, ml, TB, lsrc, cut);
        if (cut >= 0) { ip = readlane(TP, cut); force_scalar = true; }
        cnt = 0;
        return rc;
    };

    for (;;) {
        if (ip >= iend) return ERR_DATA;
        // ---- a window of tokens: parse every "simple" sequence of the next 64 input bytes, walk the chain, append them to the batch
        if (iend - ip >= 24 && !force_scalar) {
            LV<uint32_t> tb, o0, o1, ex;
            LV<int> lit_l, walk_l, off_l, ml_l;
            LV<bool> good;
            enum : int { WALK_BAD = 0x2000 };
            LV<int> lsrc_l;
            LV<bool> litok;
            FOR_LANES(l) {
                const int at = ip + l;
                tb[l] = lds[imin(at, clampmax)];
                const uint32_t lb = lds[imin(at + 1, clampmax)];     // literal-length byte, meaningful when the nibble is 15
                const int litn = (int)(tb[l] >> 4);
                const bool lext = litn == 15;
                lit_l[l] = lext ? 15 + (int)lb : litn;
                litok[l] = !lext | (lb < 255);
                lsrc_l[l] = l + (lext ? 2 : 1);
                const int hp = imin(ip + lsrc_l[l] + lit_l[l], clampmax);
                o0[l] = lds[hp];
                o1[l] = lds[hp + 1];
                ex[l] = lds[hp + 2];
            }
            // The walk only needs where the next token starts -- token byte and literal-length byte, the FIRST round trip.  The offset and
            // the match-length byte (second round trip, requested above) arrive while the chain is walked; a token they disqualify
            // (offset 0, a second length byte) cuts the chain afterwards.
            FOR_LANES(l) {
                const bool has_ext = (tb[l] & 15) == 15;
                const int nxt = lsrc_l[l] + lit_l[l] + 2 + (has_ext ? 1 : 0);
                walk_l[l] = (litok[l] & (ip + nxt < iend)) ? nxt : (int)WALK_BAD;
            }
            uint64_t tokens = 0;
            int s = 0, t_ = 0;
#define CIMG_WALK_STEP t_ = readlane(walk_l, s); if (t_ > 63) break; tokens |= 1ull << s; s = t_;
            for (int rnd = 0; rnd < 3; ++rnd) {
                CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP CIMG_WALK_STEP
            }
#undef CIMG_WALK_STEP
            if (t_ > 63 && !(t_ & WALK_BAD)) { tokens |= 1ull << s; s = t_; }
            FOR_LANES(l) {
                const int mln = (int)(tb[l] & 15);
                const bool has_ext = mln == 15;
                off_l[l] = (int)(o0[l] | (o1[l] << 8));
                ml_l[l] = mln + 4 + (has_ext ? (int)ex[l] : 0);
                good[l] = (!has_ext | (ex[l] < 255)) & (off_l[l] != 0);
            }
            {
                const uint64_t unfit = tokens & ~ballot(good);
                if (unfit) { const int f0 = ctz64(unfit); tokens &= (1ull << f0) - 1; s = f0; }
            }
#ifdef CIMG_EMULATE
            g_emu_d2[0]++; g_emu_d2[1] += popc64(tokens); if (!tokens) g_emu_d2[7]++;
#endif
            if (tokens) {
                const int k = popc64(tokens);
                if (cnt + k > 63) { const int rc = flush(); if (rc) return rc; if (force_scalar) continue; }
                LV<int> slot, a0, a1, a2, A0, A1, A2;
                FOR_LANES(l) {
                    const bool is = (tokens >> l) & 1;
                    slot[l] = is ? cnt + lane_rank(tokens, l) : 63;
                    a0[l] = lit_l[l] | (ml_l[l] << 16);
                    a1[l] = off_l[l];
                    a2[l] = ip + l;
                }
                lane_scatter(a0, slot, A0);
                lane_scatter(a1, slot, A1);
                lane_scatter(a2, slot, A2);
                FOR_LANES(l) {
                    const bool mine = (l >= cnt) & (l < cnt + k);
                    TA[l] = mine ? A0[l] : TA[l]; TB[l] = mine ? A1[l] : TB[l]; TP[l] = mine ? A2[l] : TP[l];
                }
                cnt += k;
                ip += s;
                continue;
            }
        }
        // ---- one sequence the scalar way (lz4_decode_wave's): everything found so far is executed first
        if (cnt) { const int rc = flush(); if (rc) return rc; if (force_scalar && ip >= iend) return ERR_DATA; }
        force_scalar = false;
#ifdef CIMG_EMULATE
        g_emu_d2[6]++;
#endif
        uint64_t q;
        CIMG_FETCH8(q, ip);
        const uint32_t token = (uint32_t)(q & 0xFF);
        int lit = (int)(token >> 4);
        int ml = (int)(token & 15);
        int offset;
        const bool ext = ml == 15;
        if ((lit <= 4 || (lit == 5 && !ext)) && iend - ip >= 8) {
            const int hdr = 1 + lit + 2;
            offset = (int)((q >> (8 * (1 + lit))) & 0xFFFF);
            int extra = 0;
            if (ext) {
                extra = (int)((q >> (8 * hdr)) & 0xFF);
                if (extra == 255) {
                    ip += hdr + 1;
                    ml += 255;
                    CIMG_LENEXT(ml);
                    extra = -1;
                } else {
                    ml += extra;
                }
            }
            if (extra >= 0) ip += hdr + (ext ? 1 : 0);
            if (lit > oend - op) return ERR_DATA;
            CIMG_RETIRE();
            if (lit) {
                const uint64_t lits = q >> 8;
                FOR_LANES_W(l) { if (l < lit) lds[op + l] = (uint8_t)(lits >> (8 * (l & 7))); }
                op += lit;
            }
        } else {
            CIMG_RETIRE();
            ip++;
            if (lit == 15) {
                CIMG_LENEXT(lit);
            }
            if (lit > iend - ip || lit > oend - op) return ERR_DATA;
            if (lit > 0) {
                if (lit >= 512) lds_copy_wide(lds, op, ip, lit); else lds_copy_bytes(lds, op, ip, lit);
                ip += lit;
                op += lit;
            }
            if (ip == iend) break;
            if (iend - ip < 2) return ERR_DATA;
            uint64_t e;
            CIMG_FETCH8(e, ip);
            offset = (int)(e & 0xFFFF);
            ip += 2;
            if (ext) {
                CIMG_LENEXT(ml);
            }
        }
        if (ip > iend) return ERR_DATA;
        if (offset == 0 || offset > op - base) return ERR_DATA;
        ml += 4;
        if (ml > oend - op) return ERR_DATA;
        const int src = op - offset;
        if (ml <= 64) {
            if (offset >= ml) {
                FOR_LANES(l) { pend[l] = lds[src + (l < ml ? l : 0)]; }
            } else if (offset == 1) {
                FOR_LANES(l) { pend[l] = lds[src]; }
            } else {
                const float inv = fast_rcp((float)offset);
                FOR_LANES(l) { pend[l] = lds[src + small_mod(l < ml ? l : 0, offset, inv)]; }
            }
            pend_dst = op;
            pend_len = ml;
            CIMG_RETIRE();                                     // (the next thing may be a window of tokens: nothing stays pending)
        } else {
            lds_copy_match(lds, op, src, ml);
        }
        op += ml;
    }
    CIMG_RETIRE();
#undef CIMG_FETCH8
#undef CIMG_LENEXT
#undef CIMG_RETIRE
    return op == oend ? 0 : ERR_DATA;
}

#ifdef CIMG_LZ4_DECODE_V1
#define CIMG_LZ4_DECODE lz4_decode_wave
#else
#define CIMG_LZ4_DECODE lz4_decode_wave2
#endif

// ---- byte-plane helpers ----------------------------------------------------------------------------
#ifdef CIMG_EMULATE
inline uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
    const uint64_t both = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) r |= (uint32_t)((both >> (8 * ((sel >> (8 * i)) & 7))) & 0xFF) << (8 * i);
    return r;
}
#else
CIMG_DEV uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#endif

// blosclz_kernel.h (included behind this header: it uses the copy helpers above)
CIMG_DEV int blosclz_decode_wave(uint8_t* lds, int base, int n, int cs, int csize, int lds_limit);

// ---- the kernel body ---------------------------------------------------------------------------------
struct DecodeBlock {
    const DecodeArgs& a;
    uint8_t* lds;
    int b;                 // batch-wide block index

    // results of the uniform header walk
    int chunk, j, bsize, ns, neblock, rs, ts, filter, mode;   // mode: 0 regular, 1 memcpyed, 2 zero, 3 skip
    const uint8_t* c;      // chunk base
    uint8_t* out;          // block output

    CIMG_DEV DecodeBlock(const DecodeArgs& a_, uint8_t* lds_, int b_) : a(a_), lds(lds_), b(b_) {}

    CIMG_DEV void fail(int code) { a.status[chunk] = code; }

    CIMG_DEV void phase_a(int wave)
    {
        mode = 3;
        if (a.done && uni((int)a.done[b]) == (int)a.gen) return;           // the lean kernel already produced this block
        chunk = find_chunk(a.descs, a.nchunks, b, a.uniform_nblocks);
        const ChunkDesc d = uniform_desc(a.descs + chunk);
        j = b - d.blk0;
        c = a.comp + d.comp_off;
        out = a.raw + d.raw_off + (int64_t)j * d.blocksize;
        bsize = (j == d.nblocks - 1 && d.leftover) ? d.leftover : d.blocksize;
        // the whole 32-byte header in one round trip (wave-uniform address)
        const u128 h0 = ld128u(c), h1 = ld128u(c + 16);
        const uint32_t w0 = uni(h0.x);
        const int flags = (int)((w0 >> 16) & 0xFF);
        ts = (int)(w0 >> 24);
        const int nbytes = (int)uni(h0.y), blocksize = (int)uni(h0.z), cbytes = (int)uni(h0.w);
        const uint32_t f0 = uni(h1.x), f1 = uni(h1.y), b2 = uni(h1.w);
        if ((w0 & 0xFF) > 5) { fail(ERR_VERSION_SUPPORT); return; }
        if (nbytes != d.nbytes || blocksize != d.blocksize || ts == 0 || cbytes < HEADER_LEN) { fail(ERR_INVALID_HEADER); return; }
        if (cbytes > d.destsize) { fail(ERR_READ_BUFFER); return; }          // the header claims more than the caller's buffer holds: nothing behind the header is read
        if ((flags & (FLAG_SHUFFLE | FLAG_BITSHUFFLE)) != (FLAG_SHUFFLE | FLAG_BITSHUFFLE)) { fail(ERR_VERSION_SUPPORT); return; }
        const int special = (int)((b2 >> 28) & 7);
        if (special == SPECIAL_ZERO) { mode = 2; wave_fill_global(out, bsize, 0, wave, 4); return; }
        if (special != 0) { fail(ERR_DATA); return; }
        if (flags & FLAG_MEMCPYED) {
            if (cbytes != nbytes + HEADER_LEN) { fail(ERR_DATA); return; }
            mode = 1;
            wave_copy_g2g(c + HEADER_LEN + (int64_t)j * blocksize, out, bsize, wave, 4);
            return;
        }
        const int fmt = flags >> 5;                                  // 0 blosclz, 1 lz4 / lz4hc
        if (fmt != 0 && fmt != 1) { fail(fmt == 4 ? ((flags & FLAG_DONT_SPLIT) || ts <= 1 ? STATUS_ZSTD_PENDING : STATUS_ZSTD_PENDING_SPLIT) : ERR_CODEC_SUPPORT); return; }   // zstd: cimg_decode_zstd's
        // filter pipeline: exactly one of {none, shuffle, bitshuffle}, in the last slot
        filter = (int)((f1 >> 8) & 0xFF);
        if (f0 != 0 || (f1 & 0xFF) != 0) { fail(ERR_CODEC_SUPPORT); return; }
        if (filter != FILTER_NONE && filter != FILTER_SHUFFLE && filter != FILTER_BITSHUFFLE) { fail(ERR_CODEC_SUPPORT); return; }
        if (filter == FILTER_BITSHUFFLE && !(flags & FLAG_DONT_SPLIT)) { fail(ERR_CODEC_SUPPORT); return; }   // bit rows are never split
        const bool leftover_blk = bsize != blocksize;
        ns = (!(flags & FLAG_DONT_SPLIT) && !leftover_blk) ? ts : 1;
        neblock = bsize / ns;
        rs = fmt == 0 ? blz_region_stride(neblock) : region_stride(neblock);
        if (ns * rs + 16 > a.lds_bytes) { fail(ERR_FAILURE); return; }
        if (cbytes < HEADER_LEN + 4 * d.nblocks) { fail(ERR_READ_BUFFER); return; }      // the bstarts table itself must be inside the chunk
        const int bstart = ld32s(c + HEADER_LEN + 4 * j);
        if (bstart < HEADER_LEN + 4 * d.nblocks || bstart > cbytes) { fail(ERR_DATA); return; }
        mode = 0;
        // walk the stream table; wave w stages streams w, w+4, ...
        int pos = bstart;
        for (int s = 0; s < ns; s++) {
            if (cbytes - pos < 4) { fail(ERR_READ_BUFFER); mode = 3; return; }
            const int cs = ld32s(c + pos);
            pos += 4;
            const int payload = cs > 0 ? cs : (cs < 0 ? 1 : 0);
            if (payload > cbytes - pos) { fail(ERR_READ_BUFFER); mode = 3; return; }
            if ((s & 3) == wave) {
                const int base = s * rs;
                if (cs == 0) {
                    wave_fill_lds(lds, base, neblock, 0);
                } else if (cs < 0) {
                    const int token = c[pos];
                    if (!(token & 1) || cs < -255) { fail(ERR_RUN_LENGTH); }
                    wave_fill_lds(lds, base, neblock, (uint32_t)(-cs) & 0xFF);
                } else if (cs == neblock) {
                    wave_copy_g2l(c + pos, lds, base, neblock);
                } else if (cs > neblock) {
                    fail(ERR_DATA);
                } else {
                    const int park = base + rs - round16(cs);
                    wave_copy_g2l(c + pos, lds, park, cs);
#ifdef CIMG_PROFILE
                    const int rc = fmt == 0 ? blosclz_decode_wave(lds, base, neblock, park, cs, a.lds_bytes)
                                            : lz4_decode_wave(lds, base, neblock, park, cs, a.lds_bytes, a.dbg, b);
#else
                    const int rc = fmt == 0 ? blosclz_decode_wave(lds, base, neblock, park, cs, a.lds_bytes)
                                            : CIMG_LZ4_DECODE(lds, base, neblock, park, cs, a.lds_bytes);
#endif
                    if (rc < 0) fail(rc);
                }
            }
            pos += payload;
        }
    }

    // LDS offset of filtered byte k of the block (k in shuffled order)
    CIMG_DEV int plane_base(int p) const { return ns > 1 ? p * rs : p * (bsize / ts); }

    CIMG_DEV void phase_b(int wave)
    {
        if (mode != 0) return;
        const int tid0 = wave * 64;
        const int units = bsize >> 4;
        // the dword fast paths need every plane to start on a dword (always true for split blocks; an unsplit
        // shuffled block -- the leftover block of a chunk -- has planes of bsize / ts bytes)
        if (filter == FILTER_SHUFFLE && ts == 2 && !(bsize & 1) && !(plane_base(1) & 3)) {
            const int p0 = plane_base(0), p1 = plane_base(1);
            for (int u0 = tid0; u0 < units; u0 += 256) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        const uint32_t a0 = *reinterpret_cast<const uint32_t*>(lds + p0 + 8 * u);
                        const uint32_t a1 = *reinterpret_cast<const uint32_t*>(lds + p0 + 8 * u + 4);
                        const uint32_t b0 = *reinterpret_cast<const uint32_t*>(lds + p1 + 8 * u);
                        const uint32_t b1 = *reinterpret_cast<const uint32_t*>(lds + p1 + 8 * u + 4);
                        u128 o;
                        o.x = byte_perm(b0, a0, 0x05010400u);
                        o.y = byte_perm(b0, a0, 0x07030602u);
                        o.z = byte_perm(b1, a1, 0x05010400u);
                        o.w = byte_perm(b1, a1, 0x07030602u);
                        st128u(out + 16 * u, o);
                    }
                }
            }
        } else if (filter == FILTER_SHUFFLE && ts == 4 && !(bsize & 3) && !(plane_base(1) & 3)) {
            const int p0 = plane_base(0), p1 = plane_base(1), p2 = plane_base(2), p3 = plane_base(3);
            for (int u0 = tid0; u0 < units; u0 += 256) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        const uint32_t A = *reinterpret_cast<const uint32_t*>(lds + p0 + 4 * u);
                        const uint32_t B = *reinterpret_cast<const uint32_t*>(lds + p1 + 4 * u);
                        const uint32_t C = *reinterpret_cast<const uint32_t*>(lds + p2 + 4 * u);
                        const uint32_t D = *reinterpret_cast<const uint32_t*>(lds + p3 + 4 * u);
                        const uint32_t t0 = byte_perm(B, A, 0x05010400u), t1 = byte_perm(B, A, 0x07030602u);
                        const uint32_t v0 = byte_perm(D, C, 0x05010400u), v1 = byte_perm(D, C, 0x07030602u);
                        u128 o;
                        o.x = byte_perm(v0, t0, 0x05040100u);
                        o.y = byte_perm(v0, t0, 0x07060302u);
                        o.z = byte_perm(v1, t1, 0x05040100u);
                        o.w = byte_perm(v1, t1, 0x07060302u);
                        st128u(out + 16 * u, o);
                    }
                }
            }
        } else if (filter == FILTER_BITSHUFFLE) {
            // inverse of load_block_bitshuffle: a thread gathers the 8 * ts row bytes of one group of 8 elements,
            // transposes them back and writes 8 * ts contiguous pixels bytes
            const int ne = bsize / ts, ne8 = ne & ~7, rowbytes = ne8 >> 3;
            for (int g0 = tid0; g0 < rowbytes; g0 += 256) {
                FOR_LANES(l) {
                    const int g = g0 + l;
                    if (g < rowbytes) {
                        uint8_t* o = out + (int64_t)8 * g * ts;
                        if (ts == 2 || ts == 4) {
                            uint64_t y[4];
                            for (int j = 0; j < ts; j++) {
                                uint64_t x = 0;
                                for (int k = 0; k < 8; k++) x |= (uint64_t)lds[(8 * j + k) * rowbytes + g] << (8 * k);
                                y[j] = bit_transpose8(x);
                            }
                            if (ts == 2) {
                                const uint32_t a0 = (uint32_t)y[0], a1 = (uint32_t)(y[0] >> 32), b0 = (uint32_t)y[1], b1 = (uint32_t)(y[1] >> 32);
                                u128 q;
                                q.x = byte_perm(b0, a0, 0x05010400u);
                                q.y = byte_perm(b0, a0, 0x07030602u);
                                q.z = byte_perm(b1, a1, 0x05010400u);
                                q.w = byte_perm(b1, a1, 0x07030602u);
                                st128u(o, q);
                            } else {
                                for (int half = 0; half < 2; half++) {
                                    const uint32_t A = (uint32_t)(y[0] >> (32 * half)), B = (uint32_t)(y[1] >> (32 * half));
                                    const uint32_t Cc = (uint32_t)(y[2] >> (32 * half)), D = (uint32_t)(y[3] >> (32 * half));
                                    const uint32_t t0 = byte_perm(B, A, 0x05010400u), t1 = byte_perm(B, A, 0x07030602u);
                                    const uint32_t v0 = byte_perm(D, Cc, 0x05010400u), v1 = byte_perm(D, Cc, 0x07030602u);
                                    u128 q;
                                    q.x = byte_perm(v0, t0, 0x05040100u);
                                    q.y = byte_perm(v0, t0, 0x07060302u);
                                    q.z = byte_perm(v1, t1, 0x05040100u);
                                    q.w = byte_perm(v1, t1, 0x07060302u);
                                    st128u(o + 16 * half, q);
                                }
                            }
                        } else {
                            for (int j = 0; j < ts; j++) {
                                uint64_t x = 0;
                                for (int k = 0; k < 8; k++) x |= (uint64_t)lds[(8 * j + k) * rowbytes + g] << (8 * k);
                                const uint64_t y = bit_transpose8(x);
                                for (int i = 0; i < 8; i++) o[i * ts + j] = (uint8_t)(y >> (8 * i));
                            }
                        }
                    }
                }
            }
            const int done = ne8 * ts;
            for (int k0 = done + tid0; k0 < bsize; k0 += 256) {
                FOR_LANES(l) { if (k0 + l < bsize) out[k0 + l] = lds[k0 + l]; }
            }
            return;
        } else if (filter == FILTER_NONE || ts == 1) {
            // planes are consecutive slices of the block
            for (int u0 = tid0; u0 < units; u0 += 256) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        const int k = 16 * u;
                        if (ns > 1 && ((neblock & 15) != 0)) {
                            for (int i = 0; i < 16; i++) out[k + i] = lds[((k + i) / neblock) * rs + (k + i) % neblock];
                        } else {
                            const int off = ns > 1 ? (k / neblock) * rs + k % neblock : k;
                            st128u(out + k, ld128a(lds + off));
                        }
                    }
                }
            }
        } else {
            // generic typesize: byte gather
            const int ne = bsize / ts;
            for (int u0 = tid0; u0 < units; u0 += 256) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        for (int i = 0; i < 16; i++) {
                            const int k = 16 * u + i;
                            out[k] = (k < ne * ts) ? lds[plane_base(k % ts) + k / ts] : lds[k];
                        }
                    }
                }
            }
        }
        // tail: bsize % 16 bytes (and, for odd sizes, the verbatim bytes after ne*ts)
        if (wave == 0) {
            const int done = units << 4;
            const int ne = bsize / ts;
            FOR_LANES(l) {
                const int k = done + l;
                if (k < bsize) {
                    if (filter == FILTER_SHUFFLE && ts > 1)
                        out[k] = (k < ne * ts) ? lds[plane_base(k % ts) + k / ts] : lds[k];
                    else
                        out[k] = lds[ns > 1 ? (k / neblock) * rs + k % neblock : k];
                }
            }
        }
    }
};

}  // namespace cimg
