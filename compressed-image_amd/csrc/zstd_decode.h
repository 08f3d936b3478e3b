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
#include "decode_kernel.h"      // lz_batch_execute
#include <cstring>
#include <cstddef>
#include <type_traits>

namespace cimg {

enum : int { ZSTD_HUF_LOG_MAX = 11, ZSTD_FSE_LOG_MAX = 9 };

struct ZstdFseEntry { uint8_t sym, nb; uint16_t base; };      // (read as one little-endian dword in the sequence loop)
static_assert(sizeof(ZstdFseEntry) == 4, "one dword per entry");

// ---- a frame WALKED instead of decoded (zstd_walk_kernel.h) ---------------------------------------------------------------
// The entropy decoder needs tables and a bit stream, not the 32 KiB its output takes: a launch of walkers (eight waves a CU instead
// of three) reads what the frames SAY into a plan in global memory -- per frame a list of ZstdOp, the sequences as 8-byte records,
// the Huffman-coded literals decoded -- and a second launch, whose waves hold an output block and nothing else, replays the plans.
enum : uint32_t { ZOP_RAW = 1, ZOP_FILL = 2, ZOP_SEQ = 3, ZOP_END = 4 };
struct ZstdOp {
    uint32_t kind;         // ZOP_*
    uint32_t size;         // RAW / FILL: bytes of the block; SEQ: the block's literal bytes (regen); END: the frame's regenerated size
    uint64_t ptr;          // RAW: where the bytes lie; FILL: the byte; SEQ: where the literals lie -- or, lit_fill set, the byte they all are
    uint32_t nseq, rec;    // SEQ: its sequences are records rec .. rec + nseq - 1 of the plan
    uint32_t lit_fill;     // SEQ: 1 = run-length literals
    uint32_t stream;       // the stream of the blosc2 block this frame is
};
static_assert(sizeof(ZstdOp) == 32, "ops are read as two 16-byte loads");
// a sequence: literal length | match length << 21 | offset << 42 (all three at most the output capacity, < 2^21)
CIMG_HD uint64_t zstd_record(uint32_t ll, uint32_t ml, uint32_t of) { return (uint64_t)ll | ((uint64_t)ml << 21) | ((uint64_t)of << 42); }
// A walker may leave the sequences of a compressed block undecoded as well (ZstdWork::defer): the block becomes a JOB -- its three
// FSE tables copied to the plan, where its bit stream lies, where its records go -- and a third kind of launch decodes the jobs of
// many blocks at once, one LANE per block (zstd_seq_kernel.h).
struct ZstdSeqJob {
    uint64_t bs;                               // address of the bit stream (in the chunk)
    uint32_t bl, nseq, rec;                    // its bytes; the block's sequences; its first record
    uint32_t tab;                              // byte offset of its tables in the plan's table area: ll[512] ml[512] of[256], 4 bytes an entry
    uint8_t ll_log, of_log, ml_log, first;     // first: the first compressed block of its frame (the repeat offsets start over: 1, 4, 8)
    uint32_t pad;
};
static_assert(sizeof(ZstdSeqJob) == 32, "jobs are read as two 16-byte loads");
// ... and the Huffman-coded literals of a block likewise (zstd_lit_kernel.h: one lane per STREAM, sixteen blocks a wave): the
// decoding table copied to the plan, the streams where they lie in the chunk, the literals' place in the plan's literal area.
struct ZstdLitJob {
    uint64_t src;                              // address of the streams: the jump table of a 4-stream section first
    uint32_t lsz, regen;                       // bytes at src; literals to regenerate
    uint32_t out;                              // their place in the plan's literal area
    uint32_t tab;                              // byte offset of the decoding table in the plan's table area: 2^log entries, symbol | code length << 8
    uint32_t log, streams;                     // table log (<= 11); 1 or 4 streams
    uint32_t pad[8];
};
static_assert(sizeof(ZstdLitJob) == 64, "literal jobs are read as four 16-byte loads");
enum : int { ZSTD_HUF_TABLE_BYTES = 2 << ZSTD_HUF_LOG_MAX };   // 4096
enum : int { ZSTD_JOB_TABLE_BYTES = 4 * ((1 << ZSTD_FSE_LOG_MAX) + (1 << ZSTD_FSE_LOG_MAX) + (1 << (ZSTD_FSE_LOG_MAX - 1))) };   // 5120
enum : int { ZSTD_WALK_OVERFLOW = -2000 };     // internal: the plan does not fit its slot -- the block goes to the decoder that needs none

struct ZstdWork {
    uint16_t huf[1 << ZSTD_HUF_LOG_MAX];            // symbol | code length << 8: one load per decoded literal
    ZstdFseEntry ll[1 << ZSTD_FSE_LOG_MAX], ml[1 << ZSTD_FSE_LOG_MAX], of[1 << (ZSTD_FSE_LOG_MAX - 1)], wt[64];   // (offset codes: table log <= 8)
    int16_t freq[256];
    uint16_t sdesc[256];
    uint8_t weights[256];
    int32_t ll_log, ml_log, of_log, huf_log, have_huf, have_tables;
    int32_t rank_count[ZSTD_HUF_LOG_MAX + 2], rank_idx[ZSTD_HUF_LOG_MAX + 2];
    // the memory around the output that may be READ with aligned dword loads (both 4-byte aligned; zstd_execute_batch fetches 16
    // bytes of a copy's source with five of them and never reads outside): the kernel's whole LDS; exactly the output buffer in
    // the host tests.  nullptr: every copy goes the byte-exact serial way.
    const uint8_t* mem_lo;
    const uint8_t* mem_hi;
    // the sequences section of the block being decoded: arguments and results of zstd_sequences (an out-of-line function reads its
    // arguments here instead of taking a dozen of them through vector registers)
    const uint8_t* seq_bs;   // the bit stream and its length; seq_bs_lds: it lies in LDS (the stage, or a frame staged whole)
    uint8_t* seq_dst;        // the frame's output
    const uint8_t* seq_lit;  // the block's literals
    int32_t seq_bl, seq_bs_lds, seq_nseq, seq_dcap, seq_dpos, seq_regen, seq_lpos;
    int32_t r0, r1, r2;      // the frame's repeat offsets
    int32_t tail;            // 1: the frame lies in global memory -- the section being decoded is copied to `stage` when it fits there
    int32_t stage_cap;       // bytes at stage
    uint8_t* stage;          // 16-byte aligned
    // walker: ops == nullptr is the decoder proper.  Counts are in units (ops, records, bytes); *_n is where the next one goes.
    ZstdOp* ops;
    uint64_t* recs;
    uint8_t* lits;
    int32_t op_n, op_cap, rec_n, rec_cap, lit_n, lit_cap, stream;
    int32_t defer;           // 1: sequences are left to the lane decoder (jobs, tabs below)
    ZstdSeqJob* jobs;
    uint8_t* tabs;
    int32_t job_n, job_cap, tab_n, tab_cap, frame_jobs, huf_tab;    // huf_tab: where the current Huffman table lies in the plan (treeless blocks use it again)
    ZstdLitJob* litjobs;
    int32_t litjob_n, litjob_cap;
};
static_assert(offsetof(ZstdWork, huf) == 0, "the Huffman table is copied from the start of the work area");
static_assert(offsetof(ZstdWork, ml) == offsetof(ZstdWork, ll) + 4 * (1 << ZSTD_FSE_LOG_MAX) && offsetof(ZstdWork, of) == offsetof(ZstdWork, ml) + 4 * (1 << ZSTD_FSE_LOG_MAX),
              "a job's tables are copied in one piece: ll, ml, of");

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
// The same loop through LDS-typed pointers (round 4): as generic pointers every table read and every refill of the bits was a FLAT
// load, and streams that did not fit the stage were read from GLOBAL memory bit by bit (the natural family: 21 KiB of coded
// literals per block against an 8 KiB stage -- 740 cycles a round).  Here each of the four lanes keeps 64 bits of its stream in a
// register pair, refilled with one 8-byte LDS read about every fifth symbol, and reads the table through an address-space-3
// pointer.  The streams are seen through WINDOWS: lane k's LDS bytes at win + k * wstride are bytes [a_k, a_k + wlen_k) of its stream.
// Streams that lie in LDS whole are one window each (a = 0); streams in global memory travel through the stage a quarter of
// it at a time, from their ends downwards (they are read backwards), and the loop returns 1 when a lane needs bytes below its
// window -- nothing of that symbol is consumed -- for the caller to move the windows and call again.  Checks and results as below.
struct ZstdHuf4 {
    LV<int> size, count, obase, top, state, n, lo, a;      // per lane (lanes 0 .. 3 are the streams)
    LV<uint32_t> c0, c1;                                   // bits [lo, lo + 64) of the lane's stream
    bool primed = false;
};
template <class BP, class HP>
CIMG_DEV int zstd_huf4_run(ZstdHuf4& h, BP win, int wstride, uint8_t* out, int per, int last, HP huf, int log)
{
    const int mask = (1 << log) - 1;
    LV<bool> starve;
    FOR_LANES(l) { starve[l] = false; }
    // nb bits below `top` of the lane's stream (top moves down by nb); bits below position 0 read as zero.  `starve`: the bytes
    // lie below the lane's window -- nothing is consumed
    auto take = [&](int l, int nb) -> uint32_t {
        const int t = h.top[l] - nb;
        if (nb <= 0) { h.top[l] = t; return 0u; }
        const BP sp = win + (l & 3) * wstride - h.a[l];    // (sp[i] = byte i of the stream, for i inside the window)
        if (h.size[l] >= 8 && t >= 0) {
            if (h.lo[l] < 0 || t < h.lo[l]) {
                const int b = ((t + nb + 7) >> 3) - 8;
                const int b0 = b > 0 ? b : 0;
                if (b0 < h.a[l]) { starve[l] = true; return 0u; }
                h.lo[l] = 8 * b0;
                // eight bytes from any address: three aligned dwords and two funnel shifts (an unaligned 8-byte read of LDS is
                // taken apart into byte reads by the compiler)
                // (the host build reads exactly eight bytes: its buffers end where the streams end)
#ifdef CIMG_EMULATE
                uint64_t c;
                __builtin_memcpy(&c, sp + b0, 8);
                h.c0[l] = (uint32_t)c; h.c1[l] = (uint32_t)(c >> 32);
#else
                const BP q = sp + b0;
                const uint32_t sh = (uint32_t)((uintptr_t)q & 3u);
                using WP = typename std::conditional<std::is_same<BP, const uint8_t*>::value, const uint32_t*, cimg_lds_cu32p>::type;
                const WP wp = (WP)(q - sh);
                const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2];
                h.c0[l] = alignbyte(w1, w0, sh); h.c1[l] = alignbyte(w2, w1, sh);
#endif
            }
            h.top[l] = t;
            const uint64_t cc = ((uint64_t)h.c1[l] << 32) | h.c0[l];
            return (uint32_t)(cc >> (t - h.lo[l])) & (uint32_t)((1u << nb) - 1);
        }
        // (the rare path, byte by byte: streams shorter than eight bytes, and the last reads of a stream)
        if (h.a[l] > 0) { starve[l] = true; return 0u; }     // (only reached near the start of a stream: its window must begin there)
        h.top[l] = t;
        uint64_t acc = 0;
        int bit = t, miss = 0;
        if (bit < 0) { miss = -bit; bit = 0; }
        if (miss >= nb) return 0u;
        for (int q = 0; q < 5; q++) {
            const int bq = (bit >> 3) + q;
            const uint64_t v = (bq >= 0 && bq < h.size[l]) ? (uint64_t)sp[bq] : 0;
            acc |= v << (8 * q);
        }
        return (uint32_t)((acc >> (bit & 7)) & ((1ull << (nb - miss)) - 1)) << miss;
    };
    LV<bool> act, bad;
    FOR_LANES(l) { act[l] = l < 4; }
    if (!h.primed) {
        FOR_LANES(l) { if (act[l]) h.state[l] = (int)take(l, log); }
        if (ballot(starve)) return 1;                        // (cannot happen: the first windows end at the ends of the streams)
        h.primed = true;
    }
    for (int guard = 0; guard <= (per > last ? per : last) + 1; ++guard) {
        LV<bool> go;
        FOR_LANES(l) { go[l] = act[l] & (h.top[l] > -log); }
        if (!ballot(go)) break;
        FOR_LANES(l) { bad[l] = go[l] & (h.n[l] >= h.count[l]); }
        if (ballot(bad)) return ERR_DATA;
        FOR_LANES_W(l) {
            if (go[l]) {
                const uint32_t e = huf[h.state[l] & mask];
                const int nb = (int)(e >> 8);
                const uint32_t bits = take(l, nb);
                if (!starve[l]) {
                    out[h.obase[l] + h.n[l]] = (uint8_t)e;
                    h.state[l] = ((h.state[l] << nb) + (int)bits) & mask;
                    h.n[l] += 1;
                }
            }
        }
        if (ballot(starve)) return 1;
    }
    FOR_LANES(l) { bad[l] = act[l] & ((h.top[l] != -log) | (h.n[l] != h.count[l])); }
    return ballot(bad) ? ERR_DATA : 0;
}

CIMG_DEV void zstd_stage(uint8_t* dst, const uint8_t* src, int n);

// the four streams of a literals section at ls (sizes z0 .. z3).  in_lds: they lie in LDS (the stage, or a frame staged whole);
// else they are brought through w->stage, a quarter of it per stream and window
template <class HP>
CIMG_DEV int zstd_huf_stream4_lds(const uint8_t* ls, bool in_lds, int z0, int z1, int z2, int z3, uint8_t* out, int per, int last, HP huf, int log, const ZstdWork* w)
{
    ZstdHuf4 h;
    LV<int> base;
    LV<bool> bad;
    FOR_LANES(l) {
        const int k = l & 3;
        base[l] = k == 0 ? 0 : k == 1 ? z0 : k == 2 ? z0 + z1 : z0 + z1 + z2;
        h.size[l] = k == 0 ? z0 : k == 1 ? z1 : k == 2 ? z2 : z3;
        h.count[l] = k == 3 ? last : per;
        h.obase[l] = k * per;
        h.n[l] = 0; h.lo[l] = -1; h.c0[l] = 0; h.c1[l] = 0; h.a[l] = 0; h.state[l] = 0;
        const int lastb = h.size[l] >= 1 ? (int)ls[base[l] + h.size[l] - 1] : 0;
        bad[l] = (l < 4) & (lastb == 0);
        h.top[l] = h.size[l] * 8 - (8 - zstd_highbit((uint32_t)(lastb | 1)));     // (| 1: keep the arithmetic defined for a bad stream)
    }
    if (ballot(bad)) return ERR_DATA;
    if (in_lds) {
        // one window per stream: stream k begins at ls + base_k, i.e. "stride" is not constant -- the four windows are given one
        // after the other by shifting a_k instead: byte i of stream k lies at ls + base_k + i = win + k * 0 - (-(base_k)) + i
        FOR_LANES(l) { h.a[l] = -base[l]; }
        // (a negative a never starves: b0 >= 0 > a)
        return zstd_huf4_run(h, (cimg_lds_cu8p)CIMG_AS_LDS_CU8(ls), 0, out, per, last, huf, log);
    }
    const int W = (w->stage_cap / 4) & ~15;                  // bytes of stage per stream
    if (W < 64) return -1000;                                // (no stage to speak of: the caller takes the generic loop)
    for (int round = 0; round < 4096; ++round) {
        // windows: stream k's bytes [a_k, b_k), b_k just above the byte its reader stands in, at stage + k * W
        for (int k = 0; k < 4; ++k) {
            const int topk = readlane(h.top, k), sizek = readlane(h.size, k), basek = readlane(base, k);
            int bk = ((topk > 0 ? topk : 0) + 7) / 8 + 1;
            if (bk > sizek) bk = sizek;
            const int ak = bk > W ? bk - W : 0;
            zstd_stage(w->stage + k * W, ls + basek + ak, bk - ak);
            FOR_LANES(l) { if (l == k) { h.a[l] = ak; h.lo[l] = -1; } }
        }
        const int rc = zstd_huf4_run(h, (cimg_lds_cu8p)CIMG_AS_LDS_CU8(w->stage), W, out, per, last, huf, log);
        if (rc != 1) return rc;
    }
    return ERR_DATA;
}

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

// ---- up to 64 decoded sequences, executed by the wave (round 4) ------------------------------------------------------------
// Until round 3 every sequence was decoded AND executed in turn: two wave-wide byte copies (literals, match) per sequence, each a
// dependent LDS round trip with masked byte stores -- half of a sequence's 2300 cycles, for copies of a dozen bytes (libzstd at
// level 22 on shuffled float planes: 1856 sequences per 32 KiB block, 5.6 literal and 12 match bytes each).  Now the scalar
// entropy loop only fills lanes -- lane k holds (literal length, match length, offset) of sequence k, repeat offsets resolved --
// and this function executes the batch lane-parallel:
//   * output and literal positions by two prefix sums; all bounds checks at once;
//   * literals: every lane fetches the first 16 bytes of its run (all loads before any store: a run's destination may overlap
//     the unread literals of the runs in front of it -- they sit at the end of the output area, zstd_block), runs longer than that
//     are finished one after the other in sequence order by the whole wave, then every lane stores its first bytes;
//   * matches, in rounds: with F = the destination of the first match not yet done, every match of at most 16 bytes that does
//     not overlap itself (or is a byte fill, offset 1) and whose source ends in front of F -- or lies in its own literals -- reads
//     only finished output: one lane per match, all of them in one step; when only the first pending match is left to do (long,
//     self-overlapping, or next to unreadable memory) the whole wave copies it the old way.
// Same checks, same results as the serial loop, also on damaged frames (tests/emu runs both on the same streams).
CIMG_DEV u128 zstd_fetch16(const uint8_t* p)
{
    const uintptr_t a = (uintptr_t)p & ~(uintptr_t)3;
    const uint32_t sh = (uint32_t)((uintptr_t)p & 3u);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(a);
    const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
    u128 r;
    r.x = alignbyte(q1, q0, sh); r.y = alignbyte(q2, q1, sh); r.z = alignbyte(q3, q2, sh); r.w = alignbyte(q4, q3, sh);
    return r;
}
// the first n (0 .. 16) bytes of v to d: whole dwords, then the last 0 .. 3 bytes
CIMG_DEV void zstd_store_upto16(uint8_t* d, const u128& v, int n)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    CIMG_UNROLL
    for (int j = 0; j < 4; j++) { if (n >= 4 * j + 4) lds_st32u(d + 4 * j, w[j]); }
    const int t = n > 0 ? (n > 16 ? 16 : n) & ~3 : 0;
    const uint32_t last = t < 16 ? w[(t >> 2) & 3] : 0;
    CIMG_UNROLL
    for (int k = 0; k < 3; k++) { if (n > t + k && t + k < 16) d[t + k] = (uint8_t)(last >> (8 * k)); }
}

// the byte movers on a typed pointer (DP = cimg_lds_u8p in the kernel: ds_* instructions instead of flat ones)
template <class DP> CIMG_DEV u128 zstd_fetch16_t(DP p)
{
    const uint32_t sh = (uint32_t)((uintptr_t)p & 3u);
    using WP = typename std::conditional<std::is_same<DP, uint8_t*>::value, const uint32_t*, cimg_lds_cu32p>::type;
    const WP q = (WP)(p - sh);
    const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
    u128 r;
    r.x = alignbyte(q1, q0, sh); r.y = alignbyte(q2, q1, sh); r.z = alignbyte(q3, q2, sh); r.w = alignbyte(q4, q3, sh);
    return r;
}
template <class DP> CIMG_DEV void zstd_store_upto16_t(DP d, const u128& v, int n)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    CIMG_UNROLL
    for (int j = 0; j < 4; j++) { if (n >= 4 * j + 4) __builtin_memcpy(d + 4 * j, &w[j], 4); }
    const int t = n > 0 ? (n > 16 ? 16 : n) & ~3 : 0;
    const uint32_t last = t < 16 ? w[(t >> 2) & 3] : 0;
    CIMG_UNROLL
    for (int k = 0; k < 3; k++) { if (n > t + k && t + k < 16) d[t + k] = (uint8_t)(last >> (8 * k)); }
}
template <class DP> CIMG_DEV void zstd_copy_t(DP dst, DP src, int n)
{
    for (int k0 = 0; k0 < n; k0 += 64) { FOR_LANES_W(l) { if (k0 + l < n) dst[k0 + l] = src[k0 + l]; } }
}
template <class DP> CIMG_DEV void zstd_match_t(DP dst, int offset, int n)
{
    if (offset >= 64 || offset >= n) {
        for (int k0 = 0; k0 < n; k0 += 64) { FOR_LANES_W(l) { if (k0 + l < n) dst[k0 + l] = dst[k0 + l - offset]; } }
        return;
    }
    const DP pat = dst - offset;
    for (int k0 = 0; k0 < n; k0 += 64) { FOR_LANES_W(l) { if (k0 + l < n) dst[k0 + l] = pat[(k0 + l) % offset]; } }
}

#ifdef CIMG_EMULATE
extern long g_emu_zx_batches, g_emu_zx_rounds, g_emu_zx_par, g_emu_zx_serial, g_emu_zx_longlit;   // test-side statistics only
#endif
template <class DP>
CIMG_DEV int zstd_execute_batch(DP dst, int dcap, int* dpos_io, DP lit, int regen, int* lpos_io, int nb,
                                const LV<int>& vll, const LV<int>& vml, const LV<int>& vof, const uint8_t* mlo_in, const uint8_t* mhi_in)
{
    const int dpos = *dpos_io, lpos = *lpos_io;
    LV<int> len, opos, lsum, ll, ml, of;
    LV<bool> act;
    FOR_LANES(l) {
        act[l] = l < nb;
        ll[l] = act[l] ? vll[l] : 0; ml[l] = act[l] ? vml[l] : 0; of[l] = act[l] ? vof[l] : 1;
        len[l] = ll[l] + ml[l];
    }
    // (a length that would overflow the sums below cannot belong to a frame that fits: refused before it is added up)
    LV<bool> bad;
    FOR_LANES(l) { bad[l] = act[l] & ((ll[l] < 0) | (ml[l] < 0) | (ll[l] > regen) | (ml[l] > dcap)); }
    if (ballot(bad)) return ERR_DATA;
    int acc, ltot;
    wave_exscan(len, opos, acc);
    wave_exscan(ll, lsum, ltot);
    if (ltot > regen - lpos || acc > dcap - dpos) return ERR_DATA;
    LV<int> D, M;
    FOR_LANES(l) {
        D[l] = dpos + opos[l];                               // where the literals of the sequence go
        M[l] = D[l] + ll[l];                                 // where its match goes
        bad[l] = act[l] & ((of[l] <= 0) | (of[l] > M[l]));   // (the serial loop: offset > dpos once the literals are out)
    }
    if (ballot(bad)) return ERR_DATA;
    const uint8_t* const mlo = reinterpret_cast<const uint8_t*>(uni64((int64_t)reinterpret_cast<uintptr_t>(mlo_in)));
    const uint8_t* const mhi = reinterpret_cast<const uint8_t*>(uni64((int64_t)reinterpret_cast<uintptr_t>(mhi_in)));
#ifdef CIMG_ABL_ZSTD_NO_EXEC      /* timing experiment only: nothing is copied */
    *dpos_io = dpos + acc; *lpos_io = lpos + ltot;
    if (acc >= 0) return 0;
#endif
#if defined(CIMG_ZSTD_SHARED_EXECUTOR) || defined(CIMG_EMULATE)
    // NOT the device's default (measured: 6.8 against 5.9 ms on the tiled family, 13.7 against 14.0 on the natural one; build with
    // -DCIMG_ZSTD_SHARED_EXECUTOR); the emulator's tests and the fuzzer run every other batch through it.
    // The batch executor of the LZ4 decoder (decode_kernel.h: lz_batch_execute -- literal runs lane-parallel, matches in dependency
    // rounds by token ranges, sixteen matches of up to 64 bytes a step) on this batch: offsets relative to the output's first
    // ALIGNED dword.  (Without a readable range around the output -- one mode of the host tests -- the byte-exact form below.)
    {
        const uintptr_t gd = (uintptr_t)(const uint8_t*)dst;
        const int mis = (int)(gd & 3u);
#if defined(CIMG_EMULATE) && !defined(CIMG_ZSTD_SHARED_EXECUTOR)
        static long emu_turn = 0;
        const bool shared_turn = (emu_turn++ & 1) != 0;
#else
        const bool shared_turn = true;
#endif
        if (shared_turn && mlo != nullptr && gd - mis >= (uintptr_t)mlo && (uintptr_t)mhi >= gd - mis + 8) {
            const int64_t span = (int64_t)((uintptr_t)mhi - (gd - mis));
            const int clampmax = (int)(((span < (1 << 18) ? span : (1 << 18) - 4) - 4) & ~3ll);
            const int lit0 = (int)((uintptr_t)(const uint8_t*)lit - gd) + mis + lpos;
            LV<int> lsrc;
            FOR_LANES(l) { lsrc[l] = lit0 + lsum[l]; }
            int op = dpos + mis, cut = -1;
            const int rc = lz_batch_execute<DP, true>(dst - mis, mis, dcap + mis, clampmax, op, nb, ll, ml, of, lsrc, cut);
            if (rc < 0 || cut >= 0) return ERR_DATA;
            *dpos_io = dpos + acc;
            *lpos_io = lpos + ltot;
            return 0;
        }
    }
#endif
    // ---- literals
    {
        LV<u128> first;
        LV<bool> fast, slow;
        FOR_LANES(l) {
            const DP sp = lit + lpos + lsum[l];
            const uintptr_t a = (uintptr_t)(const uint8_t*)(sp) & ~(uintptr_t)3;
            fast[l] = act[l] & (ll[l] > 0) & (mlo != nullptr) & (a >= (uintptr_t)mlo) & (a + 20 <= (uintptr_t)mhi);
            slow[l] = act[l] & (ll[l] > 0) & !fast[l];
            if (mlo != nullptr) first[l] = zstd_fetch16_t<DP>(fast[l] ? sp : dst);          // (a lane that is not fast fetches from the start of the output and drops it)
            else { first[l].x = 0; first[l].y = 0; first[l].z = 0; first[l].w = 0; }
        }
        // runs with more than 16 bytes (and runs next to unreadable memory: whole) in sequence order, by the whole wave
        LV<bool> longer;
        FOR_LANES(l) { longer[l] = slow[l] | (fast[l] & (ll[l] > 16)); }
        uint64_t todo = ballot(longer);
        const uint64_t whole = ballot(slow);
#ifdef CIMG_EMULATE
        g_emu_zx_batches++; g_emu_zx_longlit += popc64(todo);
#endif
        while (todo) {
            const int t = ctz64(todo);
            todo &= todo - 1;
            const int n = readlane(ll, t), d = readlane(D, t), sp = lpos + readlane(lsum, t);
            const int skip = ((whole >> t) & 1) ? 0 : 16;
            zstd_copy_t<DP>(dst + d + skip, lit + sp + skip, n - skip);
        }
        FOR_LANES_W(l) { if (fast[l]) zstd_store_upto16_t<DP>(dst + D[l], first[l], ll[l] < 16 ? ll[l] : 16); }
    }
    // ---- matches
    uint64_t pending;
    {
        LV<bool> has;
        FOR_LANES(l) { has[l] = act[l] & (ml[l] > 0); }
        pending = ballot(has);
    }
    // Which sequences of the batch does a match's source touch?  Sequence j's output is [D_j, D_j+1): two binary searches over D
    // (crossbar gathers) give every match the range lo .. hi of sequences its source bytes lie in; it is ready when the MATCHES of
    // those below itself are done (literals are all in place).  (Round 4's first form only looked at the first pending match's
    // destination: level-22 streams then took 35 rounds a batch.)
    LV<int> lo_i, hi_i;
    {
        LV<int> Dx, ca, cb, xa, xb;
        FOR_LANES(l) {
            Dx[l] = act[l] ? D[l] : 0x7FFFFFFF;
            const int S = M[l] - of[l];
            const int need = of[l] == 1 ? 1 : (of[l] < ml[l] ? of[l] : ml[l]);
            xa[l] = S; xb[l] = S + need - 1;
            ca[l] = 0; cb[l] = 0;
        }
        CIMG_UNROLL
        for (int k = 5; k >= 0; --k) {
            LV<int> ta, tb, pa, pb;
            FOR_LANES(l) { ta[l] = ca[l] + (1 << k) - 1; tb[l] = cb[l] + (1 << k) - 1; }
            lane_gather(Dx, ta, pa);
            lane_gather(Dx, tb, pb);
            FOR_LANES(l) {
                if (ta[l] < 64 && pa[l] <= xa[l]) ca[l] += 1 << k;
                if (tb[l] < 64 && pb[l] <= xb[l]) cb[l] += 1 << k;
            }
        }
        FOR_LANES(l) {
            lo_i[l] = ca[l] > 0 ? ca[l] - 1 : 0;
            hi_i[l] = imin(cb[l] - 1, l - 1);
        }
    }
    for (int round = 0; round < 128 && pending; ++round) {
        const int t0 = ctz64(pending);
        const int F = readlane(M, t0);
        LV<bool> ready;
        FOR_LANES(l) {
            const int S = M[l] - of[l];
            const bool fill = of[l] == 1;
            const uintptr_t a = (uintptr_t)(const uint8_t*)(dst + S) & ~(uintptr_t)3;
            const bool readable = (mlo != nullptr) & (a >= (uintptr_t)mlo) & (a + 20 <= (uintptr_t)mhi);
            const bool apart = (of[l] >= ml[l]) | fill;                                    // does not read what it writes (a fill reads one byte)
            const uint64_t upto = hi_i[l] >= 63 ? ~0ull : ((1ull << (hi_i[l] + 1)) - 1);
            const uint64_t from = lo_i[l] >= 64 ? 0ull : (~0ull << lo_i[l]);
            const uint64_t deps = hi_i[l] >= lo_i[l] ? (upto & from) : 0ull;
            const bool done_src = (deps & pending) == 0;                                   // every match its source touches is done
            ready[l] = ((pending >> l) & 1) & (ml[l] <= 16) & apart & done_src & readable;
        }
        const uint64_t rmask = ballot(ready);
#ifdef CIMG_EMULATE
        g_emu_zx_rounds++; if (rmask) g_emu_zx_par += popc64(rmask); else g_emu_zx_serial++;
#endif
        if (rmask) {
            LV<u128> v;
            FOR_LANES(l) {
                const int S = M[l] - of[l];
                v[l] = zstd_fetch16_t<DP>(ready[l] ? dst + S : dst);                              // (rmask != 0: there is a readable range)
                if (of[l] == 1) { const uint32_t f = (v[l].x & 0xFF) * 0x01010101u; v[l].x = f; v[l].y = f; v[l].z = f; v[l].w = f; }
            }
            FOR_LANES_W(l) { if (ready[l]) zstd_store_upto16_t<DP>(dst + M[l], v[l], ml[l]); }
            pending &= ~rmask;
        } else {
            // the first pending match: all output in front of it is final
            zstd_match_t<DP>(dst + F, readlane(of, t0), readlane(ml, t0));
            pending &= pending - 1;
        }
    }
    if (pending) return ERR_FAILURE;                         // (cannot happen: every round retires at least one match)
    *dpos_io = dpos + acc;
    *lpos_io = lpos + ltot;
    return 0;
}

// ---- what 64 sequences say -----------------------------------------------------------------------------------------------
// Lane k holds sequence k's three table entries (symbol | state bits << 8 | state base << 16) and the bit position just above its
// extra bits (offset's first = highest, then match length's, then literal length's).  Lengths and offset values are one lane's
// arithmetic each; what stays serial is the history of repeat offsets (RFC 8878 3.1.1.5), a scalar walk over the batch.
CIMG_DEV void zstd_resolve_batch(const uint8_t* bs_in, int bl_in, int nb_in, const LV<uint32_t>& vpl, const LV<uint32_t>& vpo, const LV<uint32_t>& vpm,
                                         const LV<int>& vtop, LV<int>& vll, LV<int>& vml, LV<int>& vof, int& r0_io, int& r1_io, int& r2_io)
{
    const uint8_t* const bs = reinterpret_cast<const uint8_t*>(uni64((int64_t)reinterpret_cast<uintptr_t>(bs_in)));
    const int bl = uni(bl_in), nb = uni(nb_in);
    int r0 = uni(r0_io), r1 = uni(r1_io), r2 = uni(r2_io);
    LV<int> ov;
    LV<bool> nolit, rep;
    FOR_LANES(l) {
        const bool act = l < nb;
        const int ls = act ? (int)(vpl[l] & 0xFF) : 0, os = act ? (int)(vpo[l] & 31) : 2, ms = act ? (int)(vpm[l] & 0xFF) : 0;
        const int xc = zstd_ll_bits(ls), xb = zstd_ml_bits(ms);
        const int top = vtop[l];
        const uint32_t xo = act ? zstd_rbits_lane(bs, bl, top, os) : 0u;
        const uint32_t xm = act ? zstd_rbits_lane(bs, bl, top - os, xb) : 0u;
        const uint32_t xl = act ? zstd_rbits_lane(bs, bl, top - os - xb, xc) : 0u;
        vll[l] = zstd_ll_base(ls) + (int)xl;
        vml[l] = act ? zstd_ml_base(ms) + (int)xm : 0;
        ov[l] = (int)((1u << os) + xo);                         // (a lane beyond the batch: 4, i.e. nothing for the walk below)
        vof[l] = ov[l] - 3;
        nolit[l] = act & (vll[l] == 0);
        rep[l] = act & ((uint32_t)ov[l] <= 3u);
    }
    const uint64_t zl = ballot(nolit), reps = ballot(rep);
    // the history: a new offset pushes the three down; a repeat code picks one of them (one further when the sequence has no
    // literals; the fourth choice is the first offset minus one) and moves it to the front
    for (int k = 0; k < nb; ++k) {
        if (!((reps >> k) & 1)) { r2 = r1; r1 = r0; r0 = uni(readlane(vof, k)); continue; }
        const int idx = uni(readlane(ov, k)) + (int)((zl >> k) & 1);
        int offset;
        if (idx == 1) offset = r0;
        else if (idx == 2) { offset = r1; r1 = r0; r0 = offset; }
        else { offset = idx == 3 ? r2 : r0 - 1; r2 = r1; r1 = r0; r0 = offset; }
        writelane(vof, k, offset);
    }
    r0_io = r0; r1_io = r1; r2_io = r2;
}

// ---- the sequences section of one block (round 4: out of line, tables and bit stream through LDS-typed pointers) ------------
// Until round 3 this loop was part of zstd_block, inlined into a kernel of ten thousand instructions: 197 spilled scalars, and --
// because the work area reaches it through a pointer the compiler cannot trace to the __shared__ array -- every table entry and
// every refill of the bit container a FLAT load (several hundred cycles and a vmcnt wait, three to four per sequence): 1950 cycles a
// sequence before a single byte was copied.  Here: a function of its own (its registers are its own), the three tables behind
// address-space-3 pointers (BP = cimg_lds_cu8p: the bit stream too, which is nearly always in the stage), the next sequence's
// table entries requested as soon as the states are known, and the sequences handed to zstd_execute_batch 64 at a time.
// bits [bit, bit + n) of a stream, byte by byte, bytes outside [0, size) read as zero; bit may be negative (the reader's rare path:
// streams shorter than eight bytes, and the last reads of a stream, which may reach below its first bit)
template <class BP> CIMG_DEV_OUTLINE uint32_t zstd_bits_slow(BP src_in, int size_in, int bit_in, int n_in)
{
    const BP src = (BP)reinterpret_cast<const uint8_t*>(uni64((int64_t)reinterpret_cast<uintptr_t>((const uint8_t*)src_in)));
    const int size = uni(size_in), n = uni(n_in);
    int bit = uni(bit_in);
    int miss = 0;
    if (bit < 0) { miss = -bit; bit = 0; }
    if (miss >= n) return 0;
    uint64_t acc = 0;
    const int b0 = bit >> 3;
    for (int k = 0; k < 5; k++) {
        const int b = b0 + k;
        const uint64_t v = (b >= 0 && b < size) ? (uint64_t)uni((uint32_t)src[b]) : 0;
        acc |= v << (8 * k);
    }
    return (uint32_t)((acc >> (bit & 7)) & ((1ull << (n - miss)) - 1)) << miss;
}
template <class BP> struct ZstdBits {                     // the backward bit reader (ZstdBack) on a typed pointer, 32-bit positions
    BP src;
    int size, off, lo;                                    // cont holds bits [lo, lo + 64); lo < 0: nothing loaded yet
    uint64_t cont;
    CIMG_DEV void init(BP s, int n, int o) { src = s; size = n; off = o; lo = -1; cont = 0; }
    CIMG_DEV uint32_t get(int n)
    {
        if (n <= 0) return 0;
        const int t = off - n;
        off = t;
        if (size >= 8 && t >= 0) {
            if (lo < 0 || t < lo) {
                const int b = ((t + n + 7) >> 3) - 8;
                lo = 8 * (b > 0 ? b : 0);
                uint64_t c;
                __builtin_memcpy(&c, src + (lo >> 3), 8);
                cont = (uint64_t)uni((uint32_t)c) | ((uint64_t)uni((uint32_t)(c >> 32)) << 32);
            }
            return (uint32_t)(cont >> ((t - lo) & 63)) & (uint32_t)((1ull << n) - 1);      // (t - lo == 64 only with n == 0: a byte-aligned reader that stands still)
        }
        return uni(zstd_bits_slow<BP>(src, size, t, n));   // (uni: what a real call returns arrives in a vector register)
    }
    // the same without the range checks: for callers that know size >= 8 and that off stays >= 0 (zstd_sequences: far from the
    // start of the stream); n may be 0
    CIMG_DEV uint32_t getf(int n)
    {
        const int t = off - n;
        if (t < lo || lo < 0) {
            lo = 8 * (((off + 7) >> 3) - 8);
            uint64_t c;
            __builtin_memcpy(&c, src + (lo >> 3), 8);
            cont = (uint64_t)uni((uint32_t)c) | ((uint64_t)uni((uint32_t)(c >> 32)) << 32);
        }
        off = t;
        return (uint32_t)(cont >> ((t - lo) & 63)) & (uint32_t)((1ull << n) - 1);      // (t - lo == 64 only with n == 0: a byte-aligned reader that stands still)
    }
};
// length codes above the directly coded ones (literal lengths from 16, match lengths from 35): (base << 8) | extra bits
CIMG_DEV_OUTLINE uint32_t zstd_ll_code_slow(int c_in) { const int c = uni(c_in); return ((uint32_t)zstd_ll_base(c) << 8) | (uint32_t)zstd_ll_bits(c); }
CIMG_DEV_OUTLINE uint32_t zstd_ml_code_slow(int c_in) { const int c = uni(c_in); return ((uint32_t)zstd_ml_base(c) << 8) | (uint32_t)zstd_ml_bits(c); }

CIMG_DEV int32_t zstd_field(const int32_t* p) { return (int32_t)uni((uint32_t)*p); }

template <class BP, bool WALK = false>
CIMG_DEV_OUTLINE int zstd_sequences(ZstdWork* w_in)
{
    ZstdWork* const w = reinterpret_cast<ZstdWork*>(uni64((int64_t)reinterpret_cast<uintptr_t>(w_in)));
    // (the tables: LDS on the device whenever this instance is the LDS one; the generic instance reads them as it finds them)
    constexpr bool in_lds = !std::is_same<BP, const uint8_t*>::value;
    using TP = typename std::conditional<in_lds, cimg_lds_cu32p, const uint32_t*>::type;
    const TP tll = (TP)reinterpret_cast<const uint32_t*>(w->ll), tof = (TP)reinterpret_cast<const uint32_t*>(w->of), tml = (TP)reinterpret_cast<const uint32_t*>(w->ml);
    const int ll_log = zstd_field(&w->ll_log), of_log = zstd_field(&w->of_log), ml_log = zstd_field(&w->ml_log);
    const int nseq = zstd_field(&w->seq_nseq), bl = zstd_field(&w->seq_bl);
    const int dcap = zstd_field(&w->seq_dcap), regen = zstd_field(&w->seq_regen);
    int dpos = zstd_field(&w->seq_dpos), lpos = 0;
    const uint8_t* const bs_g = reinterpret_cast<const uint8_t*>(uni64((int64_t)reinterpret_cast<uintptr_t>(w->seq_bs)));
    uint8_t* const dst = reinterpret_cast<uint8_t*>(uni64((int64_t)reinterpret_cast<uintptr_t>(w->seq_dst)));
    const uint8_t* const lit = reinterpret_cast<const uint8_t*>(uni64((int64_t)reinterpret_cast<uintptr_t>(w->seq_lit)));
    int r0 = zstd_field(&w->r0), r1 = zstd_field(&w->r1), r2 = zstd_field(&w->r2);
    int rec_n = WALK ? zstd_field(&w->rec_n) : 0;
    uint64_t* const recs = WALK ? reinterpret_cast<uint64_t*>(uni64((int64_t)reinterpret_cast<uintptr_t>(w->recs))) : nullptr;
    const BP bs = (BP)bs_g;
    ZstdBits<BP> br;
    br.init(bs, bl, bl * 8 - (8 - zstd_highbit(uni((uint32_t)bs[bl - 1]))));
    const int llm = (1 << ll_log) - 1, ofm = (1 << of_log) - 1, mlm = (1 << ml_log) - 1;
    int sl = (int)br.get(ll_log), so = (int)br.get(of_log), sm = (int)br.get(ml_log);
    // The loop below is the SERIAL part of a sequence and nothing else: the three states walk down the bit stream.  What a sequence
    // says -- its lengths and its offset -- is read afterwards, 64 sequences at a time, one lane each (zstd_resolve_batch): lane k
    // keeps sequence k's three table entries and the bit position just above its extra bits.
    LV<uint32_t> vpl, vpo, vpm;
    LV<int> vtop;
    FOR_LANES(l) { vpl[l] = 0; vpo[l] = 0; vpm[l] = 0; vtop[l] = 0; }
    // (the entries travel from one iteration to the next in VECTOR registers -- that is where a load lands -- and are made scalar at
    // the top of the iteration that uses them: carried as scalars they would have to be waited for in the iteration that requests
    // them; carried as vectors WITHOUT the uni() up here the compiler takes the whole loop for divergent control flow)
    uint32_t pl_v = tll[sl & llm], po_v = tof[so & ofm], pm_v = tml[sm & mlm];
    int err = 0;
    // One sequence.  FAST: the reader stands at least 96 bits above the start of a stream of at least eight bytes -- more than a
    // sequence can consume (31 + 16 + 16 extra bits, 26 bits of state) -- so no read needs a range check, and the symbols are not
    // tested either (the tables are built from validated headers: nothing above the last code of its alphabet gets in).
    auto step = [&](auto fast_tag, int i) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const uint32_t pl = uni(pl_v), po = uni(po_v), pm = uni(pm_v);
        const int ls = (int)(pl & 0xFF), lnb = (int)((pl >> 8) & 0xFF), lbase = (int)(pl >> 16);
        const int os = (int)(po & 0xFF), onb = (int)((po >> 8) & 0xFF), obase = (int)(po >> 16);
        const int ms = (int)(pm & 0xFF), mnb = (int)((pm >> 8) & 0xFF), mbase = (int)(pm >> 16);
        if (!FAST && (os > 31 || ls > 35 || ms > 52)) { err = ERR_DATA; return; }
        // the extra bits of offset, match length and literal length lie one behind the other in the stream: stepped over here
        // (most sequences of an image have a literal length below 16 and a match length below 32: coded directly, no extra bits)
        int extra = os & 31;
        if (ms >= 32) extra += zstd_ml_bits(ms);
        if (ls >= 16) extra += zstd_ll_bits(ls);
        const int k = i & 63;
        writelane(vtop, k, br.off); writelane(vpl, k, pl); writelane(vpo, k, po); writelane(vpm, k, pm);
        br.off -= extra;
        if (i + 1 < nseq) {
            const uint32_t V = FAST ? br.getf(lnb + mnb + onb) : br.get(lnb + mnb + onb);               // <= 9 + 9 + 8 bits
            so = obase + (int)(V & ((1u << onb) - 1));
            sm = mbase + (int)((V >> onb) & ((1u << mnb) - 1));
            sl = lbase + (int)(V >> (onb + mnb));
            // the next sequence's entries: requested now, used at the top of the next iteration
            pl_v = tll[sl & llm]; po_v = tof[so & ofm]; pm_v = tml[sm & mlm];
        }
        if (!FAST && br.off < 0) { err = ERR_DATA; return; }
    };
    for (int i = 0; i < nseq; i++) {
        if (bl >= 8 && br.off >= 96) step(std::true_type{}, i);
        else step(std::false_type{}, i);
        if (err) return err;
        const int k = i & 63;
        if (k == 63 || i + 1 == nseq) {
            LV<int> vll, vml, vof;
            zstd_resolve_batch(bs_g, bl, k + 1, vpl, vpo, vpm, vtop, vll, vml, vof, r0, r1, r2);
            if constexpr (WALK) {
                // the walker: the batch becomes records of the plan (one 8-byte store a lane), after the checks of the executor
                // that need no output -- the replay makes the others
                const int nb = k + 1;
                LV<bool> bad;
                LV<int> len, t0, t1;
                FOR_LANES(l) {
                    const bool act = l < nb;
                    if (!act) { vll[l] = 0; vml[l] = 0; vof[l] = 1; }
                    bad[l] = act & ((vll[l] < 0) | (vml[l] < 0) | (vll[l] > regen) | (vml[l] > dcap) | (vof[l] <= 0) | (vof[l] > dcap));
                    len[l] = bad[l] ? 0 : vll[l] + vml[l];
                    if (bad[l]) vll[l] = 0;
                }
                if (ballot(bad)) return ERR_DATA;
                int acc, ltot;
                wave_exscan(len, t0, acc);
                wave_exscan(vll, t1, ltot);
                if (ltot > regen - lpos || acc > dcap - dpos) return ERR_DATA;
                FOR_LANES_W(l) { if (l < nb) recs[rec_n + l] = zstd_record((uint32_t)vll[l], (uint32_t)vml[l], (uint32_t)vof[l]); }
                rec_n += nb; dpos += acc; lpos += ltot;
            } else {
                // (the output area -- and the literals at its end -- lie in LDS whenever this is the LDS instance)
                using DP = typename std::conditional<in_lds, cimg_lds_u8p, uint8_t*>::type;
                const int rc = zstd_execute_batch<DP>((DP)dst, dcap, &dpos, (DP)const_cast<uint8_t*>(lit), regen, &lpos, k + 1, vll, vml, vof, w->mem_lo, w->mem_hi);
                if (rc < 0) return rc;
            }
        }
    }
    if (br.off != 0) return ERR_DATA;
    FOR_LANES_W(l) { w->seq_lpos = lpos; w->r0 = r0; w->r1 = r1; w->r2 = r2; if (WALK) w->rec_n = rec_n; }
    return dpos;
}

// ---- one compressed block ---------------------------------------------------------------------------------------------
struct ZstdFrameState { int r0, r1, r2; };      // the three repeat offsets

CIMG_DEV bool zstd_walking(const ZstdWork* w) { return uni64((int64_t)reinterpret_cast<uintptr_t>(w->ops)) != 0; }
// the walker's next op (its stream number is filled in here)
CIMG_DEV int zstd_emit(ZstdWork* w, ZstdOp op)
{
    const int n = zstd_field(&w->op_n);
    if (n >= zstd_field(&w->op_cap)) return ZSTD_WALK_OVERFLOW;
    op.stream = (uint32_t)zstd_field(&w->stream);
    ZstdOp* const to = reinterpret_cast<ZstdOp*>(uni64((int64_t)reinterpret_cast<uintptr_t>(w->ops))) + n;
    FOR_LANES_W(l) { if (l == 0) *to = op; }
    FOR_LANES_W(l) { w->op_n = n + 1; }
    return 0;
}

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
    // (a walker has no output: coded literals are decoded into the plan's literal area, raw ones stay where they lie, a run is one byte)
    const bool walk = zstd_walking(w);
    uint64_t lit_word = 0;                                 // walker: what the SEQ op says about the literals
    uint32_t lit_fill = 0;
    uint8_t* litbuf;
    if (walk) {
        const int at = zstd_field(&w->lit_n);
        if (ltype >= 2 && regen > zstd_field(&w->lit_cap) - at) return ZSTD_WALK_OVERFLOW;
        litbuf = reinterpret_cast<uint8_t*>(uni64((int64_t)reinterpret_cast<uintptr_t>(w->lits))) + at;
        if (ltype >= 2) { FOR_LANES_W(l) { w->lit_n = at + ((regen + 15) & ~15); } lit_word = (uint64_t)reinterpret_cast<uintptr_t>(litbuf); }
    } else litbuf = dst + (dcap - regen);
    int pos = hdr;
    const uint8_t* lit = litbuf;
    // A frame in global memory (w->tail): every bit read and every literal run would wait for a load from there, once per
    // sequence.  Raw literals are brought over in one piece; the coded literals, then the sequences, go through w->stage when
    // they fit it (they are read where they lie when not).
    const int stage_cap = w->tail ? w->stage_cap : 0;
    if (ltype == 0) {
        if (pos + regen > size) return ERR_DATA;
        if (walk) lit_word = (uint64_t)reinterpret_cast<uintptr_t>(src + pos);
        else zstd_stage(litbuf, src + pos, regen);         // (also from a frame in LDS: the batch executor reads its literals from the output area)
        pos += regen;
    } else if (ltype == 1) {
        if (pos + 1 > size) return ERR_DATA;
        if (walk) { lit_word = (uint64_t)zstd_u8(src, pos); lit_fill = 1; }
        else zstd_fill(litbuf, zstd_u8(src, pos), regen);
        pos += 1;
    } else {
        if (pos + comp > size) return ERR_DATA;
        const uint8_t* ls = src + pos;
        int lsz = comp;
        bool ls_in_lds = !w->tail;                         // (a frame staged whole)
        if (comp <= stage_cap) { zstd_stage(w->stage, ls, comp); ls = w->stage; ls_in_lds = true; }
        // (a section that does not fit the stage: its HEAD does -- the tree description is at most 129 bytes, and read byte by byte
        // from global memory it is a load latency per byte: 0.2 ms a block, which was most of what a walker did)
        const bool head_only = comp > stage_cap && stage_cap >= 256;
        if (head_only && ltype == 2) zstd_stage(w->stage, ls, stage_cap);
        if (ltype == 2) {
#ifdef CIMG_ABL_ZSTD_NO_TREE     /* timing experiment only: the table is not built (one with the same header length would be) */
            const int t = lsz > 130 ? 64 : 1;
            w->huf_log = 11;
#else
            const int t = head_only ? zstd_huf_read_tree(w->stage, stage_cap, w) : zstd_huf_read_tree(ls, lsz, w);
#endif
            if (t < 0) return t;
            ls += t; lsz -= t;
            if (walk && zstd_field(&w->defer)) {
                // (the lane decoder of literals reads this table from the plan)
                const int tn = zstd_field(&w->tab_n);
                if (tn + ZSTD_HUF_TABLE_BYTES > zstd_field(&w->tab_cap)) return ZSTD_WALK_OVERFLOW;
                uint8_t* const tdst = reinterpret_cast<uint8_t*>(uni64((int64_t)reinterpret_cast<uintptr_t>(w->tabs))) + tn;
                const uint8_t* const tsrc = reinterpret_cast<const uint8_t*>(w->huf);
                for (int i0 = 0; i0 < ZSTD_HUF_TABLE_BYTES; i0 += 1024) { FOR_LANES_W(l) { st128a(tdst + i0 + 16 * l, ld128u(tsrc + i0 + 16 * l)); } }
                FOR_LANES_W(l) { w->huf_tab = tn; w->tab_n = tn + ZSTD_HUF_TABLE_BYTES; }
            }
        } else if (!w->have_huf) return ERR_DATA;
        if (walk && zstd_field(&w->defer)) {
            // the block's literals become a job of the lane decoder: where the streams lie in the CHUNK (not in the stage)
            const int jn = zstd_field(&w->litjob_n);
            if (jn >= zstd_field(&w->litjob_cap)) return ZSTD_WALK_OVERFLOW;
            const int consumed = comp - lsz;                   // the tree description
            ZstdLitJob job;
            memset(&job, 0, sizeof(job));
            job.src = (uint64_t)reinterpret_cast<uintptr_t>(src + pos + consumed); job.lsz = (uint32_t)lsz; job.regen = (uint32_t)regen;
            job.out = (uint32_t)(litbuf - reinterpret_cast<uint8_t*>(uni64((int64_t)reinterpret_cast<uintptr_t>(w->lits))));
            job.tab = (uint32_t)zstd_field(&w->huf_tab); job.log = (uint32_t)zstd_field(&w->huf_log); job.streams = (uint32_t)streams;
            if (streams == 4) {
                if (lsz < 6) return ERR_DATA;
                if (3 * ((regen + 3) / 4) > regen) return ERR_DATA;
            } else if (lsz < 1) return ERR_DATA;
            ZstdLitJob* const jto = reinterpret_cast<ZstdLitJob*>(uni64((int64_t)reinterpret_cast<uintptr_t>(w->litjobs))) + jn;
            FOR_LANES_W(l) { if (l == 0) *jto = job; }
            FOR_LANES_W(l) { w->litjob_n = jn + 1; }
        } else if (streams == 1) {
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
            // (streams in the stage, or in a frame staged whole: LDS on the device -- the loop with typed pointers)
            // (the table -- and the work area -- lie in LDS on the device: mem_lo says "this is the kernel"; the host tests take this
            // path too, with plain pointers)
            const bool ls_lds = ls_in_lds;
            int rc = -1000;
#ifdef CIMG_ABL_ZSTD_NO_HUF      /* timing experiment only: literals are not decoded */
            rc = 0;
#endif
            // (streams in global memory: the windowed form measured 14.8 against 13.3 ms on the natural family -- CIMG_ZSTD_HUF_WINDOWS
            // builds keep it for the next attempt; the lane loop below reads them where they lie)
#ifdef CIMG_ZSTD_HUF_WINDOWS
            const bool typed = w->mem_lo != nullptr;
#else
            const bool typed = w->mem_lo != nullptr && ls_lds;
#endif
            if (typed && rc == -1000)
                rc = zstd_huf_stream4_lds(ls + 6, ls_lds, s1, s2, s3, s4, litbuf, per, regen - 3 * per, (cimg_lds_cu16p)CIMG_AS_LDS_CU16(w->huf), w->huf_log, w);
            if (rc == -1000) rc = zstd_huf_stream4(ls + 6, s1, s2, s3, s4, litbuf, per, regen - 3 * per, w);
            if (rc < 0) return rc;
        }
        pos += comp;
    }
    // ---- sequences section
    if (pos >= size) return ERR_DATA;
    // (of a frame in global memory the head of the section -- sequence count, modes, up to three table descriptions: some 160 bytes
    // at most -- is brought into the stage first: the descriptions are read bit by bit.  hsrc[p - hdel] is byte p of the block.)
    const uint8_t* hsrc = src;
    int hdel = 0, hend = size;
    if (stage_cap >= 512 && size - pos > 0) {
        const int hb = imin(size - pos, stage_cap);
        zstd_stage(w->stage, src + pos, hb);
        hsrc = w->stage; hdel = pos; hend = pos + hb;
    }
    int nseq = zstd_u8(hsrc, pos++ - hdel);
    if (nseq >= 128) {
        if (nseq == 255) { if (pos + 2 > hend) return ERR_DATA; nseq = zstd_u8(hsrc, pos - hdel) + (zstd_u8(hsrc, pos + 1 - hdel) << 8) + 0x7F00; pos += 2; }
        else { if (pos + 1 > hend) return ERR_DATA; nseq = ((nseq - 128) << 8) + zstd_u8(hsrc, pos++ - hdel); }
    }
    int lpos = 0;                                          // literals consumed
    int rec_first = 0;
    if (nseq > 0) {
        if (pos >= hend) return ERR_DATA;
        const int modes = zstd_u8(hsrc, pos++ - hdel);
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
                if (pos >= hend) return ERR_DATA;
                if (zstd_u8(hsrc, pos - hdel) > maxsym) return ERR_DATA;
                zstd_fse_rle(t, zstd_u8(hsrc, pos++ - hdel));
                *lg = 0;
            } else if (mode == 2) {
                int nsym = 0;
                const int h = zstd_fse_read_header(hsrc + (pos - hdel), hend - pos, maxlog, maxsym, w, lg, &nsym);
                if (h < 0) return h;
                const int rc = zstd_fse_build(t, *lg, nsym, w);
                if (rc < 0) return rc;
                pos += h;
            } else if (!(w->have_tables & (1 << k))) return ERR_DATA;
            w->have_tables |= 1 << k;
        }
        const uint8_t* bs = src + pos;
        const int bl = size - pos;
        if (walk && zstd_field(&w->defer)) {
            // the block's sequences become a job of the lane decoder: tables to the plan, the bit stream stays where it lies
            if (bl < 1 || zstd_u8(bs, bl - 1) == 0) return ERR_DATA;
            const int jn = zstd_field(&w->job_n), tn = zstd_field(&w->tab_n), rn = zstd_field(&w->rec_n);
            if (jn >= zstd_field(&w->job_cap) || tn + ZSTD_JOB_TABLE_BYTES > zstd_field(&w->tab_cap) || nseq > zstd_field(&w->rec_cap) - rn) return ZSTD_WALK_OVERFLOW;
            uint8_t* const tdst = reinterpret_cast<uint8_t*>(uni64((int64_t)reinterpret_cast<uintptr_t>(w->tabs))) + tn;
            const uint8_t* const tsrc = reinterpret_cast<const uint8_t*>(w->ll);
            for (int i0 = 0; i0 < ZSTD_JOB_TABLE_BYTES; i0 += 1024) { FOR_LANES_W(l) { st128a(tdst + i0 + 16 * l, ld128u(tsrc + i0 + 16 * l)); } }
            ZstdSeqJob job;
            job.bs = (uint64_t)reinterpret_cast<uintptr_t>(bs); job.bl = (uint32_t)bl; job.nseq = (uint32_t)nseq; job.rec = (uint32_t)rn; job.tab = (uint32_t)tn;
            job.ll_log = (uint8_t)zstd_field(&w->ll_log); job.of_log = (uint8_t)zstd_field(&w->of_log); job.ml_log = (uint8_t)zstd_field(&w->ml_log);
            job.first = zstd_field(&w->frame_jobs) == 0 ? 1 : 0; job.pad = 0;
            ZstdSeqJob* const jto = reinterpret_cast<ZstdSeqJob*>(uni64((int64_t)reinterpret_cast<uintptr_t>(w->jobs))) + jn;
            FOR_LANES_W(l) { if (l == 0) *jto = job; }
            FOR_LANES_W(l) { w->job_n = jn + 1; w->tab_n = tn + ZSTD_JOB_TABLE_BYTES; w->rec_n = rn + nseq; w->frame_jobs = 1; }
            ZstdOp op;
            op.kind = ZOP_SEQ; op.size = (uint32_t)regen; op.ptr = lit_word; op.nseq = (uint32_t)nseq; op.rec = (uint32_t)rn; op.lit_fill = lit_fill; op.stream = 0;
            const int erc = zstd_emit(w, op);
            return erc < 0 ? erc : dpos;                   // (how far the block gets is for the replay to say)
        }
        int bs_lds = w->tail ? 0 : 1;                      // (a frame staged whole lies in LDS)
        if (bl >= 1 && bl <= stage_cap) { zstd_stage(w->stage, bs, bl); bs = w->stage; bs_lds = 1; }
        if (bl < 1 || zstd_u8(bs, bl - 1) == 0) return ERR_DATA;
        FOR_LANES_W(l) {
            w->seq_bs = bs; w->seq_bl = bl; w->seq_bs_lds = bs_lds; w->seq_nseq = nseq;
            w->seq_dst = dst; w->seq_dcap = dcap; w->seq_dpos = dpos; w->seq_lit = lit; w->seq_regen = regen; w->seq_lpos = 0;
            w->r0 = fs->r0; w->r1 = fs->r1; w->r2 = fs->r2;
        }
#ifdef CIMG_ABL_ZSTD_NO_SEQ      /* timing experiment only */
        if (nseq >= 0) return dpos;
#endif
        const bool typed = bs_lds && zstd_field((const int32_t*)&w->seq_bs_lds) && w->mem_lo != nullptr;
        int rc;
        if (walk) {
            if (nseq > zstd_field(&w->rec_cap) - zstd_field(&w->rec_n)) return ZSTD_WALK_OVERFLOW;
            rec_first = zstd_field(&w->rec_n);
            rc = typed ? zstd_sequences<cimg_lds_cu8p, true>(w) : zstd_sequences<const uint8_t*, true>(w);
        } else rc = typed ? zstd_sequences<cimg_lds_cu8p>(w) : zstd_sequences<const uint8_t*>(w);
        if (rc < 0) return rc;
        dpos = rc;
        lpos = zstd_field(&w->seq_lpos);
        fs->r0 = zstd_field(&w->r0); fs->r1 = zstd_field(&w->r1); fs->r2 = zstd_field(&w->r2);
    }
    const int rest = regen - lpos;
    if (rest > dcap - dpos) return ERR_DATA;
    if (walk) {
        ZstdOp op;
        op.kind = ZOP_SEQ; op.size = (uint32_t)regen; op.ptr = lit_word; op.nseq = (uint32_t)nseq; op.rec = (uint32_t)rec_first; op.lit_fill = lit_fill; op.stream = 0;
        const int rc = zstd_emit(w, op);
        if (rc < 0) return rc;
    } else zstd_copy(dst + dpos, lit + lpos, rest);
    return dpos + rest;
}

// One frame at src[0, size) -> dst[0, cap).  Returns the regenerated size or a negative blosc2 error code.
// A walker (w->ops set; dst unused) leaves the frame's ops, records and coded literals in its plan instead: zstd_replay_frame below
// is the other half.  ZSTD_WALK_OVERFLOW: the plan's slot is too small for this frame.
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
    const bool walk = zstd_walking(w);
    if (walk) { FOR_LANES_W(l) { w->frame_jobs = 0; } }
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
            if (walk) {
                ZstdOp op;
                op.kind = ZOP_RAW; op.size = (uint32_t)bsz; op.ptr = (uint64_t)reinterpret_cast<uintptr_t>(src + pos); op.nseq = 0; op.rec = 0; op.lit_fill = 0; op.stream = 0;
                const int rc = zstd_emit(w, op);
                if (rc < 0) return rc;
            } else zstd_copy(dst + dpos, src + pos, bsz);
            dpos += bsz; pos += bsz;
        } else if (type == 1) {
            if (pos + 1 > size || bsz > cap - dpos) return ERR_DATA;
            if (walk) {
                ZstdOp op;
                op.kind = ZOP_FILL; op.size = (uint32_t)bsz; op.ptr = (uint64_t)zstd_u8(src, pos); op.nseq = 0; op.rec = 0; op.lit_fill = 0; op.stream = 0;
                const int rc = zstd_emit(w, op);
                if (rc < 0) return rc;
            } else zstd_fill(dst + dpos, zstd_u8(src, pos), bsz);
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
            const bool deferred = walk && zstd_field(&w->defer);
            if (fcs >= 0 && fcs != dpos && !deferred) return ERR_DATA;
            if (walk) {
                ZstdOp op;
                // (a walker that left sequences to the lane decoder does not know how far the frame got: the replay is told what the
                // frame header promised, or -- no promise -- the capacity, which is what the caller requires of a stream anyway)
                op.kind = ZOP_END; op.size = (uint32_t)(deferred ? (fcs >= 0 ? (int)fcs : cap) : dpos); op.ptr = 0; op.nseq = 0; op.rec = 0; op.lit_fill = 0; op.stream = 0;
                const int rc = zstd_emit(w, op);
                if (rc < 0) return rc;
            }
            return deferred ? cap : dpos;
        }
    }
    return ERR_DATA;
}

// ---- the other half of a walked frame -------------------------------------------------------------------------------------
// ops[first ..] up to the frame's ZOP_END -> dst[0, cap), the way zstd_decode_frame writes it: raw and run blocks, and of a
// compressed block the literals brought to the end of the output, its records executed 64 at a time (the next 64 are requested
// before the current ones are executed), the literals behind the last sequence.  Nothing a record says is trusted: the batch
// executor checks lengths, offsets and positions against the output as it does for the decoder proper.  Returns the regenerated
// size (and the index behind the frame's last op) or a negative error code.
template <class DP>
CIMG_DEV int zstd_replay_frame(const ZstdOp* ops_in, int first_in, int nops_in, int stream_in, const uint64_t* recs_in, int nrecs_in,
                               uint8_t* dst, int cap, const uint8_t* mlo, const uint8_t* mhi, int* next_op)
{
    const ZstdOp* const ops = reinterpret_cast<const ZstdOp*>(uni64((int64_t)reinterpret_cast<uintptr_t>(ops_in)));
    const uint64_t* const recs = reinterpret_cast<const uint64_t*>(uni64((int64_t)reinterpret_cast<uintptr_t>(recs_in)));
    const int first = uni(first_in), nops = uni(nops_in), stream = uni(stream_in), nrecs = uni(nrecs_in);
    int dpos = 0;
    for (int i = first; i < nops; ++i) {
        const uint32_t* const q = reinterpret_cast<const uint32_t*>(ops + i);
        const uint32_t kind = uni(q[0]), size = uni(q[1]), p_lo = uni(q[2]), p_hi = uni(q[3]), nseq = uni(q[4]), rec = uni(q[5]), lit_fill = uni(q[6]), st = uni(q[7]);
        const uint64_t ptr = (uint64_t)p_lo | ((uint64_t)p_hi << 32);
        if ((int)st != stream || (int)size < 0) return ERR_DATA;
        if (kind == ZOP_END) {
            if ((int)size != dpos) return ERR_DATA;
            *next_op = i + 1;
            return dpos;
        }
        if ((int)size > cap - dpos) return ERR_DATA;
        if (kind == ZOP_RAW) {
            zstd_stage(dst + dpos, reinterpret_cast<const uint8_t*>((uintptr_t)ptr), (int)size);
            dpos += (int)size;
        } else if (kind == ZOP_FILL) {
            zstd_fill(dst + dpos, (uint8_t)ptr, (int)size);
            dpos += (int)size;
        } else if (kind == ZOP_SEQ) {
            const int regen = (int)size;
            uint8_t* const litbuf = dst + (cap - regen);
            if (lit_fill) zstd_fill(litbuf, (uint8_t)ptr, regen);
            else zstd_stage(litbuf, reinterpret_cast<const uint8_t*>((uintptr_t)ptr), regen);
            if ((int)nseq < 0 || (int)rec < 0 || (int64_t)rec + nseq > (int64_t)nrecs) return ERR_DATA;
            int lpos = 0;
            LV<uint64_t> cur, nxt;
            FOR_LANES(l) { cur[l] = l < (int)nseq ? recs[rec + l] : 0; nxt[l] = 0; }
            for (int b = 0; b < (int)nseq; b += 64) {
                const int nb = imin(64, (int)nseq - b);
                FOR_LANES(l) { if (b + 64 + l < (int)nseq) nxt[l] = recs[rec + b + 64 + l]; }
                LV<int> vll, vml, vof;
                FOR_LANES(l) {
                    vll[l] = (int)(cur[l] & 0x1FFFFF); vml[l] = (int)((cur[l] >> 21) & 0x1FFFFF); vof[l] = (int)(cur[l] >> 42);
                }
                const int rc = zstd_execute_batch<DP>((DP)dst, cap, &dpos, (DP)litbuf, regen, &lpos, nb, vll, vml, vof, mlo, mhi);
                if (rc < 0) return rc;
                FOR_LANES(l) { cur[l] = nxt[l]; }
            }
            const int rest = regen - lpos;
            if (rest > cap - dpos) return ERR_DATA;
            zstd_copy(dst + dpos, litbuf + lpos, rest);
            dpos += rest;
        } else return ERR_DATA;
    }
    return ERR_DATA;
}

}  // namespace cimg
