// encode_kernel.h -- blosc2 chunk encode for gfx950: shuffle + run detection + LZ4 per stream.
//
// Replaces what the reference reaches through blosc2_compress_ctx (blosc2/wrapper.h:139,172, called
// per 4 MiB chunk from schunk.h:85-104).  One single-wave workgroup per STREAM (one byte plane of one
// 32 KiB block): streams differ wildly in cost (a noisy low-byte plane is scanned once and stored raw,
// a smooth high-byte plane is hundreds of matches), so the stream -- not the block -- is the
// scheduling unit; LDS per workgroup is exactly plane + hash table = 32 KiB, five workgroups per CU.
// The workgroups are persistent and pull streams from an atomic queue, most significant byte planes first
// (the hardware places workgroups statically: with one stream per workgroup some CUs got only slow planes).
//
//   phase A  the wave reads its block with 16-byte coalesced loads and keeps byte plane s in LDS
//            (the byte-shuffle filter, fused with the stream split).
//   phase B  run check, then a *bit-exact* LZ4_compress_fast of the plane with the byU16 hash table
//            (8192 x u16) in LDS.  The match search is inherently sequential (every probe reads and
//            writes the table), so it runs as 64-probe windows:
//              - lane l takes probe l of the skip schedule (positions are known in advance),
//              - a lane whose hash equals the previous lane's hash has that lane as its candidate
//                (exactly what a sequential scan would find), the other lanes ("heads") read their
//                table slot, write their position and read it back; a head whose read-back differs
//                shares its slot with another head of the window,
//              - the window is committed up to B = min(first matching lane, first head involved in
//                a slot collision): lanes <= B see exactly the table a sequential scan would show
//                them; heads > B put their old slot value back,
//              - a match at lane B is extended backwards and forwards with lane-parallel compares
//                (one LDS round trip for both); the "probe right after the match" of the sequential
//                algorithm rides as lane 0 of the next window.
//            In front of the windows sit three cheaper paths for what compressible data mostly needs (DESIGN.md
//            section 4): the scalar head (post-match probe + first two probes from 10 bytes in scalars), the
//            run path and the narrow path.  Found sequences are parked in lanes and written 64 at a time.
//            Output goes to the block's scratch slot; a per-stream record (kind, size, need) is left
//            for the layout kernel.  `need` is the smallest output budget under which LZ4 still
//            succeeds; it lets the layout kernel re-apply blosc2's running-destsize rule without
//            re-encoding (oracle/chunk.c: orc_blosc2_compress_2phase is the CPU twin).
#pragma once
#include "codec_types.h"
#include "wave.h"
#include "decode_kernel.h"   // round16, find_chunk, byte_perm, wave copies
#include "blosclz_kernel.h"
#include "zstd_encode.h"      // the sequence sink of lz4_encode_body<true> and the frame around it (codec::zstd)
#include "assemble_kernel.h"  // LayoutChunk / EmitBlock: run by the waves of the encode launch when the batch is assembled in place

namespace cimg {

struct EncodeArgs {
    const ChunkDesc* descs;
    int32_t nchunks;
    CodecParams p;
    const uint8_t* raw;       // pixels at raw + desc.raw_off
    uint8_t* scratch;         // block b owns scratch + b * p.slot_bytes
    StreamRec* recs;          // block b, stream s -> recs[b * p.streams_per_block + s]
    int32_t lds_bytes;
    int32_t total_blocks;
    int32_t want_split;       // 1: this launch encodes the planes of split blocks, 0: unsplit blocks
    uint64_t* dbg;            // diagnostics only: per-item time stamps (nullptr in production)
    uint32_t* queue;          // work queue: ENC_NQ heads, ENC_QSTRIDE words apart, zero at launch (see "the work queue" below)
    int32_t uniform_nblocks;  // > 0: every chunk has this many blocks
    int32_t block_items;      // split launch only: the first `block_items` blocks are handed out WHOLE -- read from HBM once, the
                              // byte planes encoded one after the other by the same wave (the planes that wait sit in registers);
                              // the blocks behind them plane by plane
    // codec::zstd (cimg_encode_streams_zstd): per-wave sequence records and the FSE tables of the predefined distributions
    uint32_t* zstd_seq;       // wave w owns zstd_seq + w * zstd_seq_stride (dwords)
    int32_t zstd_seq_stride;
    const ZstdEncTables* zstd_tables;
    uint32_t* queue_next;     // the heads the NEXT launch of this kind will use: wave 0 zeroes them (the launches of one kind follow
                              // each other in stream order, so nobody reads them while this launch runs)
    int32_t nwaves;           // waves of the launch
    // ---- chunks assembled INSIDE the launch (assemble != 0; round 3) ---------------------------------------------------------
    // A wave remembers the items it encoded (a linked list through next_item[]); the wave that finishes the last stream of a
    // chunk lays the chunk out (LayoutChunk: bstarts, header, the running-destsize rule) and raises ready[chunk]; when the work
    // queue is empty every wave copies ITS OWN streams from the scratch slots into place -- out of its own XCD's L2, in the time
    // the launch's tail leaves most waves idle anyway (4096 coded planes on 1280 chains: a fifth of the chains work through a
    // fourth plane while the rest are done).  Only chunks whose descriptor says so (ChunkDesc::assemble: every stream of the chunk
    // belongs to ONE launch) take part; cimg_layout_chunks / cimg_emit_blocks run behind the launch for the others.
    int32_t assemble;
    uint8_t* comp;            // chunks are written at comp + desc.comp_off
    ChunkLayout* layout;      // per chunk (device)
    ChunkLayout* layout_host; // the same in page-locked host memory, plus ONE extra entry [nchunks] whose cbytes turns negative when
                              // a wave gave up waiting for a chunk to be laid out (the host then fails the batch loudly)
    uint32_t* chunk_count;    // per chunk: streams finished so far (the closer puts 0 back)
    uint32_t* ready;          // per chunk: == gen once the chunk is laid out
    int32_t* next_item;       // per item of this launch: the item the same wave encoded before it (-1: none)
    uint32_t gen;
};

// ---- the work queue (round 4) -------------------------------------------------------------------------------------------------
// Items are dealt out by atomic counters.  Device-scope atomics on ONE address are served one after the other (about 10 ns each), so
// a launch whose 1280 waves all pop one head at t = 0 starts its last chain 13 us late.  Round 3 therefore gave wave w item w without
// asking -- which made the launch depend on every workgroup being resident: with chunks assembled inside the launch a wave that is
// out of work WAITS for its chunks to close, i.e. also for first items of workgroups that have not started, and when another
// persistent launch holds their CUs (a second engine, a second process on the card) that is a deadlock.
// Now EVERY item is popped -- an item always belongs to a running wave, so a launch completes with any number of resident waves --
// from ENC_NQ heads on separate cache lines: item i sits in sub-queue i mod ENC_NQ, wave w pops sub-queue w mod ENC_NQ (20 waves
// per head: 0.2 us of serialisation instead of 13) and, once that is dry, looks at all heads with ONE vector load and pops the next
// sub-queue that still has items.  Sub-queues advance at about the same pace, so the order of the item list (whole blocks, then
// coded planes, then noise planes) holds approximately.  Pops past the end are harmless: each launch zeroes the heads of the next.
enum : int { ENC_NQ = 64, ENC_QSTRIDE = 32 };

enum : int { LZ4_HASH_BYTES = 16384, LZ4_MAX_INPUT_U16 = 65536 + 11 - 1 };

// LDS of one stream workgroup: the plane (padded to 16), then the hash table
inline int encode_lds_bytes(int stream_bytes, int compcode = CODEC_LZ4)
{
    return compcode == CODEC_BLOSCLZ ? blz_encode_lds_bytes(stream_bytes) : round16(stream_bytes) + (compcode == CODEC_ZSTD ? (2 << ZSTD_ENC_HASH_BITS) : LZ4_HASH_BYTES);
}
// Work items of a launch.  Streams differ in cost by an order of magnitude and the hardware places
// workgroups on CUs round-robin, not first-free, so the launch is a set of persistent workgroups that
// pull items from a queue.  Split launch: item i is plane (spb-1 - i / total_blocks) of block
// i % total_blocks -- most significant byte planes (smooth, many matches, slow) first, noisy low planes
// last (longest-processing-time-first).  Unsplit launch: item i is block i.  A split launch may hand the first
// `whole_blocks` blocks out whole (one item each, in front of the plane items of the rest): a block read once instead of once
// per plane, at the price of a coarser item -- which is why only the rounds that every chain takes anyway are dealt that way
// and the last, partial round stays plane by plane (engine.hip).
CIMG_HD int encode_items(int total_blocks, int streams_per_block, bool split, int whole_blocks = 0)
{
    return split ? whole_blocks + (total_blocks - whole_blocks) * streams_per_block : total_blocks;
}
// A split launch can hand out whole blocks when every full block is the standard 32 KiB of a 2- or 4-byte type, byte
// shuffled: then the planes that wait for their turn fit in registers (16 KiB / 24 KiB = 64 / 96 VGPRs).
CIMG_HD bool encode_block_items_ok(int typesize, int filter, int blocksize)
{
    return (typesize == 2 || typesize == 4) && filter == FILTER_SHUFFLE && blocksize == 32768;
}

// item time stamps (diagnostics); a -DCIMG_PROFILE build keeps its cycle laps in the same 16 words per item instead
#if defined(CIMG_PROFILE) && !defined(CIMG_EMULATE)
#define CIMG_ITEM_STAMP(dbg, item, k) ((void)0)
#else
#define CIMG_ITEM_STAMP(dbg, item, k) debug_stamp(dbg, item, k)
#endif

#ifdef CIMG_EMULATE
extern long g_emu_windows, g_emu_matches, g_emu_collisions;   // test-side statistics only
#define CIMG_STAT(x) (++(x))
#else
#define CIMG_STAT(x) ((void)0)
#endif

template <int HB = 13>
CIMG_DEV uint32_t lz4_hash(uint32_t v) { return (v * 2654435761u) >> (32 - HB); }
// sum_{x=0}^{n-1} (x >> 6)
CIMG_DEV int skip_prefix(int n) { const int q = n >> 6, r = n & 63; return q * (32 * (q - 1) + r); }

// write an LZ4 length extension for `rem` (rem/255 times 255, then rem%255) at out[pos..)
CIMG_DEV void emit_len_ext(cimg_global_u8p out, int pos, int rem)
{
    const int n255 = rem / 255, last = rem - 255 * n255;
#ifndef CIMG_EMULATE
#pragma unroll 1
#endif
    for (int c = 0; c <= n255; c += 64) {
        FOR_LANES(l) { if (c + l <= n255) out[pos + c + l] = (uint8_t)(c + l < n255 ? 255 : last); }
    }
}

CIMG_DEV void emit_literals(const uint8_t* in, int from, cimg_global_u8p out, int pos, int count)
{
#ifndef CIMG_EMULATE
#pragma unroll 1
#endif
    for (int c = 0; c < count; c += 64) {
        FOR_LANES(l) { if (c + l < count) out[pos + c + l] = in[from + c + l]; }
    }
}

// Continues a forward match count: bytes in[ip + 4 + k] == in[mp + 4 + k] for k = from, from + 1, ... up to maxc,
// 256 bytes per LDS round trip.  Returns the total count (match code).  Loads are unguarded: the hash table
// follows the plane in LDS, and lanes past maxc are cut by the min.
CIMG_DEV int match_more(const uint8_t* in, int ip, int mp, int maxc, int from, int n)
{
    int mcode = from;
    for (int it = 0; it <= n / 256 + 1; ++it) {                   // k reaches maxc <= n within that many steps
        LV<int> len;
        LV<bool> stop;
        FOR_LANES(l) {
            const int k = mcode + 4 * l;
            const uint32_t x = lds_ld32u(in, ip + 4 + k) ^ lds_ld32u(in, mp + 4 + k);
            const int ln = imin(x ? (int)(__builtin_ctz(x) >> 3) : 4, imax(maxc - k, 0));
            len[l] = ln;
            stop[l] = ln < 4;
        }
        const uint64_t sm = ballot(stop);
        if (sm) { const int f = ctz64(sm); return mcode + 4 * f + readlane(len, f); }
        mcode += 256;
    }
    return mcode;
}

CIMG_DEV int div255(int x) { return (int)(((uint64_t)(uint32_t)x * 0x80808081ull) >> 39); }

// Writes the np parked sequences (lane k = k-th sequence) at out[op..) and advances op.  The limited-output
// checks of LZ4_compress_generic are evaluated for all of them at once: the encoder's result is 0 as soon as
// ANY check fails and `need` is the maximum of all left-hand sides, so checking late gives the same answer.
// Returns false when the output does not fit cap.
CIMG_DEV bool emit_pending(const uint8_t* in, cimg_global_u8p out, int cap, int& op, int& need, int np,
                           const LV<int>& P_anchor, const LV<int>& P_lit, const LV<int>& P_off, const LV<int>& P_mcode)
{
    LV<int> size, start, le, me;
    FOR_LANES(l) {
        const int lit = P_lit[l], mc = P_mcode[l];
        le[l] = lit >= 15 ? div255(lit - 15) + 1 : 0;             // literal-length extension bytes
        me[l] = mc >= 15 ? div255(mc - 15) + 1 : 0;               // match-length extension bytes
        size[l] = l < np ? lit + 3 + le[l] + me[l] : 0;
    }
    int total;
    wave_exscan(size, start, total);
    LV<int> lhs, tokpos, litpos;
    LV<bool> fail, ext, shortlit, longlit;
    FOR_LANES(l) {
        const int lit = P_lit[l], mc = P_mcode[l];
        const bool act = l < np;
        tokpos[l] = op + start[l];
        litpos[l] = tokpos[l] + 1 + le[l];
        const int lhs1 = tokpos[l] + 1 + lit + 8 + div255(lit);
        const int lhs2 = litpos[l] + lit + 8 + div255(mc + 240);
        fail[l] = act && (lhs1 > cap || lhs2 > cap);
        lhs[l] = act ? imax(lhs1, lhs2) : 0;
        ext[l] = act && (le[l] > 0 || me[l] > 0);
        shortlit[l] = act && lit > 0 && lit <= 8;
        longlit[l] = act && lit > 8;
    }
    if (ballot(fail)) return false;
    need = imax(need, wave_max(lhs));
    FOR_LANES(l) {
        if (l < np) {
            const int lit = P_lit[l], mc = P_mcode[l], off = P_off[l];
            const int offpos = litpos[l] + lit;
            out[tokpos[l]] = (uint8_t)(((lit >= 15 ? 15 : lit) << 4) | (mc >= 15 ? 15 : mc));
            out[offpos] = (uint8_t)(off & 0xFF);
            out[offpos + 1] = (uint8_t)(off >> 8);
        }
    }
    if (ballot(ext)) {
        // Length bytes, one byte per lane and round -- for the lanes with at most four of them.  The few with more (a match of
        // 1035 bytes or longer: 17 for a 4 KiB row of an image) would set the round count for everybody: those are written 64
        // bytes at a time by the whole wave, one after the other.
        for (int r = 0; r < 4; ++r) {
            LV<bool> more;
            FOR_LANES(l) {
                const int lit = P_lit[l], mc = P_mcode[l];
                const bool act = l < np;
                const int le4 = le[l] <= 4 ? le[l] : 0, me4 = me[l] <= 4 ? me[l] : 0;
                if (act && r < le4) out[tokpos[l] + 1 + r] = (uint8_t)(r == le4 - 1 ? lit - 15 - 255 * (le4 - 1) : 255);
                if (act && r < me4) out[litpos[l] + lit + 2 + r] = (uint8_t)(r == me4 - 1 ? mc - 15 - 255 * (me4 - 1) : 255);
                more[l] = act && (r + 1 < le4 || r + 1 < me4);
            }
            if (!ballot(more)) break;
        }
        LV<bool> bigl, bigm;
        FOR_LANES(l) { bigl[l] = l < np && le[l] > 4; bigm[l] = l < np && me[l] > 4; }
        for (uint64_t m = ballot(bigl); m; m &= m - 1) {
            const int k = ctz64(m);
            emit_len_ext(out, readlane(tokpos, k) + 1, readlane(P_lit, k) - 15);
        }
        for (uint64_t m = ballot(bigm); m; m &= m - 1) {
            const int k = ctz64(m);
            emit_len_ext(out, readlane(litpos, k) + readlane(P_lit, k) + 2, readlane(P_mcode, k) - 15);
        }
    }
    if (ballot(shortlit)) {
        for (int r = 0; r < 8; ++r) {                              // short literal runs: one lane per sequence
            LV<bool> more;
            FOR_LANES(l) {
                if (shortlit[l] && r < P_lit[l]) out[litpos[l] + r] = in[P_anchor[l] + r];
                more[l] = shortlit[l] && r + 1 < P_lit[l];
            }
            if (!ballot(more)) break;
        }
    }
    uint64_t longm = ballot(longlit);
    while (longm) {                                                // long literal runs: the whole wave copies one
        const int k = ctz64(longm);
        longm &= longm - 1;
        emit_literals(in, readlane(P_anchor, k), out, readlane(litpos, k), readlane(P_lit, k));
    }
    op += total;
    return true;
}

// Is there ANY match in the plane?  (No, for the low mantissa bytes of an image.)  The probes of LZ4's first search sit at
// positions that do not depend on the data for as long as nothing is found, so that search is walked here window by
// window with the bare minimum -- slots read, written, read back, one candidate compare -- and without any of the
// sequence state of lz4_encode_body being alive.  Two probes of a window that share a slot are settled in place (the later
// one owns the slot and has the earlier one as its candidate); three or more, or any candidate that compares equal, end
// the walk with "maybe": the caller clears the table again and runs the full encoder, which costs a compressible plane one
// wasted window.  Returns true when the search reached the end of the plane without a match: the block is its literals.
template <int HB = 13>
CIMG_DEV bool lz4_no_match_at_all(const uint8_t* in, cimg_lds_vu16p tab16, int n, int s64, int f64, int mflimit_p1)
{
    bool verdict = false;                                               // one way out of the loop (see lz4_encode_body)
    for (int t0 = 0; t0 <= n; t0 += 64) {                               // hard bound: a window commits 64 probes
        // TWO windows per LDS round trip while nothing happens: the second window's slot reads are issued behind the first
        // window's writes (LDS executes a wave's operations in order), so it sees exactly the table sequential LZ4 would show it.
        // Anything eventful in either -- two probes of one window sharing a slot included -- puts both back and leaves the
        // first to the one-window form below.
        {
            LV<int> p1, p2;
            LV<bool> in2;
            FOR_LANES(l) {
                const int ta = t0 + l, tb = t0 + 64 + l;
                int pa = 1 + ta;
                if (ta > 0) pa = 2 + skip_prefix(s64 + ta - 1) - f64;
                const int pb = 2 + skip_prefix(s64 + tb - 1) - f64, gb = (s64 + tb - 1) >> 6;
                p1[l] = pa; p2[l] = pb;
                in2[l] = pb + gb <= mflimit_p1;                                  // the later window inside means the earlier one is
            }
            if (!~ballot(in2)) {
                LV<uint32_t> v1, v2, h1, h2, o1, o2, r1, r2, c1, c2;
                FOR_LANES(l) { v1[l] = lds_ld32u(in, p1[l]); v2[l] = lds_ld32u(in, p2[l]); h1[l] = lz4_hash<HB>(v1[l]); h2[l] = lz4_hash<HB>(v2[l]); }
                FOR_LANES(l) { o1[l] = tab16[h1[l]]; }
                FOR_LANES_W(l) { tab16[h1[l]] = (uint16_t)p1[l]; }
                FOR_LANES(l) { r1[l] = tab16[h1[l]]; }
                FOR_LANES(l) { o2[l] = tab16[h2[l]]; }
                FOR_LANES_W(l) { tab16[h2[l]] = (uint16_t)p2[l]; }
                FOR_LANES(l) { r2[l] = tab16[h2[l]]; c1[l] = lds_ld32u(in, (int)o1[l]); c2[l] = lds_ld32u(in, (int)o2[l]); }
                LV<bool> eventful;
                FOR_LANES(l) {
                    eventful[l] = (r1[l] != (uint32_t)(uint16_t)p1[l]) | (c1[l] == v1[l]) | (r2[l] != (uint32_t)(uint16_t)p2[l]) | (c2[l] == v2[l]);
                }
                if (!ballot(eventful)) { CIMG_STAT(g_emu_windows); CIMG_STAT(g_emu_windows); t0 += 64; continue; }
                FOR_LANES_W(l) { tab16[h2[l]] = (uint16_t)o2[l]; }          // the later window first: it may hold the earlier one's positions
                FOR_LANES_W(l) { tab16[h1[l]] = (uint16_t)o1[l]; }
            }
        }
        LV<int> pos;
        LV<bool> inside;
        FOR_LANES(l) {
            const int t = t0 + l;
            int p = 1 + t, gap = 1;                                      // the search starts at position 1
            if (t > 0) { p = 2 + skip_prefix(s64 + t - 1) - f64; gap = (s64 + t - 1) >> 6; }
            pos[l] = p;
            inside[l] = p + gap <= mflimit_p1;
        }
        const uint64_t in_mask = ballot(inside);                         // a prefix of the lanes
        if (!in_mask) { verdict = true; break; }
        LV<uint32_t> v, h, old, rb, cv;
        LV<bool> lost, equal;
        FOR_LANES(l) { v[l] = lds_ld32u(in, inside[l] ? pos[l] : 0); h[l] = lz4_hash<HB>(v[l]); }
        FOR_LANES(l) { old[l] = tab16[h[l]]; }
        FOR_LANES_W(l) { if (inside[l]) tab16[h[l]] = (uint16_t)pos[l]; }
        FOR_LANES(l) { rb[l] = tab16[h[l]]; cv[l] = lds_ld32u(in, (int)old[l]); }
        FOR_LANES(l) {
            lost[l] = inside[l] & (rb[l] != (uint32_t)(uint16_t)pos[l]);
            equal[l] = inside[l] & (cv[l] == v[l]);
        }
        if (ballot(equal)) break;
        if (ballot(lost)) {
            // the lanes that lost their slot write again: one loser per slot means pairs, and its second read-back is its own
            LV<uint32_t> rb2, wv;
            LV<bool> crowd, same;
            FOR_LANES_W(l) { if (lost[l]) tab16[h[l]] = (uint16_t)pos[l]; }
            FOR_LANES(l) { rb2[l] = tab16[h[l]]; wv[l] = lds_ld32u(in, (int)rb[l]); }
            FOR_LANES(l) {
                crowd[l] = lost[l] & (rb2[l] != (uint32_t)(uint16_t)pos[l]);
                same[l] = lost[l] & (wv[l] == v[l]);                     // the other probe of the pair has the same four bytes
            }
            if (ballot(crowd) | ballot(same)) break;
            // the later probe of each pair owns the slot
            FOR_LANES_W(l) { if (inside[l] && rb2[l] != (uint32_t)(uint16_t)pos[l] && (uint32_t)pos[l] > rb2[l]) tab16[h[l]] = (uint16_t)pos[l]; }
        }
        CIMG_STAT(g_emu_windows);
        if (~in_mask) { verdict = true; break; }                         // the window that saw the end of the plane
    }
    return verdict;
}

// Bit-exact LZ4_compress_fast(in, out, n, cap, accel) in limited-output mode, byU16 table, by one wave.
// in: LDS plane (8 readable bytes past the end), tab: 16 KiB LDS.  Returns bytes written, 0 if the
// result does not fit cap.  need_out = smallest cap that still succeeds.
// SEQ = true (the zstd encoder, zstd_encode.h): the same search, but what it finds goes to `sink` as (literal length, offset, match
// length) records, and `out` receives only the literal bytes -- cap then bounds those, and the return value is the number of
// sequences (0: none, or the literals do not fit).
template <bool SEQ = false>
CIMG_DEV int lz4_encode_body(const uint8_t* in, uint8_t* tab, int n, uint8_t* out_generic, int cap, int accel, int& need_out, uint64_t* dbg = nullptr, int item = 0, SeqSink* sink = nullptr)
{
    // 13 bits = LZ4's byU16 table, which the byte-pinned LZ4 stream needs; the zstd frames around the same match finder are pinned
    // to nothing, and a 4096-slot table lets a fourth chain of 32 KiB streams live on a CU (zstd_encode.h)
    constexpr int HB = SEQ ? ZSTD_ENC_HASH_BITS : 13;
    cimg_global_u8p out = CIMG_AS_GLOBAL(out_generic);
    CIMG_PROF_DECL;
    (void)dbg; (void)item;

    // volatile: other lanes of the wave write the same slots, and the read-back after a write is exactly
    // how collisions are detected -- the compiler must not forward the lane's own store to that load
    cimg_lds_vu16p tab16 = CIMG_AS_LDS_VU16(tab);
    {
        const u128 z = {0, 0, 0, 0};
        for (int u0 = 0; u0 < (2 << HB) / 16; u0 += 64) {
            FOR_LANES(l) { st128a(tab + 16 * (u0 + l), z); }
        }
    }
    const int mflimit_p1 = n - 11, matchlimit = n - 5;
    const int s64 = accel << 6;
    const int f64 = skip_prefix(s64);
    int anchor = 0, op = 0, need = 0;
    // sequences found but not yet written: lane k holds the k-th (anchor, literal count, offset, match code)
    LV<int> P_anchor, P_lit, P_off, P_mcode;
    int np = 0;
    FOR_LANES(l) { P_anchor[l] = 0; P_lit[l] = 0; P_off[l] = 0; P_mcode[l] = 0; }

    bool hopeless = false;
    if (n >= 13 + 64) {
        hopeless = lz4_no_match_at_all<HB>(in, tab16, n, s64, f64, mflimit_p1);
        if (hopeless) anchor = 0;
        else {
            const u128 z = {0, 0, 0, 0};
            for (int u0 = 0; u0 < (2 << HB) / 16; u0 += 64) {
                FOR_LANES(l) { st128a(tab + 16 * (u0 + l), z); }
            }
        }
    }
    if (n >= 13 && !hopeless) {
        {   // first byte
            LV<uint32_t> v0;
            FOR_LANES(l) { v0[l] = lds_ld32u(in, 0); }
            FOR_LANES_W(l) { tab16[lz4_hash<HB>(v0[l])] = 0; }              // every lane: same slot, same value
        }
        CIMG_PROF_LAP(0);                                  // table clear + first byte
        int sstart = 1;     // search start position
        int t0 = 0;         // probes of this search already committed
        int pre = 0;        // 1: lane 0 is the probe right after a match (position sstart - 1)
        LV<uint32_t> Wn;    // bytes [ip - 2, ip + 10) around the end of the last match, lanes 0..2 (scalar head)
        FOR_LANES(l) { Wn[l] = 0; }
        [[maybe_unused]] int chain_credit = 2;   // zero-literal chain below: tried while it pays -- a score in [-8, 8], + 3 for a zero-literal sequence, - 2 for a
                                // miss of the chain's first probe or (while the chain is off) a sequence with literals; the chain runs above 0
        int guard = 0;      // every wave must reach an exit: a window commits at least one probe, a head iteration finds a match (>= 4 bytes)
                            // or is followed by a window, so 2 n + 4 iterations is a hard bound
        // ONE way out of the loop: every exit is a `break` with `ending` saying why.  With returns inside the loop the compiler
        // unifies the exits through a selector that EVERY iteration is then dispatched on (≈ 20 scalar instructions and
        // three taken branches per sequence).
        int ending = 1;     // 1: the search reached the end of the plane; 0: the output does not fit; < 0: a loop guard tripped
        for (;;) {
            if (++guard > 2 * n + 4) { ending = -1; break; }
            // ---- lay the window out ------------------------------------------------------------------
            LV<int> pos;
            LV<bool> valid;
            LV<uint32_t> v, h, back;
            int ip = 0, mp = 0, mcode = 0, backrun = 0;
            bool zero_lit = false;
            // ---- run fast path (a lambda: used by the scalar head below and, near the end of a stream, after the layout)
            // The probe right after a match very often lands on the first byte of a run (flat image areas:
            // high byte planes, masks, alpha).  Then probe 0 of the new search has the same four bytes, the
            // same hash and -- unless the post-match probe itself matches -- is a match at offset 1.  That
            // outcome needs only the post-match probe's own table slot, so the 64-lane window machinery is
            // skipped: slot read + run-length scan in one LDS round trip, candidate check in a second.
            auto run_path = [&](const uint32_t v0, const uint32_t backv) {
                const int ip0 = sstart - 1;
                const uint32_t h0 = lz4_hash<HB>(v0);
                const uint32_t bbbb = (v0 & 0xFF) * 0x01010101u;            // v0 == v1 means v0 is four equal bytes
                FOR_LANES_W(l) { tab16[lz4_hash<HB>(backv)] = (uint16_t)(sstart - 3); }   // the "put(ip - 2)" refill (pre == 1 on every call)
                LV<uint32_t> slot, scan;
                LV<uint32_t> before;
                FOR_LANES(l) {
                    slot[l] = tab16[h0];
                    scan[l] = lds_ld32u(in, ip0 + 5 + 4 * l);                 // bytes after the five known run bytes
                    before[l] = in[ip0 - 1];
                }
                const int old0 = (int)readlane(slot, 0);
                // second round trip: the candidate's bytes against the bytes at the probe, 256 of them -- this
                // decides the hit AND, for a hit, already is the match extension (no literals: nothing backwards)
                LV<uint32_t> cw, iw;
                FOR_LANES(l) {
                    cw[l] = lds_ld32u(in, old0 + 4 * l);
                    iw[l] = lds_ld32u(in, ip0 + 4 * l);
                }
                // (without this the compiler moves the loads of scan / before / iw into the branches below: a third round trip)
                needed_here(scan); needed_here(before); needed_here(iw);
                const bool hit0 = readlane(cw, 0) == v0;
                CIMG_STAT(g_emu_matches);
                if (hit0) {
                    CIMG_PROF_COUNT(5);
                    // zero-literal match at the post-match probe
                    FOR_LANES_W(l) { tab16[h0] = (uint16_t)ip0; }
                    ip = ip0; mp = old0; zero_lit = true; backrun = 0;
                    const int maxc = matchlimit - (ip0 + 4);
                    LV<int> len;
                    LV<bool> stop;
                    FOR_LANES(l) {
                        const int k = 4 * (l - 1);                          // lane 0 holds the four matched bytes
                        const uint32_t x = cw[l] ^ iw[l];
                        const int ln = imin(x ? (int)(__builtin_ctz(x) >> 3) : 4, imax(maxc - k, 0));
                        len[l] = ln;
                        stop[l] = (l >= 1) & (ln < 4);
                    }
                    const uint64_t sm = ballot(stop);
                    if (sm) { const int f = ctz64(sm); mcode = 4 * (f - 1) + readlane(len, f); }
                    else mcode = match_more(in, ip0, old0, maxc, 252, n);
                } else {
                    FOR_LANES_W(l) { tab16[h0] = (uint16_t)(ip0 + 1); }
                    ip = ip0 + 1; mp = ip0; zero_lit = false; mcode = 0;
                    backrun = (readlane(before, 0) == (v0 & 0xFF)) ? 1 : 0;      // room is min(ip - anchor, mp) = 1
                    // offset-1 match: it runs to the end of the run (or matchlimit)
                    const int maxc = matchlimit - (ip + 4);
                    for (int scans = 0; scans <= n / 256 + 2; ++scans) {              // bounded: 256 bytes a step, the run ends at matchlimit
                        LV<int> len;
                        LV<bool> stop;
                        FOR_LANES(l) {
                            const int k = mcode + 4 * l;
                            int vb = maxc - k;
                            vb = vb < 0 ? 0 : (vb > 4 ? 4 : vb);
                            int ln = 0;
                            if (vb > 0) {
                                const uint32_t x = scan[l] ^ bbbb;
                                ln = x ? (int)(__builtin_ctz(x) >> 3) : 4;
                                if (ln > vb) ln = vb;
                            }
                            len[l] = ln;
                            stop[l] = ln < 4;
                        }
                        const uint64_t sm = ballot(stop);
                        if (sm) { const int f = ctz64(sm); mcode += 4 * f + readlane(len, f); break; }
                        mcode += 256;
                        FOR_LANES(l) { scan[l] = lds_ld32u(in, ip + 4 + mcode + 4 * l); }
                    }
                }
            };
            // ---- a match at ip with candidate mp: (unless the path that found it already did) extend it both ways with one LDS
            // round trip, park the sequence, set the next search up.  Called from every place a match is found, each of
            // which knows statically whether the extension is done -- no flags travel through the loop.
            // Returns 0: go on, 1: the plane is finished, 2: the output does not fit.
            const auto sequence = [&](auto extended_tag, int ip, int mp, int mcode, int backrun, const bool zero_lit) -> int {
            if constexpr (!decltype(extended_tag)::value) {
            CIMG_PROF_COUNT(3);
            const int room = zero_lit ? 0 : imin(ip - anchor, mp);
            const int maxc = matchlimit - (ip + 4);
                LV<bool> eq, stop;
                LV<int> len;
                // No guards around the loads: bytes past the plane end are readable (the hash table follows
                // it in LDS) and lanes past matchlimit are cut by the min with vb, so both directions go out
                // in ONE LDS round trip.
                FOR_LANES(l) {
                    const int kb = l < room ? l + 1 : 0;
                    const uint32_t pa = in[ip - kb], pb = in[mp - kb];
                    const int k = 4 * l;
                    const uint32_t x = lds_ld32u(in, ip + 4 + k) ^ lds_ld32u(in, mp + 4 + k);
                    eq[l] = (l < room) & (pa == pb);
                    const int ln = imin(x ? (int)(__builtin_ctz(x) >> 3) : 4, imax(maxc - k, 0));
                    len[l] = ln;
                    stop[l] = ln < 4;
                }
                backrun = ctz64(~ballot(eq));
                const uint64_t sm = ballot(stop);
                if (sm) {
                    const int f = ctz64(sm);
                    mcode = 4 * f + readlane(len, f);
                } else {
                    CIMG_PROF_COUNT(4);
                    mcode = match_more(in, ip, mp, maxc, 256, n);         // long match: keep counting, 256 bytes a step
                }
                if (backrun == 64) {                              // rare: more than 64 bytes backwards
                    int left = room - 64;
                    while (left > 0) {
                        FOR_LANES(l) { eq[l] = l < left && in[ip - 1 - backrun - l] == in[mp - 1 - backrun - l]; }
                        const int r = ctz64(~ballot(eq));
                        backrun += r; left -= r;
                        if (r < 64) break;
                    }
                }
            }   // not extended yet
            CIMG_PROF_LAP(4); CIMG_PROF_COUNT(2);               // match extension
            ip -= backrun; mp -= backrun; mcode += backrun;
            const int lit = zero_lit ? 0 : ip - anchor;
            // (the zero-literal chain pays above ~40 % hits of the post-match probe: the score follows that rate whether or not the chain runs)
#ifdef CIMG_ZERO_LIT_CHAIN
            chain_credit = zero_lit ? imin(chain_credit + 3, 8) : imax(chain_credit - (chain_credit > 0 ? 0 : 2), -8);
#endif
            // the ten bytes the scalar head of the NEXT search needs sit at the end of this match: request them now, so
            // that the LDS round trip runs behind the bookkeeping below instead of in front of the next search
            FOR_LANES(l) { Wn[l] = lds_ld32u(in, ip + mcode + 4 - 2 + 4 * (l < 2 ? l : 2)); }
            // ---- park the sequence; budget checks and stores happen 64 sequences at a time -------------------
            {
                const int slot = np;
                FOR_LANES(l) {
                    if (l == slot) { P_anchor[l] = anchor; P_lit[l] = lit; P_off[l] = ip - mp; P_mcode[l] = mcode; }
                }
                if (++np == 64) {
                    bool fits_;
                    if constexpr (SEQ) fits_ = zstd_take_parked(in, out, cap, op, *sink, np, P_anchor, P_lit, P_off, P_mcode);
                    else fits_ = emit_pending(in, out, cap, op, need, np, P_anchor, P_lit, P_off, P_mcode);
                    if (!fits_) return 2;
                    np = 0;
                }
            }
            ip += mcode + 4;
            anchor = ip;
            CIMG_PROF_LAP(5);                                   // budget checks + emit
            if (ip >= mflimit_p1) return 1;
            sstart = ip + 1;
            t0 = 0;
            pre = 1;
            return 0;
            };
#define CIMG_SEQUENCE_IN_HEAD(EXT) { const int r_ = sequence(std::EXT##_type{}, ip, mp, mcode, backrun, zero_lit); if (r_) { if (r_ == 2) ending = 0; stop = true; break; } continue; }
#define CIMG_SEQUENCE(EXT) { const int r_ = sequence(std::EXT##_type{}, ip, mp, mcode, backrun, zero_lit); if (r_) { if (r_ == 2) ending = 0; break; } continue; }
            // ---- scalar head ------------------------------------------------------------------------------------
            // Right after a match the next match is nearly always found by the post-match probe or one of the
            // first two probes of the new search (tiled family: 114 of 129 sequences, natural: 99 %).  Those three
            // need 10 input bytes: one LDS round trip brings them into scalars, and the run path / a three-probe
            // version of the narrow path below run without laying out a 64-probe window at all.
            // The head is a LOOP of its own: as long as it keeps finding the next match, sequence follows sequence in here and
            // none of the window state of the outer iteration is touched.
            bool headed = false, stop = false;
            if (pre && t0 == 0 && s64 == 64)                      // (every sequence leaves pre == 1, t0 == 0: only the range is tested again)
            while (mflimit_p1 - sstart >= 3) {
                if (++guard > 2 * n + 4) { ending = -1; stop = true; break; }
#ifdef CIMG_ZERO_LIT_CHAIN   /* measured: the chain itself runs a sequence in ~900 cycles against ~1600 through the head, but launches got no faster (LABNOTES.md, round 5) */
                // ---- zero-literal chain (round 5) -------------------------------------------------------------------------------
                // On sequence-dense data -- photographs -- the probe right after a match is the next match 60 - 95 % of the time (a
                // zero-literal sequence).  That case gets a loop of its own with nothing in it that it does not need: the refill, ONE
                // slot, the candidate's 256 bytes against the 256 bytes at the probe in one round trip (the comparison and the whole
                // extension: no literals, nothing backwards), the sequence parked.  The ten bytes the next turn starts from are the
                // bytes just compared, still in registers (no round trip for them).  About 75 instructions and two LDS round trips a
                // sequence against ~185 and four through the head below.  A miss leaves the table as it found it but for the refill
                // (which the head writes again: same slot, same value) and drops into the head.
                if (chain_credit > 0) {
                    int chained = 0;
                    CIMG_PROF_LAP(1);
                    uint32_t cw0 = readlane(Wn, 0), cw1 = readlane(Wn, 1);          // bytes [ip - 2, ip + 2), [ip + 2, ip + 6)
                    for (;;) {
                        if (mflimit_p1 - sstart < 3) break;                       // the last bytes of a plane go through the general paths
                        if (++guard > 2 * n + 4) { ending = -1; stop = true; break; }
                        const int ip0 = sstart - 1;
                        const uint32_t v0 = (cw0 >> 16) | (cw1 << 16);
                        const uint32_t h0 = lz4_hash<HB>(v0);
                        FOR_LANES_W(l) { tab16[lz4_hash<HB>(cw0)] = (uint16_t)(sstart - 3); }   // the "put(ip - 2)" refill
                        LV<uint32_t> slot;
                        FOR_LANES(l) { slot[l] = tab16[h0]; }
                        const int c0 = (int)readlane(slot, 0);
                        LV<uint32_t> cw, iw;
                        FOR_LANES(l) {
                            cw[l] = lds_ld32u(in, c0 + 4 * l);
                            iw[l] = lds_ld32u(in, ip0 + 4 * l);
                        }
                        needed_here(cw); needed_here(iw);                       // (one round trip: without this the bytes at the probe are loaded behind the branch)
                        if (readlane(cw, 0) != v0) { chain_credit -= chained ? 0 : 2; CIMG_PROF_LAP(0); CIMG_PROF_COUNT(4); break; }
                        FOR_LANES_W(l) { tab16[h0] = (uint16_t)ip0; }
                        CIMG_STAT(g_emu_matches);
                        const int maxc = matchlimit - (ip0 + 4);
                        LV<int> len;
                        LV<bool> stopl;
                        FOR_LANES(l) {
                            const int k = 4 * (l - 1);                              // lane 0 holds the four matched bytes
                            const uint32_t x = cw[l] ^ iw[l];
                            const int ln = imin(x ? (int)(__builtin_ctz(x) >> 3) : 4, imax(maxc - k, 0));
                            len[l] = ln;
                            stopl[l] = (l >= 1) & (ln < 4);
                        }
                        const uint64_t sm = ballot(stopl);
                        int mc;
                        if (sm) { const int f = ctz64(sm); mc = 4 * (f - 1) + readlane(len, f); }
                        else mc = match_more(in, ip0, c0, maxc, 252, n);
                        // park the sequence (no literals)
                        setlane(P_anchor, np, ip0);
                        setlane(P_lit, np, 0);
                        setlane(P_off, np, ip0 - c0);
                        setlane(P_mcode, np, mc);
                        if (++np == 64) {
                            bool fits_;
                            if constexpr (SEQ) fits_ = zstd_take_parked(in, out, cap, op, *sink, np, P_anchor, P_lit, P_off, P_mcode);
                            else fits_ = emit_pending(in, out, cap, op, need, np, P_anchor, P_lit, P_off, P_mcode);
                            if (!fits_) { ending = 0; stop = true; break; }
                            np = 0;
                        }
                        const int d = mc + 4;                                       // the match covers [ip0, ip0 + d)
                        CIMG_PROF_LAP(7); CIMG_PROF_COUNT(7);
                        anchor = ip0 + d;
                        chained = 1;
                        chain_credit = imin(chain_credit + 3, 8);
                        if (anchor >= mflimit_p1) { stop = true; break; }           // the plane is finished (ending stays 1)
                        sstart = anchor + 1;
                        // the bytes around the end of the match, from the dwords just compared (iw[k] = bytes [ip0 + 4 k, + 4))
                        if (d <= 249) {
                            const int a = (d - 2) >> 2;
                            const uint32_t sh = (uint32_t)(d - 2) & 3u;
                            const uint32_t q0 = readlane(iw, a), q1 = readlane(iw, a + 1), q2 = readlane(iw, a + 2);
                            cw0 = alignbyte(q1, q0, sh);
                            cw1 = alignbyte(q2, q1, sh);
                        } else {
                            LV<uint32_t> far;
                            FOR_LANES(l) { far[l] = lds_ld32u(in, anchor - 2 + 4 * (l < 2 ? l : 2)); }
                            cw0 = readlane(far, 0); cw1 = readlane(far, 1);
                        }
                    }
                    if (stop) break;
                    if (chained) {                                              // the head below reads its ten bytes from Wn
                        if (mflimit_p1 - sstart < 3) break;                     // (what the head's own loop condition says)
                        FOR_LANES(l) { Wn[l] = lds_ld32u(in, anchor - 2 + 4 * (l < 2 ? l : 2)); }
                    }
                }
#endif
                const int ip0 = sstart - 1;
                const uint32_t w0 = readlane(Wn, 0), w1 = readlane(Wn, 1);     // requested when the previous match was parked
                const uint32_t v0 = (w0 >> 16) | (w1 << 16), v1 = (w0 >> 24) | (w1 << 8), v2 = w1;   // bytes at ip0, ip0 + 1, ip0 + 2
                CIMG_PROF_LAP(1);
                if (v0 == v1) {
                    run_path(v0, w0);
                    CIMG_PROF_LAP(2); CIMG_PROF_COUNT(0);
                    CIMG_SEQUENCE_IN_HEAD(true)
                } else {
                    const uint32_t h0 = lz4_hash<HB>(v0), h1 = lz4_hash<HB>(v1), h2 = lz4_hash<HB>(v2);
                    // a probe whose hash equals an EARLIER probe's would have to see that probe's write: only hits before
                    // the first such probe are taken ("read all slots, then write" equals sequential LZ4 up to there)
                    const int clean = h1 == h0 ? 1 : ((h2 == h0 || h2 == h1) ? 2 : 3);    // probes 0 .. clean - 1 have pairwise different hashes
                    FOR_LANES_W(l) { tab16[lz4_hash<HB>(w0)] = (uint16_t)(sstart - 3); }
                    LV<uint32_t> hl, vl, old3;
                    LV<bool> hit3;
                    FOR_LANES(l) {
                        hl[l] = l == 0 ? h0 : (l == 1 ? h1 : h2);
                        vl[l] = l == 0 ? v0 : (l == 1 ? v1 : v2);
                        old3[l] = tab16[hl[l]];
                    }
                    FOR_LANES(l) { hit3[l] = (l < clean) & (lds_ld32u(in, (int)old3[l]) == vl[l]); }
                    const uint64_t hm = ballot(hit3);
                    if (hm) {
                        const int m3 = ctz64(hm);
                        FOR_LANES_W(l) { if (l <= m3) tab16[hl[l]] = (uint16_t)(ip0 + l); }
                        ip = ip0 + m3;
                        mp = (int)readlane(old3, m3);
                        zero_lit = m3 == 0;
                        CIMG_STAT(g_emu_matches);
                        CIMG_PROF_COUNT(7);
                        CIMG_PROF_LAP(7);
                        CIMG_SEQUENCE_IN_HEAD(false)
                    }
                    CIMG_PROF_LAP(7);
                }
                headed = true;                                    // three probes, no match: on to the window, which holds them again
                break;
            }
            if (stop) break;
            // ---- the head's third probe, once more ----------------------------------------------------------------
            // The scalar head only takes hits in front of the first probe that shares a table slot with an earlier one.
            // Sequential LZ4 goes on: such a probe has the earlier probe's POSITION as its candidate (the slot was just
            // written) and compares with that probe's four bytes.  The second probe cannot hit that way (v1 != v0 when the
            // head gets that far), the third can -- and in the high byte planes of an image it usually does: it sits one
            // byte into a run whose first byte was the second probe (14 of the 15 searches of a tiled plane that the head
            // gives up on).  Decided here from the ten bytes the head worked on (still in Wn), before any window is laid
            // out.  (Here and not inside the head loop: there the extra code cost the three-probe path 7 % on data that
            // never takes it.)
            if (headed) {
                const uint32_t w0 = readlane(Wn, 0), w1 = readlane(Wn, 1);
                const uint32_t v0 = (w0 >> 16) | (w1 << 16), v1 = (w0 >> 24) | (w1 << 8), v2 = w1;
                const uint32_t h0 = lz4_hash<HB>(v0), h1 = lz4_hash<HB>(v1), h2 = lz4_hash<HB>(v2);
                const bool e21 = h2 == h1;
                if ((e21 && v2 == v1) || (!e21 && h2 == h0 && v2 == v0)) {
                    // the three probes in order (the refill of ip - 2 was written by the head); of two sharing a slot the
                    // later one owns it
                    const int p0 = sstart - 1;
                    FOR_LANES_W(l) { tab16[h0] = (uint16_t)p0; }
                    FOR_LANES_W(l) { tab16[h1] = (uint16_t)(p0 + 1); }
                    FOR_LANES_W(l) { tab16[h2] = (uint16_t)(p0 + 2); }
                    ip = p0 + 2;
                    mp = e21 ? p0 + 1 : p0;
                    zero_lit = false;
                    CIMG_STAT(g_emu_matches);
                    CIMG_SEQUENCE(false)
                }
            }
            const int backpos = pre ? sstart - 3 : 0;             // the "put(ip - 2)" refill after a match
            const bool dense = t0 == 0 && s64 == 64;
            int nv = 64;                                          // valid lanes are a prefix
            if (dense) {
                // first window of a search at acceleration 1 (every window right after a match): probe t sits at
                // sstart + t with gap 1, so the valid prefix is known without looking at the lanes
                const int first = sstart - pre;
                nv = imin(64, pre + imax(mflimit_p1 - sstart, 0));    // the pre lane always; probe t while sstart + t + 1 <= mflimit_p1
                FOR_LANES(l) {
                    pos[l] = first + l;
                    valid[l] = l < nv;
                    v[l] = lds_ld32u(in, l < nv ? first + l : 0);
                    h[l] = lz4_hash<HB>(v[l]);
                    back[l] = lds_ld32u(in, backpos);
                }
            } else {
                FOR_LANES(l) {
                    const int t = t0 + l - pre;                   // probe number, -1 for the pre lane
                    // probe t sits at sstart + (t ? 1 + sum_{u<t-1} ((s64+u)>>6) : 0); the next one is one gap further
                    int p = sstart + t, gap = 1;
                    if (t > 0) { p = sstart + 1 + skip_prefix(s64 + t - 1) - f64; gap = (s64 + t - 1) >> 6; }
                    pos[l] = p;
                    valid[l] = t < 0 || p + gap <= mflimit_p1;
                    v[l] = lds_ld32u(in, valid[l] ? p : 0);
                    h[l] = lz4_hash<HB>(v[l]);
                    back[l] = lds_ld32u(in, backpos);
                }
                nv = popc64(ballot(valid));
            }
            CIMG_PROF_LAP(1);                                   // positions + v read
            if (nv == 0) break;                                   // -> last literals
            CIMG_STAT(g_emu_windows);
            // near the end of a stream the scalar head is not taken: the run path from the laid-out lanes
            if (pre && nv >= 2) {
                const uint32_t v0 = readlane(v, 0), v1 = readlane(v, 1);
                if (v0 == v1) {
                    run_path(v0, readlane(back, 0));
                    CIMG_PROF_LAP(2); CIMG_PROF_COUNT(0);
                    CIMG_SEQUENCE(true)
                }
            }
            // ---- narrow path ----------------------------------------------------------------------------------
            // In compressible data the next match is almost always found by the post-match probe or one of the
            // first probes of the new search.  With pairwise different hashes among those few probes, sequential
            // LZ4 and "read all slots, then write" agree, so the 64-lane collision machinery is not needed:
            // slots in one LDS round trip, candidates in a second, writes only for the probes actually consumed.
            if (!headed && pre && nv >= 4) {
                FOR_LANES_W(l) { tab16[lz4_hash<HB>(back[l])] = (uint16_t)backpos; }
                LV<uint32_t> h1, h2, h3, old4;
                lane_prev(h, h1);
                lane_prev(h1, h2);
                lane_prev(h2, h3);
                LV<bool> dup, hit4;
                FOR_LANES(l) {
                    old4[l] = tab16[h[l]];
                    dup[l] = ((l >= 1) & (h[l] == h1[l])) | ((l >= 2) & (h[l] == h2[l])) | ((l >= 3) & (h[l] == h3[l]));
                }
                // candidates of the four probes, and in the same round trip the 256 bytes that extend a match of the
                // post-match probe (lane 0, by far the most frequent hit: it has no literals, so nothing backwards)
                const int c0 = (int)readlane(old4, 0), p0 = sstart - 1;
                LV<uint32_t> cw, iw;
                FOR_LANES(l) {
                    hit4[l] = (l < 4) & (lds_ld32u(in, (int)old4[l]) == v[l]);
                    cw[l] = lds_ld32u(in, c0 + 4 + 4 * l);
                    iw[l] = lds_ld32u(in, p0 + 4 + 4 * l);
                }
                needed_here(cw); needed_here(iw);                 // (one round trip with the candidates: see run_path)
                const uint64_t hm = ballot(hit4);
                if (hm) {
                    const int m4 = ctz64(hm);
                    if (!(ballot(dup) & ((2ull << m4) - 1))) {
                        FOR_LANES_W(l) { if (l <= m4) tab16[h[l]] = (uint16_t)pos[l]; }
                        ip = readlane(pos, m4);
                        mp = (int)readlane(old4, m4);
                        zero_lit = m4 == 0;
                        CIMG_STAT(g_emu_matches);
                        CIMG_PROF_COUNT(7);
                        if (m4 == 0) {
                            backrun = 0;
                            const int maxc = matchlimit - (p0 + 4);
                            LV<int> len;
                            LV<bool> stop;
                            FOR_LANES(l) {
                                const uint32_t x = cw[l] ^ iw[l];
                                const int ln = imin(x ? (int)(__builtin_ctz(x) >> 3) : 4, imax(maxc - 4 * l, 0));
                                len[l] = ln;
                                stop[l] = ln < 4;
                            }
                            const uint64_t sm = ballot(stop);
                            if (sm) { const int f = ctz64(sm); mcode = 4 * f + readlane(len, f); }
                            else mcode = match_more(in, p0, c0, maxc, 256, n);
                            CIMG_PROF_LAP(7);
                            CIMG_SEQUENCE(true)
                        }
                        CIMG_PROF_LAP(7);
                        CIMG_SEQUENCE(false)
                    }
                }
                CIMG_PROF_LAP(7);
            }
            if (pre) { FOR_LANES_W(l) { tab16[lz4_hash<HB>(back[l])] = (uint16_t)backpos; } }
            // a lane with the same hash as its left neighbour has that neighbour as candidate
            LV<uint32_t> ph, pv;
            LV<int> ppos;
            lane_prev(h, ph);
            lane_prev(v, pv);
            lane_prev(pos, ppos);
            LV<bool> head, cont;
            LV<uint32_t> old;
            FOR_LANES(l) {
                cont[l] = valid[l] && l > 0 && h[l] == ph[l];
                head[l] = valid[l] && !cont[l];
                old[l] = tab16[h[l]];                              // every lane reads (harmless), only heads use it
            }
            FOR_LANES_W(l) { if (head[l]) tab16[h[l]] = (uint16_t)pos[l]; }
            LV<bool> loser, hit;
            LV<int> cand;
            FOR_LANES(l) {
                const uint32_t rb = tab16[h[l]];
                const uint32_t mvh = lds_ld32u(in, (int)old[l]);
                const uint32_t mv = head[l] ? mvh : pv[l];
                loser[l] = head[l] && rb != (uint32_t)pos[l];
                hit[l] = valid[l] && mv == v[l];
                cand[l] = head[l] ? (int)old[l] : ppos[l];
            }
            uint64_t involved = ballot(loser);
            if (involved) {
                CIMG_STAT(g_emu_collisions);
                // heads that lost a slot write again; a winner whose slot changes has company too
                FOR_LANES_W(l) { if (loser[l]) tab16[h[l]] = (uint16_t)pos[l]; }
                LV<bool> inv;
                FOR_LANES(l) { inv[l] = head[l] && (loser[l] || tab16[h[l]] != (uint16_t)pos[l]); }
                involved = ballot(inv);
            }
            const int k1 = ctz64(involved);                           // 64 if no collision
            const int limit = imin(nv - 1, k1);
            const uint64_t below = limit >= 63 ? ~0ull : ((1ull << (limit + 1)) - 1);
            const uint64_t hits = ballot(hit) & below;
            const int m = hits ? ctz64(hits) : -1;
            const int B = m >= 0 ? m : limit;
            const uint64_t headmask = ballot(head), contmask = ballot(cont);
            const uint64_t above = B >= 63 ? 0ull : (~0ull << (B + 1));
            if (headmask & above) {
                FOR_LANES_W(l) { if (head[l] && l > B) tab16[h[l]] = (uint16_t)old[l]; }
                FOR_LANES_W(l) { if (head[l] && l == B) tab16[h[l]] = (uint16_t)pos[l]; }
            }
            if (contmask & ~above) {
                // the last lane of every committed run of equal hashes owns the slot
                FOR_LANES_W(l) {
                    if (cont[l] && l <= B && (l == B || !((contmask >> ((l + 1) & 63)) & 1) || l == 63)) tab16[h[l]] = (uint16_t)pos[l];
                }
            }
            CIMG_PROF_LAP(3); CIMG_PROF_COUNT(1);               // window machinery
            if (m < 0) {
                if (B + 1 < 64 && B + 1 >= nv) break;             // the next probe would pass mflimit
                t0 += B + 1 - pre;
                pre = 0;
                continue;
            }
            CIMG_STAT(g_emu_matches);
            ip = readlane(pos, m);
            mp = readlane(cand, m);
            zero_lit = pre && m == 0;
            CIMG_SEQUENCE(false)
        }
#undef CIMG_SEQUENCE
#undef CIMG_SEQUENCE_IN_HEAD
        if (ending <= 0) { CIMG_PROF_LAP(5); CIMG_PROF_STORE(dbg, item); return ending; }
    }
    if constexpr (SEQ) {
        if (np && !zstd_take_parked(in, out, cap, op, *sink, np, P_anchor, P_lit, P_off, P_mcode)) return 0;
        // the bytes behind the last match close the literal area
        const int run = n - anchor;
        if (op + run > cap) return 0;
        emit_literals(in, anchor, out, op, run);
        sink->lit_total = op + run;
        need_out = 0;
        return sink->nseq;
    }
    if (np && !emit_pending(in, out, cap, op, need, np, P_anchor, P_lit, P_off, P_mcode)) { CIMG_PROF_LAP(5); CIMG_PROF_STORE(dbg, item); return 0; }
    // ---- last literals ------------------------------------------------------------------------------------
    {
        const int run = n - anchor;
        const int lhs = op + run + 1 + (run + 240) / 255;
        if (lhs > cap) { CIMG_PROF_LAP(6); CIMG_PROF_STORE(dbg, item); return 0; }
        need = imax(need, lhs);
        const uint32_t token = (uint32_t)((run >= 15 ? 15 : run) << 4);
        FOR_LANES(l) { if (l == 0) out[op] = (uint8_t)token; }
        op++;
        if (run >= 15) { emit_len_ext(out, op, run - 15); op += (run - 15) / 255 + 1; }
        emit_literals(in, anchor, out, op, run);
        op += run;
    }
    CIMG_PROF_LAP(6);                                       // last literals
    CIMG_PROF_STORE(dbg, item);
    need_out = need;
    return op;
}

// Out of line on the GPU: the persistent workgroup loop around it keeps a lot of scalar state alive, and
// inlining this body there pushed the kernel into SGPR spilling.  lds_passed is only used by the emulator.
CIMG_DEV_NOINLINE int lz4_encode_wave(uint8_t* lds_passed, int in_off, int tab_off, int n, uint8_t* out, int cap, int accel, int* need_ptr, uint64_t* dbg = nullptr, int item = 0)
{
    uint8_t* const lds_base = CIMG_LDS_BASE(lds_passed);
    const uint8_t* in = lds_base + in_off;
    uint8_t* tab = lds_base + tab_off;
    int need_out = 0;
    const int result = lz4_encode_body(in, tab, n, out, cap, accel, need_out, dbg, item);
    *need_ptr = need_out;
    return result;
}

}  // namespace cimg
#include "encode_rt_kernel.h"   // the same encoder with its hash table in registers (round 5): EncodeStream<CODEC_LZ4_RT>
namespace cimg {

// EncodeStream's template argument for the register-table form of the LZ4 encoder (not a blosc2 codec id: the chunks are LZ4 / LZ4HC
// chunks, byte for byte those of EncodeStream<CODEC_LZ4>)
enum : int { CODEC_LZ4_RT = 101 };

// true if all n bytes of the LDS plane equal its first byte
CIMG_DEV bool plane_is_run(const uint8_t* in, int n, uint32_t& value)
{
    LV<uint32_t> first;
    FOR_LANES(l) { first[l] = in[0]; }
    value = readlane(first, 0);
    const uint32_t w = value * 0x01010101u;
    // 1 KiB per step (16 bytes per lane); almost every plane leaves at the first step, constant channels (alpha,
    // masks) run through all of them
    const int units = n >> 4;
    for (int c = 0; c < units; c += 64) {
        LV<bool> bad;
        FOR_LANES(l) {
            const u128 q = ld128a(in + 16 * (c + l < units ? c + l : 0));
            bad[l] = (c + l < units) & ((q.x != w) | (q.y != w) | (q.z != w) | (q.w != w));
        }
        if (ballot(bad)) return false;
    }
    LV<bool> bad;
    FOR_LANES(l) { bad[l] = 16 * units + l < n && in[16 * units + l] != (uint8_t)value; }
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

// ---- chunks assembled inside the encode launch (EncodeArgs::assemble) ----------------------------------------------------------
// Out of line on the GPU: these run once per work item / once per wave, outside the codec loop, and inlined they cost the
// codec loop registers (the LZ4 kernel sits at the 256-VGPR ceiling: with them inlined it spilled to scratch).
CIMG_DEV AssembleArgs assemble_args(kernarg_ptr<EncodeArgs> ap)
{
    const auto a = fresh(ap);
    AssembleArgs aa;
    aa.descs = a->descs; aa.nchunks = a->nchunks; aa.raw = a->raw;
    aa.p.typesize = a->p.typesize; aa.p.clevel = a->p.clevel; aa.p.compcode = a->p.compcode; aa.p.filter = a->p.filter; aa.p.accel = a->p.accel;
    aa.p.max_blocksize = a->p.max_blocksize; aa.p.slot_bytes = a->p.slot_bytes; aa.p.streams_per_block = a->p.streams_per_block;
    aa.scratch = a->scratch; aa.recs = a->recs;
    aa.comp = a->comp; aa.layout = a->layout; aa.uniform_nblocks = a->uniform_nblocks; aa.layout_host = a->layout_host; aa.skip_assembled = 0;
    return aa;
}

// block and stream of a work item (encode_items)
CIMG_DEV void encode_item_place(kernarg_ptr<EncodeArgs> a, int item, int& b, int& s)
{
    const int whole = a->want_split ? a->block_items : a->total_blocks;
    if (item < whole) {
        b = item; s = 0;
    } else {
        const int rest = a->total_blocks - whole, idx = item - whole;
        b = whole + idx % rest;
        s = a->p.streams_per_block - 1 - idx / rest;
    }
}

// `finished` streams of `chunk` are done: their payloads sit in the scratch slots (plain stores: only this wave reads them again),
// their records were written through to memory (agent-scope stores).  Counts them; the wave whose streams complete the chunk lays
// it out and raises ready[chunk].  Returns the new head of the wave's item list (the item is only remembered when its chunk is
// assembled in the launch).
CIMG_DEV_OUTLINE int encode_account(kernarg_ptr<EncodeArgs> ap_in, int item_in, int last_in, int chunk_in, int finished_in)
{
    // (a real call passes its arguments in vector registers: the pointer and the numbers are made scalar again)
    const kernarg_ptr<EncodeArgs> ap = CIMG_OWN_KERNARGS(EncodeArgs, ap_in);
    const int item = uni(item_in), last = uni(last_in), chunk = uni(chunk_in), finished = uni(finished_in);
    uint32_t* cnt;
    int nstreams;
    {
        const auto a = fresh(ap);
        if (!uni(a->descs[chunk].assemble)) return last;
        cnt = a->chunk_count + chunk;
        nstreams = uni(a->descs[chunk].nstreams);
        int32_t* nx = a->next_item + item;
        FOR_LANES_W(l) { if (l == 0) *nx = last; }              // read back by this wave only
    }
    stores_performed();                                      // the records (written through) before the count says so
    LV<uint32_t> got;
    FOR_LANES(l) { got[l] = 0; }
    FOR_LANES_W(l) { if (l == 0) got[l] = atomic_add_agent(cnt, (uint32_t)finished); }
    const int before = uni((int)readlane(got, 0));
    if (before + finished != nstreams) return item;
    {
        const AssembleArgs aa = assemble_args(ap);
        LayoutChunk lc(aa, chunk);
        const int64_t comp_off = uni64(aa.descs[chunk].comp_off);
        lc.through = (((uintptr_t)aa.comp + (uintptr_t)comp_off) & 3) == 0;      // bstarts[] on a 4-byte boundary: written through
        if (!lc.through) fence_acquire();                    // (else everybody's records are read past the L2: LayoutChunk::run)
        lc.run();
        FOR_LANES_W(l) { if (l == 0) atomic_store_agent(cnt, 0u); }  // the next batch counts from zero again
        // header, bstarts, layout[chunk] before the flag says so: written through and acknowledged -- or, from an odd address,
        // written back by an L2 flush
        if (lc.through) stores_performed(); else fence_release();
    }
    const auto a = fresh(ap);
    uint32_t* flag = a->ready + chunk;
    const uint32_t gen = a->gen;
    FOR_LANES_W(l) { if (l == 0) atomic_store_agent(flag, gen); }
    CIMG_ITEM_STAMP(a->dbg, item, 1);                            // diagnostics (tools/diag_assemble.py): this item's wave closed a chunk
    return item;
}

// The wave found the work queue empty: it copies the streams IT encoded into place, chunk by chunk as they become ready.  A chunk
// that is not laid out yet is waited for (s_sleep + one atomic load per try; every chain that still encodes keeps its own SIMD
// slot, so waiting waves hold nobody up) -- with a hard bound, after which the batch FAILS.
CIMG_DEV_OUTLINE void encode_emit_own(kernarg_ptr<EncodeArgs> ap_in, int last_in)
{
    const kernarg_ptr<EncodeArgs> ap = CIMG_OWN_KERNARGS(EncodeArgs, ap_in);
    const int last = uni(last_in);
    const AssembleArgs aa = assemble_args(ap);
    uint32_t* ready;
    int32_t* next_item;
    uint32_t gen;
    int items, want_split, ts_arg;
    {
        const auto a = fresh(ap);
        ready = a->ready; next_item = a->next_item; gen = a->gen; want_split = a->want_split; ts_arg = a->p.typesize;
        items = encode_items(a->total_blocks, a->p.streams_per_block, a->want_split != 0, a->block_items);
    }
    // Pass after pass over the wave's own items: an item whose chunk is laid out is copied into place and marked; the others are
    // looked at again in the next pass (a nap in between).  Most of a wave's items belong to chunks that were finished long
    // before it ran out of work; only the chunks that close at the very end of the launch are waited for.
    enum : int { PASS_LIMIT = 1 << 20, ITEM_DONE = 1 << 30 };   // x (s_sleep + the polls of a pass) is seconds: only a lost chunk gets there
    int known_ready = -1, known_not = -1;                    // per pass: the chunks last seen ready / not ready (a wave's items share chunks)
    bool pending = true;
    for (int pass = 0; pass < PASS_LIMIT && pending; ++pass) {
        pending = false;
        known_not = -1;
        int it = last;
        for (int n = 0; n <= items && it >= 0; ++n) {        // bounded: a wave cannot own more than every item
            LV<int32_t> nx;
            FOR_LANES(l) { nx[l] = next_item[it]; }
            const int link = uni(readlane(nx, 0));
            const bool marked = link >= 0 && (link & ITEM_DONE) != 0;                   // (the end of the list, -1, is not a mark: marks are non-negative)
            const int next = marked ? (link & ~ITEM_DONE) - 1 : link;                   // a marked link holds (next + 1) | ITEM_DONE
            if (!marked) {
                int b, s;
                encode_item_place(fresh(ap), it, b, s);
                const int chunk = find_chunk(aa.descs, aa.nchunks, b, aa.uniform_nblocks);
                bool is_ready = chunk == known_ready;
                if (!is_ready && chunk != known_not) {
                    LV<uint32_t> got;
                    FOR_LANES(l) { got[l] = 0; }
                    FOR_LANES_W(l) { if (l == 0) got[l] = atomic_load_agent(ready + chunk); }
                    is_ready = uni(readlane(got, 0)) == gen;
                    if (is_ready) known_ready = chunk;               // (what the closer and other waves wrote is read past the L2: EmitBlock::run_streams)
                    else known_not = chunk;
                }
                if (is_ready) {
                    const ChunkDesc d = uniform_desc(aa.descs + chunk);
                    const int j = b - d.blk0;
                    const bool leftover_blk = (j == d.nblocks - 1 && d.leftover);
                    const int ns = (d.split && !leftover_blk) ? ts_arg : 1;
                    const bool whole = ns > 1 && want_split && it < fresh(ap)->block_items;
                    EmitBlock eb(aa, b);
                    if (whole || ns == 1) eb.run_streams(0, ns);
                    else eb.run_streams(s, s + 1);
                    FOR_LANES_W(l) { if (l == 0) next_item[it] = (next + 1) | ITEM_DONE; }
                    CIMG_ITEM_STAMP(fresh(ap)->dbg, it, 2);                 // diagnostics: the item's streams are in place
                } else {
                    pending = true;
                }
            }
            it = next;
        }
        if (pending) wave_nap();
    }
    if (pending) { FOR_LANES_W(l) { if (l == 0 && aa.layout_host) aa.layout_host[aa.nchunks].cbytes = -1; } }
}

// the next item of the launch for wave w, or -1 when every sub-queue is dry ("the work queue" at the top of this file).  Out of line:
// it runs once per work item, and its vector loads must not count against the codec loop's registers.
CIMG_DEV_OUTLINE int encode_pop_item(uint32_t* heads_in, int items_in, int w_in)
{
    uint32_t* const heads = reinterpret_cast<uint32_t*>(uni64((int64_t)reinterpret_cast<uintptr_t>(heads_in)));
    const int items = uni(items_in), w = uni(w_in);
    const int myq = w & (ENC_NQ - 1);
    {
        const int cnt = (items - myq + ENC_NQ - 1) / ENC_NQ;             // items of sub-queue myq (<= 0: none)
        LV<uint32_t> got;
        FOR_LANES(l) { got[l] = 0; }
        FOR_LANES_W(l) { if (l == 0) got[l] = queue_pop(heads + myq * ENC_QSTRIDE); }
        const uint32_t v = uni(readlane(got, 0));
        if ((int64_t)v < (int64_t)cnt) return myq + ENC_NQ * (int)v;
    }
    // its own sub-queue is dry: look at all heads (one vector load), take from the next one that still has items
    for (int tries = 0; tries < ENC_NQ; ++tries) {
        LV<uint32_t> h;
        LV<bool> has;
        FOR_LANES(l) {
            h[l] = atomic_load_agent(heads + l * ENC_QSTRIDE);
            const int cnt_l = (items - l + ENC_NQ - 1) / ENC_NQ;
            has[l] = (int64_t)h[l] < (int64_t)cnt_l;
        }
        const uint64_t m = ballot(has);
        if (!m) return -1;
        const uint64_t behind = myq < 63 ? m >> (myq + 1) << (myq + 1) : 0;   // the thieves spread: each starts behind its own sub-queue
        const int q = ctz64(behind ? behind : m);
        const int cnt = (items - q + ENC_NQ - 1) / ENC_NQ;
        LV<uint32_t> got;
        FOR_LANES(l) { got[l] = 0; }
        FOR_LANES_W(l) { if (l == 0) got[l] = queue_pop(heads + q * ENC_QSTRIDE); }
        const uint32_t v = uni(readlane(got, 0));
        if ((int64_t)v < (int64_t)cnt) return q + ENC_NQ * (int)v;
    }
    return -1;
}

// one single-wave workgroup = one stream.  CODEC selects the stream codec at compile time: the two encoders live in
// two kernels (merged into one, the scalar state of both pushed the LZ4 kernel from 38 to 84 spilled SGPRs)
template <int CODEC>
struct EncodeStream {
    kernarg_ptr<EncodeArgs> ap;     // wave.h: fields are loaded where they are used, not held in SGPRs across the codec loop
    uint8_t* lds;
    int w;

    CIMG_DEV EncodeStream(kernarg_ptr<EncodeArgs> a_, uint8_t* lds_, int w_) : ap(a_), lds(lds_), w(w_) {}

    // phase A for a split block: keep byte plane s (or slice s when no shuffle) of the block
    CIMG_DEV void load_plane(const uint8_t* src, int bsize, int ts, int s, int neblock, bool shuf)
    {
        if (!shuf) {
            wave_copy_g2l(src + (int64_t)s * neblock, lds, 0, neblock);
            return;
        }
        const int units = bsize >> 4;                    // 16 source bytes = 16/ts plane bytes
        // A single wave hides HBM latency only through loads in flight: 16 x 16 B per lane are requested
        // before the first is used (a 32 KiB block is two such rounds instead of eight rounds of four).
        constexpr int DEPTH = 16;
        if (ts == 2) {
            const uint32_t sel = s ? 0x07050301u : 0x06040200u;
            int u0 = 0;
            for (; u0 + 64 * DEPTH <= units; u0 += 64 * DEPTH) {
                LV<u128> x[DEPTH];
                CIMG_UNROLL
                for (int k = 0; k < DEPTH; k++) { FOR_LANES(l) { x[k][l] = ld128u(src + 16 * (u0 + 64 * k + l)); } }
                CIMG_UNROLL
                for (int k = 0; k < DEPTH; k++) {
                    FOR_LANES(l) {
                        uint32_t* d = reinterpret_cast<uint32_t*>(lds + 8 * (u0 + 64 * k + l));
                        d[0] = byte_perm(x[k][l].y, x[k][l].x, sel);
                        d[1] = byte_perm(x[k][l].w, x[k][l].z, sel);
                    }
                }
            }
            for (; u0 < units; u0 += 64) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        const u128 x = ld128u(src + 16 * u);
                        uint32_t* d = reinterpret_cast<uint32_t*>(lds + 8 * u);
                        d[0] = byte_perm(x.y, x.x, sel);
                        d[1] = byte_perm(x.w, x.z, sel);
                    }
                }
            }
        } else if (ts == 4) {
            const uint32_t s1 = (s & 2) ? 0x07030602u : 0x05010400u;      // bytes {0,1}/{2,3} of two elements
            const uint32_t s2 = (s & 1) ? 0x07060302u : 0x05040100u;      // then byte s of four elements
            int u0 = 0;
            for (; u0 + 64 * DEPTH <= units; u0 += 64 * DEPTH) {
                LV<u128> x[DEPTH];
                CIMG_UNROLL
                for (int k = 0; k < DEPTH; k++) { FOR_LANES(l) { x[k][l] = ld128u(src + 16 * (u0 + 64 * k + l)); } }
                CIMG_UNROLL
                for (int k = 0; k < DEPTH; k++) {
                    FOR_LANES(l) {
                        const uint32_t t = byte_perm(x[k][l].y, x[k][l].x, s1), q = byte_perm(x[k][l].w, x[k][l].z, s1);
                        *reinterpret_cast<uint32_t*>(lds + 4 * (u0 + 64 * k + l)) = byte_perm(q, t, s2);
                    }
                }
            }
            for (; u0 < units; u0 += 64) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        const u128 x = ld128u(src + 16 * u);
                        const uint32_t t = byte_perm(x.y, x.x, s1), q = byte_perm(x.w, x.z, s1);
                        *reinterpret_cast<uint32_t*>(lds + 4 * u) = byte_perm(q, t, s2);
                    }
                }
            }
        } else {
            for (int e0 = 0; e0 < neblock; e0 += 64) {
                FOR_LANES(l) { if (e0 + l < neblock) lds[e0 + l] = src[(int64_t)(e0 + l) * ts + s]; }
            }
        }
        // bytes of the block past the last whole 16-byte unit
        const int e_done = (units << 4) / ts;
        if ((ts == 2 || ts == 4) && e_done < neblock) {
            FOR_LANES(l) { if (e_done + l < neblock) lds[e_done + l] = src[(int64_t)(e_done + l) * ts + s]; }
        }
    }

    // phase A of a block item: ONE pass over the 32 KiB block.  The most significant byte plane goes to LDS (it is encoded
    // first), the others stay in registers: keep[p][k] holds dword(s) k of plane p as this lane will write them to LDS later.
    //   typesize 2: unit u = 16 source bytes = 8 bytes of each plane; lane l handles units 64 k + l, k = 0 .. 31
    //   typesize 4: unit u = 16 source bytes = 4 bytes of each plane
    CIMG_DEV void load_block_ts2(const uint8_t* src, LV<uint32_t> (&keep)[96])
    {
        constexpr int DEPTH = 16;
        CIMG_UNROLL
        for (int r = 0; r < 2; r++) {
            LV<u128> x[DEPTH];
            CIMG_UNROLL
            for (int k = 0; k < DEPTH; k++) { FOR_LANES(l) { x[k][l] = ld128u(src + 16 * (64 * (DEPTH * r + k) + l)); } }
            CIMG_UNROLL
            for (int k = 0; k < DEPTH; k++) {
                FOR_LANES(l) {
                    const int u = 64 * (DEPTH * r + k) + l;
                    uint32_t* d = reinterpret_cast<uint32_t*>(lds + 8 * u);
                    d[0] = byte_perm(x[k][l].y, x[k][l].x, 0x07050301u);           // plane 1 (high bytes) -> LDS
                    d[1] = byte_perm(x[k][l].w, x[k][l].z, 0x07050301u);
                    keep[2 * (DEPTH * r + k)][l] = byte_perm(x[k][l].y, x[k][l].x, 0x06040200u);        // plane 0 -> registers
                    keep[2 * (DEPTH * r + k) + 1][l] = byte_perm(x[k][l].w, x[k][l].z, 0x06040200u);
                }
            }
        }
    }
    CIMG_DEV void load_block_ts4(const uint8_t* src, LV<uint32_t> (&keep)[96])
    {
        constexpr int DEPTH = 16;
        CIMG_UNROLL
        for (int r = 0; r < 2; r++) {
            LV<u128> x[DEPTH];
            CIMG_UNROLL
            for (int k = 0; k < DEPTH; k++) { FOR_LANES(l) { x[k][l] = ld128u(src + 16 * (64 * (DEPTH * r + k) + l)); } }
            CIMG_UNROLL
            for (int k = 0; k < DEPTH; k++) {
                FOR_LANES(l) {
                    const int u = 64 * (DEPTH * r + k) + l;
                    // bytes {0,1} / {2,3} of two elements, then byte s of four elements
                    const uint32_t lo01 = byte_perm(x[k][l].y, x[k][l].x, 0x05010400u), hi01 = byte_perm(x[k][l].w, x[k][l].z, 0x05010400u);
                    const uint32_t lo23 = byte_perm(x[k][l].y, x[k][l].x, 0x07030602u), hi23 = byte_perm(x[k][l].w, x[k][l].z, 0x07030602u);
                    *reinterpret_cast<uint32_t*>(lds + 4 * u) = byte_perm(hi23, lo23, 0x07060302u);              // plane 3 -> LDS
                    keep[DEPTH * r + k][l] = byte_perm(hi01, lo01, 0x05040100u);                                 // plane 0
                    keep[32 + DEPTH * r + k][l] = byte_perm(hi01, lo01, 0x07060302u);                            // plane 1
                    keep[64 + DEPTH * r + k][l] = byte_perm(hi23, lo23, 0x05040100u);                            // plane 2
                }
            }
        }
    }
    // a kept plane -> LDS (the wave is done with the plane that was there)
    CIMG_DEV void restore_plane(const LV<uint32_t> (&keep)[96], int ts, int s)
    {
        if (ts == 2) {
            CIMG_UNROLL
            for (int k = 0; k < 32; k++) {
                FOR_LANES(l) {
                    uint32_t* d = reinterpret_cast<uint32_t*>(lds + 8 * (64 * k + l));
                    d[0] = keep[2 * k][l];
                    d[1] = keep[2 * k + 1][l];
                }
            }
        } else {
            CIMG_UNROLL
            for (int k = 0; k < 32; k++) {
                FOR_LANES(l) {
                    const uint32_t v = s == 0 ? keep[k][l] : (s == 1 ? keep[32 + k][l] : keep[64 + k][l]);
                    *reinterpret_cast<uint32_t*>(lds + 4 * (64 * k + l)) = v;
                }
            }
        }
    }

    // bitshuffle filter: the first ne8 = ne - ne % 8 elements become 8 * ts bit rows of
    // ne8 / 8 bytes -- row 8 j + k holds bit k of byte j of every element, element i at bit i % 8 of byte i / 8 --
    // and the remaining bytes are copied.  A lane takes one group of 8 elements: 8 * ts contiguous source bytes,
    // one 8 x 8 bit transpose per byte position, 8 * ts single-byte LDS stores.
    CIMG_DEV void load_block_bitshuffle(const uint8_t* src, int bsize, int ts)
    {
        const int ne = bsize / ts, ne8 = ne & ~7, rowbytes = ne8 >> 3;
        for (int g0 = 0; g0 < rowbytes; g0 += 64) {
            FOR_LANES(l) {
                const int g = g0 + l;
                if (g < rowbytes) {
                    const uint8_t* e = src + (int64_t)8 * g * ts;
                    if (ts == 2) {
                        const u128 x = ld128u(e);
                        for (int j = 0; j < 2; j++) {
                            const uint32_t sel = j ? 0x07050301u : 0x06040200u;
                            const uint64_t y = bit_transpose8((uint64_t)byte_perm(x.y, x.x, sel) | ((uint64_t)byte_perm(x.w, x.z, sel) << 32));
                            for (int k = 0; k < 8; k++) lds[(8 * j + k) * rowbytes + g] = (uint8_t)(y >> (8 * k));
                        }
                    } else if (ts == 4) {
                        const u128 x = ld128u(e), z = ld128u(e + 16);
                        for (int j = 0; j < 4; j++) {
                            const uint32_t s1 = (j & 2) ? 0x07030602u : 0x05010400u;
                            const uint32_t s2 = (j & 1) ? 0x07060302u : 0x05040100u;
                            const uint32_t lo = byte_perm(byte_perm(x.w, x.z, s1), byte_perm(x.y, x.x, s1), s2);
                            const uint32_t hi = byte_perm(byte_perm(z.w, z.z, s1), byte_perm(z.y, z.x, s1), s2);
                            const uint64_t y = bit_transpose8((uint64_t)lo | ((uint64_t)hi << 32));
                            for (int k = 0; k < 8; k++) lds[(8 * j + k) * rowbytes + g] = (uint8_t)(y >> (8 * k));
                        }
                    } else {
                        for (int j = 0; j < ts; j++) {
                            uint64_t x = 0;
                            for (int i = 0; i < 8; i++) x |= (uint64_t)e[i * ts + j] << (8 * i);
                            const uint64_t y = bit_transpose8(x);
                            for (int k = 0; k < 8; k++) lds[(8 * j + k) * rowbytes + g] = (uint8_t)(y >> (8 * k));
                        }
                    }
                }
            }
        }
        const int done = ne8 * ts;
        for (int k0 = done; k0 < bsize; k0 += 64) {
            FOR_LANES(l) { if (k0 + l < bsize) lds[k0 + l] = src[k0 + l]; }
        }
    }

    // phase A for an unsplit block: the whole filtered block is the stream
    CIMG_DEV void load_block(const uint8_t* src, int bsize, int ts, bool shuf)
    {
        if (!shuf) { wave_copy_g2l(src, lds, 0, bsize); return; }
        const int ne = bsize / ts;
        for (int k0 = 0; k0 < bsize; k0 += 64) {
            FOR_LANES(l) {
                const int k = k0 + l;
                if (k < bsize) {
                    const uint8_t x = src[k];
                    if (k < ne * ts) lds[(k % ts) * ne + k / ts] = x; else lds[k] = x;
                }
            }
        }
    }

    // persistent workgroup: pull items until the queue is dry, then -- when the batch is assembled in place -- copy this wave's own
    // streams into place as their chunks close
    CIMG_DEV void run()
    {
        int items, assemble;
        uint32_t* queue;
        {
            const auto a = fresh(ap);
            items = encode_items(a->total_blocks, a->p.streams_per_block, a->want_split != 0, a->block_items);
            queue = a->queue;
            assemble = a->assemble;
            if (w == 0 && a->queue_next) {                       // the heads of the next launch of this kind
                uint32_t* const nx = a->queue_next;
                FOR_LANES_W(l) { atomic_store_agent(nx + l * ENC_QSTRIDE, 0u); }
            }
        }
        // bounded: a wave can never pop more than every item plus its final empty-queue pop
        int last = -1;                                           // the items this wave encoded, newest first (next_item[])
        for (int pops = 0; pops <= items + 1; ++pops) {
            const int item = uni(encode_pop_item(queue, items, w));
            if (item < 0) break;
            int chunk = 0;
            const int finished = run_item(item, chunk);
            if (assemble && finished > 0) last = encode_account(ap, item, last, chunk, finished);
        }
        if (assemble && last >= 0) encode_emit_own(ap, last);
    }

    // run check + codec on the stream that sits in LDS; leaves payload in the scratch slot and the record in recs
    CIMG_DEV void encode_resident(int rec_index, int neblock, uint8_t* out, int accel_or_level, uint64_t* dbg, int item)
    {
        const uint8_t* in = lds;
        StreamRec r;
        r.kind = REC_RAW; r.value = 0; r.csize = neblock; r.need = 0;
        uint32_t value;
        if (plane_is_run(in, neblock, value)) {
            r.kind = REC_RUN; r.value = (int32_t)value; r.csize = 0;
        } else {
            int need = 0;
            int cb;
            if constexpr (CODEC == CODEC_BLOSCLZ) cb = blosclz_encode_body(lds, lds + round16(neblock) + 16, neblock, out, neblock, accel_or_level, need);
            else if constexpr (CODEC == CODEC_ZSTD) {
                // one zstd frame (zstd_encode.h): the match finder's sequences go to this wave's record area, the literal bytes
                // straight to their place in the frame, then the FSE-coded sequences section is built in LDS and appended
                SeqSink sink;
                const ZstdEncTables* tabs;
                {
                    const auto a = fresh(ap);
                    sink.seq = a->zstd_seq + (size_t)w * (size_t)a->zstd_seq_stride;
                    tabs = a->zstd_tables;
                }
                sink.nseq = 0; sink.lit_total = 0;
                const int prefix = zstd_frame_prefix(neblock);
                cb = 0;
                if (neblock <= ZSTD_ENC_MAX_INPUT && prefix + 16 < neblock) {
                    const int nseq = lz4_encode_body<true>(lds, lds + round16(neblock), neblock, out + prefix, neblock - prefix, 1, need, dbg, item, &sink);
                    if (nseq > 0) cb = zstd_finish_frame(lds, round16(neblock), neblock, CIMG_AS_GLOBAL(out), neblock, sink, tabs);
                }
                need = cb;                                    // a frame fits a budget iff the budget holds its bytes
            }
            else if constexpr (CODEC == CODEC_LZ4_RT) cb = lz4_encode_rt_body(lds, neblock, out, neblock, accel_or_level, need, dbg, item);
            else cb = lz4_encode_wave(lds, 0, round16(neblock), neblock, out, neblock, accel_or_level, &need, dbg, item);
            if (cb > 0 && cb < neblock) {
                r.kind = REC_LZ4; r.csize = cb; r.need = need;
            } else {
                if (cb < 0) r.value = cb;                 // a loop guard tripped: stored raw, flagged for diagnosis
                wave_copy_l2g(lds, 0, out, neblock);
            }
        }
        // the record: its address is worked out again from the arguments (nothing of it was kept alive across the codec loop)
        // (written THROUGH to memory, agent scope: the wave that lays the chunk out may sit on another XCD, whose L2 is not this one)
        StreamRec* dst = fresh(ap)->recs + rec_index;
        FOR_LANES_W(l) {
            if (l == 0) {
                uint32_t* w = reinterpret_cast<uint32_t*>(dst);
                atomic_store_agent(w + 0, (uint32_t)r.kind); atomic_store_agent(w + 1, (uint32_t)r.value);
                atomic_store_agent(w + 2, (uint32_t)r.csize); atomic_store_agent(w + 3, (uint32_t)r.need);
            }
        }
    }

    // returns the number of streams this item finished (0: not an item of this launch) and their chunk
    CIMG_DEV int run_item(int item, int& chunk_out)
    {
        // ---- stage the stream: everything read from the arguments here is dead before the codec loop starts ----------
        int neblock, accel_or_level, rec_index, planes = 1, ts = 1;
        uint8_t* out;
        uint64_t* dbg;
        // block items: the byte planes that wait for their turn (dead otherwise).  The register-table encoder keeps its table where
        // these would live: its launches hand out planes only (engine.hip sets block_items = 0)
        constexpr bool BLOCK_ITEMS = CODEC != CODEC_LZ4_RT;
        LV<uint32_t> keep[BLOCK_ITEMS ? 96 : 1];
        {
            const auto a = fresh(ap);
            int b, s;
            encode_item_place(a, item, b, s);
            if (b >= a->total_blocks) return 0;
            const int chunk = find_chunk(a->descs, a->nchunks, b, a->uniform_nblocks);
            chunk_out = chunk;
            const ChunkDesc d = uniform_desc(a->descs + chunk);
            if (d.memcpyed) return 0;
            const int j = b - d.blk0;
            ts = a->p.typesize;
            const bool leftover_blk = (j == d.nblocks - 1 && d.leftover);
            const int bsize = leftover_blk ? d.leftover : d.blocksize;
            const int ns = (d.split && !leftover_blk) ? ts : 1;
            if ((ns > 1) != (a->want_split != 0) || s >= ns) return 0;
            neblock = bsize / ns;
            const uint8_t* src = a->raw + d.raw_off + (int64_t)j * d.blocksize;
            const int filter = a->p.filter;
            const bool shuf = filter == FILTER_SHUFFLE && ts > 1;
            const bool whole = BLOCK_ITEMS && ns > 1 && item < a->block_items;          // host guarantees: typesize 2 or 4, 32 KiB, byte shuffle
            if (whole) { planes = ns; s = ns - 1; }
            out = a->scratch + (int64_t)b * a->p.slot_bytes + (int64_t)s * neblock;
            rec_index = b * a->p.streams_per_block + s;
            accel_or_level = CODEC == CODEC_BLOSCLZ ? a->p.clevel : a->p.accel;
            dbg = a->dbg;
#if defined(CIMG_PROFILE) && !defined(CIMG_EMULATE)
            const unsigned long long prof_load0_ = cimg_cycles();
#elif !defined(CIMG_EMULATE)
            debug_stamp(dbg, item, 0);                                   // diagnostics (tools/diag_enctimeline.py): item taken
            if (dbg && __lane_id() == 0) dbg[16 * (size_t)item + 4] = (uint64_t)w;
#endif
            if constexpr (BLOCK_ITEMS) { if (whole) { if (ts == 2) load_block_ts2(src, keep); else load_block_ts4(src, keep); } }
            if (whole) {}
            else if (ns > 1) load_plane(src, bsize, ts, s, neblock, shuf);
            else if (filter == FILTER_BITSHUFFLE) load_block_bitshuffle(src, bsize, ts);
            else load_block(src, bsize, ts, shuf);
#if defined(CIMG_PROFILE) && !defined(CIMG_EMULATE)
            if (dbg && __lane_id() == 0) dbg[16 * (size_t)item + 14] = cimg_cycles() - prof_load0_;
#endif
        }
        // one stream -- or, for a block item, its planes most significant first, each brought from registers into the same
        // LDS (ONE call site: the codec body is large and must exist once in the kernel)
        for (int s = planes - 1; s >= 0; --s) {
            if (s != planes - 1) {
                if constexpr (BLOCK_ITEMS) restore_plane(keep, ts, s);
                out -= neblock;
                rec_index -= 1;
            }
            encode_resident(rec_index, neblock, out, accel_or_level, dbg, item);
        }
#if !defined(CIMG_PROFILE) && !defined(CIMG_EMULATE)
        CIMG_ITEM_STAMP(dbg, item, 3);                                       // item done
#endif
        return planes;
    }
};

}  // namespace cimg
