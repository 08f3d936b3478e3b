// zstd_kernel.h -- blocks of zstd-coded chunks (blosc2 codec format 4).  The slow path of the decoder:
// chunks the reference wrote with enums::codec::zstd (enums.h:18-24) stay readable.
//
// The engine launches this kernel only behind a batch in which cimg_decode_blocks marked some chunk STATUS_ZSTD_PENDING (codec
// format 4 in its header; engine.hip: decompress_finish clears exactly those words, launches, and reads the status again).  A block of any other chunk is left alone, whatever
// its status says.  The decoder is issue bound (every lane executes the scalar entropy decoder with the same data: wave-uniform
// control flow, same-value LDS writes), so its rate is the number of waves a CU holds -- and what a wave needs is LDS: the
// block's planes (a frame's literals are regenerated inside its own output: zstd_decode.h, zstd_block), 8 KiB through which the
// frame -- or, of a larger one, the section being decoded -- is read, the entropy tables (ZstdWork).  Two launch shapes:
//   one wave per block    52.5 KiB for 32 KiB blocks, three blocks = three waves per CU: blocks that are ONE stream (element size 1,
//                         bit-shuffled or unsplit chunks)
//   two waves per block   the streams of a split block are frames of their own: the two waves take them from a counter in LDS, each
//                         with its own stage and tables -- 72 KiB a block, two blocks = FOUR waves per CU (round 4: 5.9 -> 4.5 ms
//                         on 128 MiB of libzstd's float32 chunks)
// The filter stage at the end (the general kernel's) and the byte movers are the lane-parallel parts.
#pragma once
#include "decode_kernel.h"
#include "zstd_decode.h"

namespace cimg {

// The launch is sized for the largest block of the batch (rounded up to 64 bytes, at least 32 KiB): `area` bytes for the planes,
// 32 bytes of control words (the stream counter, what each wave has to report), then per wave the stage and the tables.  The
// kernel reads `area` back from the launch's LDS size and its wave count.
enum : int { ZSTD_KERNEL_AREA_MIN = 32768, ZSTD_KERNEL_STAGE = 8192, ZSTD_KERNEL_CTL = 32, ZSTD_KERNEL_MAX_WAVES = 2 };
enum : int { ZSTD_BLOCK_NOT_OURS = -2, ZSTD_BLOCK_FINE = 0x7FFFFFFF };       // a wave's report: the first stream that failed (-1: the block's header), or one of these
CIMG_HD int zstd_work_bytes() { return (int)((sizeof(ZstdWork) + 15) & ~(size_t)15); }
CIMG_HD int zstd_kernel_area(int max_blocksize) { const int a = (max_blocksize + 63) & ~63; return a < ZSTD_KERNEL_AREA_MIN ? ZSTD_KERNEL_AREA_MIN : a; }
CIMG_HD int zstd_kernel_lds_bytes(int max_blocksize, int waves) { return zstd_kernel_area(max_blocksize) + ZSTD_KERNEL_CTL + waves * (ZSTD_KERNEL_STAGE + zstd_work_bytes()) + 64; }

#ifdef CIMG_EMULATE
extern int g_emu_zstd_take;
#endif

// What the three kernels of the zstd read path first learn about block b: whose it is, its geometry, where its streams begin.
struct ZstdBlockGeom {
    int chunk, j, bsize, ns, neblock, ts, filter, cbytes, bstart;
    bool split_chunk;
    const uint8_t* c;
    uint8_t* out;
    // 1: a block of a zstd chunk; 0: not ours (whatever cimg_decode_blocks already settled -- damaged headers, special and memcpyed
    // chunks, its own codecs -- or, kind_of_launch = 1 / 2, a chunk of the other launch's kind); < 0: the block's error code
    CIMG_DEV int parse(const DecodeArgs& a, int b, int area, int kind_of_launch)
    {
        chunk = find_chunk(a.descs, a.nchunks, b, a.uniform_nblocks);
        const ChunkDesc d = uniform_desc(a.descs + chunk);
        j = b - d.blk0;
        c = a.comp + d.comp_off;
        out = a.raw + d.raw_off + (int64_t)j * d.blocksize;
        bsize = (j == d.nblocks - 1 && d.leftover) ? d.leftover : d.blocksize;
        const u128 h0 = ld128u(c), h1 = ld128u(c + 16);
        const uint32_t w0 = uni(h0.x);
        const int flags = (int)((w0 >> 16) & 0xFF);
        ts = (int)(w0 >> 24);
        const int nbytes = (int)uni(h0.y), blocksize = (int)uni(h0.z);
        cbytes = (int)uni(h0.w);
        const uint32_t f0 = uni(h1.x), f1 = uni(h1.y), b2 = uni(h1.w);
        bool ours = !((w0 & 0xFF) > 5 || nbytes != d.nbytes || blocksize != d.blocksize || ts == 0 || cbytes < HEADER_LEN || cbytes > d.destsize);
        ours = ours && (flags & (FLAG_SHUFFLE | FLAG_BITSHUFFLE)) == (FLAG_SHUFFLE | FLAG_BITSHUFFLE);
        ours = ours && ((b2 >> 28) & 7) == 0 && !(flags & FLAG_MEMCPYED);
        ours = ours && (flags >> 5) == 4;                                     // its own codecs, and formats nobody reads (reported already)
        split_chunk = !(flags & FLAG_DONT_SPLIT) && ts > 1;
        if (ours && kind_of_launch && split_chunk != (kind_of_launch == 2)) ours = false;
        if (!ours) return 0;
        filter = (int)((f1 >> 8) & 0xFF);
        if (f0 != 0 || (f1 & 0xFF) != 0 || (filter != FILTER_NONE && filter != FILTER_SHUFFLE && filter != FILTER_BITSHUFFLE)) return ERR_CODEC_SUPPORT;
        if (filter == FILTER_BITSHUFFLE && !(flags & FLAG_DONT_SPLIT)) return ERR_CODEC_SUPPORT;          // bit rows are never split
        if (area < ZSTD_KERNEL_AREA_MIN || blocksize > area) return ERR_CODEC_SUPPORT;
        const bool leftover_blk = bsize != blocksize;
        ns = (!(flags & FLAG_DONT_SPLIT) && !leftover_blk) ? ts : 1;
        neblock = bsize / ns;
        if (cbytes < HEADER_LEN + 4 * d.nblocks) return ERR_READ_BUFFER;
        bstart = ld32s(c + HEADER_LEN + 4 * j);
        if (bstart < HEADER_LEN + 4 * d.nblocks || bstart > cbytes) return ERR_DATA;
        return 1;
    }
    // the size word of the stream at pos (pos moves behind it); payload = the bytes the stream takes behind the word
    CIMG_DEV int stream_header(int& pos, int& cs, int& payload) const
    {
        if (cbytes - pos < 4) return ERR_READ_BUFFER;
        cs = uni(ld32s(c + pos));
        pos += 4;
        payload = cs > 0 ? cs : (cs < 0 ? 1 : 0);
        if (payload > cbytes - pos) return ERR_READ_BUFFER;
        return 0;
    }
};

struct DecodeZstdBlock {
    const DecodeArgs& a;
    uint8_t* lds;
    int b, nw;
    // what phase_a leaves for phase_b (every wave holds its own copy, as in DecodeBlock)
    int chunk, j, bsize, ns, neblock, ts, filter, area;
    const uint8_t* c;
    uint8_t* out;
    CIMG_DEV DecodeZstdBlock(const DecodeArgs& a_, uint8_t* lds_, int b_, int nw_) : a(a_), lds(lds_), b(b_), nw(nw_) {}

    CIMG_DEV uint32_t* ctl() const { return reinterpret_cast<uint32_t*>(lds + area); }
    // before phase_a (and a barrier, with two waves): the stream counter
    CIMG_DEV void init()
    {
        area = (a.lds_bytes - 64 - nw * (ZSTD_KERNEL_STAGE + zstd_work_bytes()) - ZSTD_KERNEL_CTL) & ~63;   // zstd_kernel_lds_bytes, read backwards
        FOR_LANES_W(l) { if (l == 0) ctl()[0] = 0; }
    }
    CIMG_DEV void report(int wv, int stream, int code) { FOR_LANES_W(l) { if (l == 0) { ctl()[2 + 2 * wv] = (uint32_t)stream; ctl()[3 + 2 * wv] = (uint32_t)code; } } }

    // the streams: wave wv decodes what it gets from the counter, into the block's planes
    CIMG_DEV void phase_a(int wv)
    {
        area = (a.lds_bytes - 64 - nw * (ZSTD_KERNEL_STAGE + zstd_work_bytes()) - ZSTD_KERNEL_CTL) & ~63;
        ZstdBlockGeom g;
        // (a split chunk is the two-wave launch's, every other one the one-wave launch's, when a batch has both: a.tune says so)
        int ours = g.parse(a, b, area, a.tune == 1 ? (nw > 1 ? 2 : 1) : 0);
        chunk = g.chunk; j = g.j; bsize = g.bsize; ts = g.ts; c = g.c; out = g.out;
        // (behind the walk / replay launches: only the blocks whose plan did not fit its slot)
        if (ours != 0 && a.zplan != nullptr) {
            const int32_t* const head = reinterpret_cast<const int32_t*>(a.zplan + (int64_t)(b - a.blk_first) * a.zplan_stride);
            if ((int)uni((uint32_t)head[0]) != 2) ours = 0;
        }
        if (ours == 0) { report(wv, ZSTD_BLOCK_NOT_OURS, 0); return; }
        if (ours < 0) { report(wv, -1, ours); return; }
        filter = g.filter; ns = g.ns; neblock = g.neblock;
        const int cbytes = g.cbytes, bstart = g.bstart;
        // this wave's stage and tables.  A frame of at most ZSTD_KERNEL_STAGE bytes is copied into LDS whole -- the decoder reads it
        // bit by bit, and an LDS read is a fifth of a global one; of a larger one, each block's sections go through the same bytes
        // when they fit (ZstdWork::tail)
        const int mine = area + ZSTD_KERNEL_CTL + wv * (ZSTD_KERNEL_STAGE + zstd_work_bytes());
        uint8_t* const stage = lds + mine;
        ZstdWork* w = reinterpret_cast<ZstdWork*>(stage + ZSTD_KERNEL_STAGE);
        w->stage = stage;
        w->stage_cap = ZSTD_KERNEL_STAGE;
        w->ops = nullptr;                                      // (the decoder proper: zstd_decode.h has a walker's form as well)
        w->mem_lo = lds;                                       // (the executor's 16-byte fetches may look anywhere in this workgroup's LDS)
        w->mem_hi = lds + (a.lds_bytes & ~3);
        // The stream headers are a chain (a stream begins where the one before it ends): a wave walks it from the last stream it
        // took to the next one -- every wave meets every damaged header below its own streams, and reports it under that stream.
        int at = 0, pos = bstart;
#ifdef CIMG_EMULATE
        int taken = 0;
#endif
        for (;;) {
#ifdef CIMG_EMULATE     /* the host runs the waves of a block one after the other: a wave may be told to come back after n streams */
            if (taken++ >= g_emu_zstd_take) break;
#endif
            int s;
            if (nw > 1) {
                LV<uint32_t> got;
                FOR_LANES(l) { got[l] = 0; }
                FOR_LANES_W(l) { if (l == 0) got[l] = atomic_add_workgroup(ctl(), 1u); }
                s = (int)uni(readlane(got, 0));
            } else s = at;                                      // (one wave: every stream, in order)
            if (s >= ns) break;
            int cs = 0, payload = 0;
            for (;; ++at) {
                if (cbytes - pos < 4) { report(wv, at, ERR_READ_BUFFER); return; }
                cs = uni(ld32s(c + pos));
                pos += 4;
                payload = cs > 0 ? cs : (cs < 0 ? 1 : 0);
                if (payload > cbytes - pos) { report(wv, at, ERR_READ_BUFFER); return; }
                if (at == s) break;
                pos += payload;
            }
            uint8_t* plane = lds + s * neblock;
            if (cs <= 0) {
                if (cs < 0 && (!(c[pos] & 1) || cs < -255)) { report(wv, s, ERR_RUN_LENGTH); return; }
                const uint8_t v = (uint8_t)((uint32_t)(-cs) & 0xFF);
                for (int i = 0; i < neblock; i += 64) { FOR_LANES_W(l) { if (i + l < neblock) plane[i + l] = v; } }
            } else if (cs == neblock) {
                // stored stream: 16 bytes per lane, eight loads in flight (a byte loop waits for HBM 256 times per plane)
                if (((s * neblock) & 15) == 0) wave_copy_g2l(c + pos, lds, s * neblock, neblock);
                else for (int i = 0; i < neblock; i += 64) { FOR_LANES_W(l) { if (i + l < neblock) plane[i + l] = c[pos + i + l]; } }
            } else if (cs > neblock) {
                report(wv, s, ERR_DATA); return;
            } else {
                const bool staged = cs <= ZSTD_KERNEL_STAGE;
                if (staged) wave_copy_g2l(c + pos, lds, mine, cs);
                w->tail = staged ? 0 : 1;
                const int r = zstd_decode_frame(staged ? stage : c + pos, cs, plane, neblock, w);
#if !defined(CIMG_ABL_ZSTD_NO_SEQ) && !defined(CIMG_ABL_ZSTD_NO_EXEC)
                if (r != neblock) { report(wv, s, r < 0 ? r : ERR_DATA); return; }
#endif
            }
            pos += payload;
            ++at;
        }
        report(wv, ZSTD_BLOCK_FINE, 0);
    }

    // after phase_a (and a barrier): what the waves report -- the failure of the LOWEST stream is the block's, as if the streams
    // had been decoded in order -- and the filter stage, the waves sharing its four parts
    CIMG_DEV void phase_b(int wv)
    {
        int first = ZSTD_BLOCK_FINE, code = 0;
        for (int k = 0; k < nw; ++k) {
            const int st = (int)uni(ctl()[2 + 2 * k]), cd = (int)uni(ctl()[3 + 2 * k]);
            if (st == ZSTD_BLOCK_NOT_OURS) return;
            if (st < first) { first = st; code = cd; }
        }
        // (behind the walk / replay launches: this block was left to this kernel, and the host waits to hear that it was taken --
        // whatever became of it, which the status word says)
        if (wv == 0 && a.zplan != nullptr && a.skipped) { FOR_LANES_W(l) { if (l == 0) a.skipped[1 + b] = 2 /* ZFALL_DONE */; } }
        if (first != ZSTD_BLOCK_FINE) { if (wv == 0) fail(chunk, code); return; }
        // the filter stage is the general kernel's own (decode_kernel.h: DecodeBlock::phase_b -- byte shuffle for every element
        // size, bit shuffle, none): the planes lie back to back here, i.e. its region stride is the plane size, and its four
        // waves' shares are walked one after the other
        DecodeBlock fb(a, lds, b);
        fb.chunk = chunk; fb.j = j; fb.bsize = bsize; fb.ns = ns; fb.neblock = neblock; fb.rs = neblock; fb.ts = ts;
        fb.filter = filter; fb.mode = 0; fb.c = c; fb.out = out;
        for (int q = wv; q < 4; q += nw) fb.phase_b(q);
    }

    CIMG_DEV void fail(int chunk_, int code) { FOR_LANES_W(l) { if (l == 0) a.status[chunk_] = code; } }
};

}  // namespace cimg
