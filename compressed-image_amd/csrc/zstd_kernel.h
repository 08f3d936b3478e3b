// zstd_kernel.h -- blocks of zstd-coded chunks (blosc2 codec format 4), one wave per block.  The slow path of the decoder:
// chunks the reference wrote with enums::codec::zstd (enums.h:18-24) stay readable.
//
// The engine launches this kernel only behind a batch in which cimg_decode_blocks marked some chunk STATUS_ZSTD_PENDING (codec
// format 4 in its header; engine.hip: decompress_finish clears exactly those words, launches, and reads the status again).  A block of any other chunk is left alone, whatever
// its status says.  LDS: the block's streams decoded back to back (a frame's literals are regenerated inside its own output:
// zstd_decode.h, zstd_block), 8 KiB through which the frame -- or, of a larger one, the section being decoded -- is read, the
// entropy tables (ZstdWork): 52.5 KiB for 32 KiB blocks, three blocks per CU.  The decoder is issue bound (every lane executes
// the scalar decoder with the same data: wave-uniform control flow, same-value LDS writes), so its rate is the number of waves
// a CU holds; the filter stage at the end (the general kernel's) and the byte movers are the lane-parallel parts.
#pragma once
#include "decode_kernel.h"
#include "zstd_decode.h"

namespace cimg {

// The launch is sized for the largest block of the batch (rounded up to 64 bytes, at least 32 KiB): `area` bytes for the planes,
// 16 bytes nobody uses, the stage, the tables.  The kernel reads `area` back from the launch's LDS size.
enum : int { ZSTD_KERNEL_AREA_MIN = 32768, ZSTD_KERNEL_STAGE = 8192 };
CIMG_HD int zstd_work_bytes() { return (int)((sizeof(ZstdWork) + 15) & ~(size_t)15); }
CIMG_HD int zstd_kernel_area(int max_blocksize) { const int a = (max_blocksize + 63) & ~63; return a < ZSTD_KERNEL_AREA_MIN ? ZSTD_KERNEL_AREA_MIN : a; }
CIMG_HD int zstd_kernel_lds_bytes(int max_blocksize) { return zstd_kernel_area(max_blocksize) + 16 + ZSTD_KERNEL_STAGE + zstd_work_bytes() + 64; }

struct DecodeZstdBlock {
    const DecodeArgs& a;
    uint8_t* lds;
    int b;
    CIMG_DEV DecodeZstdBlock(const DecodeArgs& a_, uint8_t* lds_, int b_) : a(a_), lds(lds_), b(b_) {}

    CIMG_DEV void run()
    {
        const int chunk = find_chunk(a.descs, a.nchunks, b, a.uniform_nblocks);
        const ChunkDesc d = uniform_desc(a.descs + chunk);
        const int j = b - d.blk0;
        const uint8_t* c = a.comp + d.comp_off;
        uint8_t* out = a.raw + d.raw_off + (int64_t)j * d.blocksize;
        const int bsize = (j == d.nblocks - 1 && d.leftover) ? d.leftover : d.blocksize;
        const u128 h0 = ld128u(c), h1 = ld128u(c + 16);
        const uint32_t w0 = uni(h0.x);
        const int flags = (int)((w0 >> 16) & 0xFF), ts = (int)(w0 >> 24);
        const int nbytes = (int)uni(h0.y), blocksize = (int)uni(h0.z), cbytes = (int)uni(h0.w);
        const uint32_t f0 = uni(h1.x), f1 = uni(h1.y), b2 = uni(h1.w);
        // whatever cimg_decode_blocks already settled -- damaged headers, special and memcpyed chunks, its own codecs -- is not ours
        if ((w0 & 0xFF) > 5 || nbytes != d.nbytes || blocksize != d.blocksize || ts == 0 || cbytes < HEADER_LEN || cbytes > d.destsize) return;
        if ((flags & (FLAG_SHUFFLE | FLAG_BITSHUFFLE)) != (FLAG_SHUFFLE | FLAG_BITSHUFFLE)) return;
        if (((b2 >> 28) & 7) != 0 || (flags & FLAG_MEMCPYED)) return;
        const int fmt = flags >> 5;
        if (fmt != 4) return;                                           // its own codecs, and formats nobody reads (reported already)
        const int filter = (int)((f1 >> 8) & 0xFF);
        if (f0 != 0 || (f1 & 0xFF) != 0 || (filter != FILTER_NONE && filter != FILTER_SHUFFLE && filter != FILTER_BITSHUFFLE)) { fail(chunk, ERR_CODEC_SUPPORT); return; }
        if (filter == FILTER_BITSHUFFLE && !(flags & FLAG_DONT_SPLIT)) { fail(chunk, ERR_CODEC_SUPPORT); return; }          // bit rows are never split
        const int area = (a.lds_bytes - zstd_work_bytes() - 64 - ZSTD_KERNEL_STAGE - 16) & ~63;   // zstd_kernel_lds_bytes, read backwards
        if (area < ZSTD_KERNEL_AREA_MIN || blocksize > area) { fail(chunk, ERR_CODEC_SUPPORT); return; }
        const bool leftover_blk = bsize != blocksize;
        const int ns = (!(flags & FLAG_DONT_SPLIT) && !leftover_blk) ? ts : 1;
        const int neblock = bsize / ns;
        if (cbytes < HEADER_LEN + 4 * d.nblocks) { fail(chunk, ERR_READ_BUFFER); return; }
        const int bstart = ld32s(c + HEADER_LEN + 4 * j);
        if (bstart < HEADER_LEN + 4 * d.nblocks || bstart > cbytes) { fail(chunk, ERR_DATA); return; }
        ZstdWork* w = reinterpret_cast<ZstdWork*>(lds + area + 16 + ZSTD_KERNEL_STAGE);
        // a frame of at most ZSTD_KERNEL_STAGE bytes is copied into LDS whole -- the decoder reads it bit by bit, and an LDS read is
        // a fifth of a global one; of a larger one, each block's sections go through the same bytes when they fit (ZstdWork::tail)
        uint8_t* const stage = lds + area + 16;
        w->stage = stage;
        w->stage_cap = ZSTD_KERNEL_STAGE;
        w->mem_lo = lds;                                       // (the executor's 16-byte fetches may look anywhere in this workgroup's LDS)
        w->mem_hi = lds + (a.lds_bytes & ~3);
        int pos = bstart;
        for (int s = 0; s < ns; s++) {
            if (cbytes - pos < 4) { fail(chunk, ERR_READ_BUFFER); return; }
            const int cs = uni(ld32s(c + pos));
            pos += 4;
            const int payload = cs > 0 ? cs : (cs < 0 ? 1 : 0);
            if (payload > cbytes - pos) { fail(chunk, ERR_READ_BUFFER); return; }
            uint8_t* plane = lds + s * neblock;
            if (cs <= 0) {
                if (cs < 0 && (!(c[pos] & 1) || cs < -255)) { fail(chunk, ERR_RUN_LENGTH); return; }
                const uint8_t v = (uint8_t)((uint32_t)(-cs) & 0xFF);
                for (int i = 0; i < neblock; i += 64) { FOR_LANES_W(l) { if (i + l < neblock) plane[i + l] = v; } }
            } else if (cs == neblock) {
                // stored stream: 16 bytes per lane, eight loads in flight (a byte loop waits for HBM 256 times per plane)
                if (((s * neblock) & 15) == 0) wave_copy_g2l(c + pos, lds, s * neblock, neblock);
                else for (int i = 0; i < neblock; i += 64) { FOR_LANES_W(l) { if (i + l < neblock) plane[i + l] = c[pos + i + l]; } }
            } else if (cs > neblock) {
                fail(chunk, ERR_DATA); return;
            } else {
                const bool staged = cs <= ZSTD_KERNEL_STAGE;
                if (staged) wave_copy_g2l(c + pos, lds, area + 16, cs);
                w->tail = staged ? 0 : 1;
                const int r = zstd_decode_frame(staged ? stage : c + pos, cs, plane, neblock, w);
#if !defined(CIMG_ABL_ZSTD_NO_SEQ) && !defined(CIMG_ABL_ZSTD_NO_EXEC)
                if (r != neblock) { fail(chunk, r < 0 ? r : ERR_DATA); return; }
#endif
            }
            pos += payload;
        }
        // the filter stage is the general kernel's own (decode_kernel.h: DecodeBlock::phase_b -- byte shuffle for every element
        // size, bit shuffle, none): the planes lie back to back here, i.e. its region stride is the plane size, and its four
        // waves' shares are walked one after the other by this one
        DecodeBlock fb(a, lds, b);
        fb.chunk = chunk; fb.j = j; fb.bsize = bsize; fb.ns = ns; fb.neblock = neblock; fb.rs = neblock; fb.ts = ts;
        fb.filter = filter; fb.mode = 0; fb.c = c; fb.out = out;
        for (int wv = 0; wv < 4; ++wv) fb.phase_b(wv);
    }

    CIMG_DEV void fail(int chunk, int code) { FOR_LANES_W(l) { if (l == 0) a.status[chunk] = code; } }
};

}  // namespace cimg
