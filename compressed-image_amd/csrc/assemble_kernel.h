// assemble_kernel.h -- turn per-stream records + scratch payloads into finished blosc2 chunks.
//
//   layout_chunks  one wave per chunk.  Wave-wide prefix sum over the chunk's stream records gives
//                  every block its bstart and the chunk its cbytes, and re-applies blosc2's
//                  running-destsize rule (the reference always passes the
//                  NOMINAL chunk size + 32 as destsize, schunk.h:73): a stream that would not fit
//                  turns the whole chunk into a memcpyed chunk; a chunk whose streams are all zero
//                  runs becomes the 32-byte special-zero chunk.  Writes header + bstarts.
//   emit_blocks    one 256-thread workgroup per block: writes the csize words and copies the
//                  payloads from the scratch slot to their final place (or the raw pixels for a
//                  memcpyed chunk).
// The CPU twin of this split is oracle/chunk.c: orc_blosc2_compress_2phase.
#pragma once
#include "codec_types.h"
#include "wave.h"
#include "decode_kernel.h"

namespace cimg {

struct AssembleArgs {
    const ChunkDesc* descs;
    int32_t nchunks;
    CodecParams p;
    const uint8_t* raw;
    const uint8_t* scratch;
    const StreamRec* recs;
    uint8_t* comp;             // chunks are written at comp + desc.comp_off
    ChunkLayout* layout;       // per chunk
    int32_t uniform_nblocks;   // > 0: every chunk has this many blocks
    ChunkLayout* layout_host;  // optional second copy of `layout` in pinned host memory (read by the host after the sync)
    int32_t skip_assembled;    // 1 (the stand-alone kernels behind an assembling encode launch): chunks with ChunkDesc::assemble are done already
};

CIMG_DEV int rec_payload(const StreamRec& r) { return r.kind == REC_RUN ? (r.value > 0 ? 1 : 0) : r.csize; }

struct LayoutChunk {
    const AssembleArgs& a;
    int chunk;
    // inside the encode launch (encode_kernel.h): what this wave writes -- bstarts, header, the layout word -- is read by waves on
    // other XCDs before the launch ends, so it is written THROUGH to memory (agent-scope stores) instead of being written back by
    // an L2 flush afterwards; possible when the chunk starts on a 4-byte boundary (the caller checks)
    bool through = false;
    CIMG_DEV LayoutChunk(const AssembleArgs& a_, int chunk_) : a(a_), chunk(chunk_) {}

    CIMG_DEV void write_header(const ChunkDesc& d, uint8_t* c, int flags, int cbytes, int blosc2_flags)
    {
        LV<uint32_t> byte;
        FOR_LANES(l) {
            uint32_t v = 0;
            if (l == 0) v = 5;                                        // BLOSC2_VERSION_FORMAT_STABLE
            else if (l == 1) v = 1;                                   // codec format version
            else if (l == OFF_FLAGS) v = (uint32_t)flags;
            else if (l == OFF_TYPESIZE) v = (uint32_t)(a.p.typesize > 255 ? 1 : a.p.typesize);
            else if (l >= OFF_NBYTES && l < OFF_NBYTES + 4) v = ((uint32_t)d.nbytes >> (8 * (l - OFF_NBYTES))) & 0xFF;
            else if (l >= OFF_BLOCKSIZE && l < OFF_BLOCKSIZE + 4) v = ((uint32_t)d.blocksize >> (8 * (l - OFF_BLOCKSIZE))) & 0xFF;
            else if (l >= OFF_CBYTES && l < OFF_CBYTES + 4) v = ((uint32_t)cbytes >> (8 * (l - OFF_CBYTES))) & 0xFF;
            else if (l == OFF_FILTERS + 5) v = (uint32_t)a.p.filter;
            else if (l == OFF_COMPCODE) v = (uint32_t)a.p.compcode;
            else if (l == OFF_BLOSC2_FLAGS) v = (uint32_t)blosc2_flags;
            byte[l] = v;
        }
        if (through) {
            // eight dwords, lanes 0 .. 7 (the bytes of dword k sit in lanes 4 k .. 4 k + 3)
            LV<int> src;
            LV<uint32_t> b0, b1, b2, b3;
            FOR_LANES(l) { src[l] = (4 * l) & 63; }
            lane_gather(byte, src, b0);
            FOR_LANES(l) { src[l] = (4 * l + 1) & 63; }
            lane_gather(byte, src, b1);
            FOR_LANES(l) { src[l] = (4 * l + 2) & 63; }
            lane_gather(byte, src, b2);
            FOR_LANES(l) { src[l] = (4 * l + 3) & 63; }
            lane_gather(byte, src, b3);
            FOR_LANES_W(l) { if (l < HEADER_LEN / 4) atomic_store_agent(reinterpret_cast<uint32_t*>(c) + l, b0[l] | (b1[l] << 8) | (b2[l] << 16) | (b3[l] << 24)); }
            return;
        }
        FOR_LANES(l) { if (l < HEADER_LEN) c[l] = (uint8_t)byte[l]; }
    }

    CIMG_DEV void run()
    {
        const ChunkDesc d = uniform_desc(a.descs + chunk);
        if (a.skip_assembled && d.assemble) return;                       // the encode launch did this chunk itself
        uint8_t* c = a.comp + d.comp_off;
        ChunkLayout lay;
        lay.cbytes = 0; lay.mode = 3;
        const int memcpy_bytes = d.nbytes + HEADER_LEN;
        bool fits = !d.memcpyed;
        int nt = HEADER_LEN + 4 * d.nblocks;
        if (fits && nt > d.destsize) fits = false;
        if (fits) {
            const int ts = a.p.typesize;
            const int nfull = d.leftover ? d.nblocks - 1 : d.nblocks;
            const int per_full = d.split ? ts : 1;
            const int total = nfull * per_full + (d.leftover ? 1 : 0);
            // records of GROUP tiles (64 streams each) are requested before the first tile is looked at: one wave per
            // chunk has nothing else to hide the load latency with
            constexpr int GROUP = 4;
            for (int g0 = 0; g0 < total && fits; g0 += 64 * GROUP) {
                LV<StreamRec> recs[GROUP];
                CIMG_UNROLL
                for (int k = 0; k < GROUP; k++) {
                    FOR_LANES(l) {
                        const int i = imin(g0 + 64 * k + l, total - 1);
                        int jb, s;
                        if (i < nfull * per_full) { jb = i / per_full; s = i - jb * per_full; } else { jb = nfull; s = 0; }
                        const StreamRec* rp = a.recs + (int64_t)(d.blk0 + jb) * a.p.streams_per_block + s;
                        if (through) {
                            // other waves' records, written through by them: read past this XCD's L2 (no invalidate needed)
                            const uint32_t* w = reinterpret_cast<const uint32_t*>(rp);
                            recs[k][l].kind = (int32_t)atomic_load_agent(w + 0); recs[k][l].value = (int32_t)atomic_load_agent(w + 1);
                            recs[k][l].csize = (int32_t)atomic_load_agent(w + 2); recs[k][l].need = (int32_t)atomic_load_agent(w + 3);
                        } else {
                            recs[k][l] = *rp;
                        }
                    }
                }
                CIMG_UNROLL
                for (int k = 0; k < GROUP; k++) {
                    const int i0 = g0 + 64 * k;
                    if (i0 >= total || !fits) break;
                    LV<int> sz, pre;
                    LV<bool> bad, first;
                    LV<int> blk;
                    FOR_LANES(l) {
                        const int i = i0 + l;
                        sz[l] = 0; bad[l] = false; first[l] = false; blk[l] = 0;
                        if (i < total) {
                            int jb, s;
                            if (i < nfull * per_full) { jb = i / per_full; s = i - jb * per_full; } else { jb = nfull; s = 0; }
                            sz[l] = 4 + rec_payload(recs[k][l]);
                            first[l] = s == 0;
                            blk[l] = jb;
                        }
                    }
                    int tile_total;
                    wave_exscan(sz, pre, tile_total);
                    FOR_LANES(l) {
                        const int i = i0 + l;
                        if (i < total) {
                            const int jb = blk[l];
                            const StreamRec r = recs[k][l];
                            const int bsize = (jb == d.nblocks - 1 && d.leftover) ? d.leftover : d.blocksize;
                            const int neblock = bsize / ((d.split && !(jb == d.nblocks - 1 && d.leftover)) ? ts : 1);
                            const int N = nt + pre[l] + 4;               // offset right after this stream's csize word
                            bool f = false;
                            if (r.kind == REC_RUN) {
                                f = N > d.destsize || (r.value > 0 && N + 1 > d.destsize);
                            } else {
                                int maxout = neblock;
                                if (N + maxout > d.destsize) { maxout = d.destsize - N; if (maxout <= 0) f = true; }
                                if (!f) {
                                    if (r.kind == REC_LZ4) { if (r.need > maxout) f = true; }
                                    else if (N + neblock > d.destsize) f = true;
                                }
                            }
                            bad[l] = f;
                            if (first[l]) { if (through) atomic_store_agent(reinterpret_cast<uint32_t*>(c + HEADER_LEN) + blk[l], (uint32_t)(nt + pre[l])); else st32(c + HEADER_LEN + 4 * blk[l], nt + pre[l]); }
                        }
                    }
                    if (ballot(bad)) fits = false;
                    nt += tile_total;
                }
            }
            if (fits) {
                if (nt == HEADER_LEN + 4 * d.nblocks + 4 * total) {
                    lay.cbytes = HEADER_LEN; lay.mode = 2;
                    write_header(d, c, d.flags, HEADER_LEN, SPECIAL_ZERO << 4);
                } else {
                    lay.cbytes = nt; lay.mode = 0;
                    write_header(d, c, d.flags, nt, 0);
                }
            }
        }
        if (!fits) {
            if (memcpy_bytes <= d.destsize) {
                lay.cbytes = memcpy_bytes; lay.mode = 1;
                write_header(d, c, d.flags | FLAG_MEMCPYED, memcpy_bytes, 0);
            } else {
                lay.cbytes = 0; lay.mode = 3;
                if (d.destsize >= HEADER_LEN) write_header(d, c, d.flags, 0, 0);
            }
        }
        FOR_LANES_W(l) {
            if (l == 0) {
                if (through) {
                    atomic_store_agent(reinterpret_cast<uint32_t*>(&a.layout[chunk].cbytes), (uint32_t)lay.cbytes);
                    atomic_store_agent(reinterpret_cast<uint32_t*>(&a.layout[chunk].mode), (uint32_t)lay.mode);
                } else {
                    a.layout[chunk] = lay;
                }
                if (a.layout_host) a.layout_host[chunk] = lay;
            }
        }
    }
};

struct EmitBlock {
    const AssembleArgs& a;
    int b;
    CIMG_DEV EmitBlock(const AssembleArgs& a_, int b_) : a(a_), b(b_) {}

    // the stand-alone kernel: `wave` of `nwaves` cooperating waves copy the whole block
    CIMG_DEV void run(int wave, int nwaves = 4) { copy(wave, nwaves, 0, MAX_STREAMS); }
    // inside the encode launch: ONE wave copies the streams [s_begin, s_end) it encoded itself.  What OTHER waves wrote -- the
    // closer's layout word and bstarts[j], the records of the block's other streams -- is read with agent-scope loads straight from
    // memory (the writers wrote it through / wrote it back); the payload is this wave's own and comes out of its own L2.  No
    // acquire fence: an L2 invalidate per wave and chunk, a thousand waves at once, cost the launch more than the copies.
    CIMG_DEV void run_streams(int s_begin, int s_end) { copy(0, 1, s_begin, s_end, true); }

    CIMG_DEV void copy(int wave, int nwaves, int s_begin, int s_end, bool coherent = false)
    {
        const int chunk = find_chunk(a.descs, a.nchunks, b, a.uniform_nblocks);
        const ChunkDesc d = uniform_desc(a.descs + chunk);
        if (a.skip_assembled && d.assemble) return;                   // the encode launch did this chunk itself
        const int j = b - d.blk0;
        uint8_t* c = a.comp + d.comp_off;
        const bool leftover_blk = (j == d.nblocks - 1 && d.leftover);
        const int bsize = leftover_blk ? d.leftover : d.blocksize;
        const int ns = (d.split && !leftover_blk) ? a.p.typesize : 1;
        const StreamRec* rp = a.recs + (int64_t)b * a.p.streams_per_block;
        // What other waves wrote: the layout word, bstarts[j], {kind, value, csize} of the block's streams.  Coherent form: ONE
        // round trip of agent-scope loads, lane 0 the mode, lane 1 bstarts[j], lanes 2 + 3 s .. 4 + 3 s the record of stream s
        // (MAX_STREAMS = 16 streams: 50 lanes).  (bstarts[] is 4-byte aligned whenever the chunk is; a caller-chosen odd chunk
        // address takes a fence instead.)
        const bool aligned = ((uintptr_t)(c + HEADER_LEN) & 3) == 0;
        LV<int32_t> meta;
        if (coherent) {
            if (!aligned) fence_acquire();
            FOR_LANES(l) {
                const int k = l - 2, sidx = k >= 0 ? k / 3 : 0, f = k >= 0 ? k - 3 * sidx : 0;
                const uint32_t* p = l == 0 ? reinterpret_cast<const uint32_t*>(&a.layout[chunk].mode)
                                  : l == 1 ? reinterpret_cast<const uint32_t*>(c + HEADER_LEN + (aligned ? 4 * j : 0))
                                           : reinterpret_cast<const uint32_t*>(rp + (sidx < ns ? sidx : 0)) + f;
                meta[l] = (l == 1 && !aligned) ? ld32s(c + HEADER_LEN + 4 * j) : (int32_t)atomic_load_agent(p);
            }
        } else {
            FOR_LANES(l) {
                const int k = l - 2, sidx = k >= 0 ? k / 3 : 0, f = k >= 0 ? k - 3 * sidx : 0;
                const StreamRec& r = rp[sidx < ns ? sidx : 0];
                meta[l] = l == 0 ? a.layout[chunk].mode : l == 1 ? ld32s(c + HEADER_LEN + 4 * j) : (f == 0 ? r.kind : f == 1 ? r.value : r.csize);
            }
        }
        const int mode = readlane(meta, 0);
        if (mode == 1) {
            // memcpyed chunk: the raw pixels of the block; a caller that owns streams [s_begin, s_end) of ns copies that share of them
            const int lo = s_begin <= 0 ? 0 : (int)((int64_t)bsize * s_begin / ns) & ~15;
            const int hi = s_end >= ns ? bsize : (int)((int64_t)bsize * s_end / ns) & ~15;
            const int64_t at = (int64_t)j * d.blocksize + lo;
            if (hi > lo) { if (nwaves == 1) wave_copy_g2g<16>(a.raw + d.raw_off + at, c + HEADER_LEN + at, hi - lo, 0, 1); else wave_copy_g2g(a.raw + d.raw_off + at, c + HEADER_LEN + at, hi - lo, wave, nwaves); }
            return;
        }
        if (mode != 0) return;
        const int neblock = bsize / ns;
        const uint8_t* slot = a.scratch + (int64_t)b * a.p.slot_bytes;
        int pos = readlane(meta, 1);
        for (int s = 0; s < ns && s < s_end && s < MAX_STREAMS; s++) {
            StreamRec r;
            r.kind = readlane(meta, 2 + 3 * s); r.value = readlane(meta, 3 + 3 * s); r.csize = readlane(meta, 4 + 3 * s); r.need = 0;
            const bool mine = s >= s_begin;
            if (wave == 0 && mine) {
                const int word = r.kind == REC_RUN ? -r.value : r.csize;
                FOR_LANES(l) {
                    if (l < 4) c[pos + l] = (uint8_t)(((uint32_t)word >> (8 * l)) & 0xFF);
                    if (l == 4 && r.kind == REC_RUN && r.value > 0) c[pos + 4] = 0x01;
                }
            }
            pos += 4;
            if (r.kind != REC_RUN && mine) { if (nwaves == 1) wave_copy_g2g<16>(slot + (int64_t)s * neblock, c + pos, r.csize, 0, 1); else wave_copy_g2g(slot + (int64_t)s * neblock, c + pos, r.csize, wave, nwaves); }
            pos += rec_payload(r);
        }
    }
};

}  // namespace cimg
