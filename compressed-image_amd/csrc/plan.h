// plan.h -- host-side chunk geometry: what blosc2_compress_ctx derives from (cparams, srcsize)
// before it touches a byte (oracle/chunk.c: orc_chunk_geometry is the
// checker's copy of the same rules).  Pure C++, shared by the engine and the emulator.
#pragma once
#include "codec_types.h"
#include <stdint.h>
#include <vector>

namespace cimg {

struct HostCParams {
    int32_t typesize = 1;
    int32_t clevel = 9;
    int32_t blocksize = 32768;                 // constants.h:11 default
    int32_t compcode = CODEC_LZ4;              // channel.h:101 default
    int32_t splitmode = SPLIT_AUTO;            // wrapper.h:328,353
    uint8_t filters[6] = {0, 0, 0, 0, 0, FILTER_SHUFFLE};   // BLOSC2_CPARAMS_DEFAULTS
    uint8_t filters_meta[6] = {0, 0, 0, 0, 0, 0};
};

inline int compformat_of(int compcode)
{
    switch (compcode) {
    case CODEC_BLOSCLZ: return 0;
    case CODEC_LZ4: case CODEC_LZ4HC: return 1;
    case CODEC_ZLIB: return 3;
    case CODEC_ZSTD: return 4;
    default: return -1;
    }
}

// the single filter the kernels implement: one of none/shuffle/bitshuffle, in the last pipeline slot
inline int single_filter(const HostCParams& p, int* filter)
{
    for (int i = 0; i < 5; i++) if (p.filters[i] != 0) return ERR_CODEC_SUPPORT;
    if (p.filters[5] > FILTER_BITSHUFFLE) return ERR_CODEC_SUPPORT;
    *filter = p.filters[5];
    return 0;
}

inline bool wants_split(const HostCParams& p, int typesize, int blocksize)
{
    if (p.splitmode == SPLIT_ALWAYS) return true;
    if (p.splitmode == SPLIT_NEVER) return false;
    const bool fast = p.compcode == CODEC_BLOSCLZ || p.compcode == CODEC_LZ4 || (p.compcode == CODEC_ZSTD && p.clevel <= 5);
    bool shuffle = false;
    for (int i = 0; i < 6; i++) shuffle |= p.filters[i] == FILTER_SHUFFLE;
    return fast && shuffle && typesize <= MAX_STREAMS && blocksize / typesize >= MIN_BUFFERSIZE;
}

// fills everything in `d` except raw_off / comp_off / blk0; returns 0 or a blosc2 error code
inline int plan_chunk(const HostCParams& p, int32_t nbytes, int32_t destsize, ChunkDesc* d)
{
    *d = ChunkDesc{};
    if (nbytes < 0) return ERR_MAX_BUFSIZE;
    if (destsize < HEADER_LEN) return ERR_MAX_BUFSIZE;
    if (p.clevel < 0 || p.clevel > 9) return ERR_CODEC_PARAM;
    if (compformat_of(p.compcode) < 0) return ERR_CODEC_SUPPORT;
    int ts = p.typesize;
    if (ts <= 0) return ERR_INVALID_PARAM;
    if (ts > 255) ts = 1;
    int bs;
    if (nbytes < ts) {
        bs = 1;
    } else {
        if (p.blocksize <= 0) return ERR_INVALID_PARAM;     // automatic block size is not on the path
        bs = p.blocksize;
        if (bs < MIN_BUFFERSIZE) bs = MIN_BUFFERSIZE;       // a forced block size below 32 bytes is raised to 32 (upstream compute_blocksize, recalled)
        if (bs > nbytes) bs = nbytes;
        if (bs > ts) bs = bs / ts * ts;
    }
    d->nbytes = nbytes;
    d->destsize = destsize;
    d->blocksize = bs;
    d->nblocks = nbytes / bs;
    d->leftover = nbytes % bs;
    if (d->leftover) d->nblocks++;
    d->memcpyed = (p.clevel == 0 || nbytes < MIN_BUFFERSIZE) ? 1 : 0;
    d->flags = FLAG_SHUFFLE | FLAG_BITSHUFFLE;
    // one rule for the header's flags byte: the dont-split bit and the codec format are set whether or not the chunk is
    // memcpyed up front (oracle/chunk.c: orc_chunk_geometry says where that comes from)
    const bool split = wants_split(p, ts, bs);
    if (!split) d->flags |= FLAG_DONT_SPLIT;
    d->flags |= compformat_of(p.compcode) << 5;
    if (d->memcpyed) d->flags |= FLAG_MEMCPYED;
    else d->split = split ? 1 : 0;
    if (d->split) d->nstreams = d->leftover ? (d->nblocks - 1) * ts + 1 : d->nblocks * ts;
    else d->nstreams = d->nblocks;
    return 0;
}

}  // namespace cimg

// ---- batch plans (shared by the HIP engine and the emulator) -------------------------------------
#include "decode_kernel.h"
#include "decode_lean_kernel.h"
#include "encode_kernel.h"

namespace cimg {

inline int uniform_blocks(const std::vector<ChunkDesc>& descs)
{
    if (descs.empty() || descs[0].nblocks <= 0) return 0;
    for (const auto& d : descs) if (d.nblocks != descs[0].nblocks) return 0;
    return descs[0].nblocks;
}

struct EncodePlan {
    std::vector<ChunkDesc> descs;
    CodecParams cp{};
    int32_t total_blocks = 0;
    int32_t lds_split = 0;      // LDS per stream workgroup of the split-block launch (0: no such blocks)
    int32_t lds_unsplit = 0;    // same for the unsplit-block launch (leftover blocks, dont-split chunks)
    int32_t uniform_nblocks = 0; // > 0: every chunk has this many blocks
};

enum : int { MAX_LDS_BYTES = 160 * 1024 };

inline int plan_encode_batch(const HostCParams& p, int nchunks, const int64_t* raw_off, const int32_t* nbytes,
                             const int64_t* comp_off, const int32_t* destsize, EncodePlan* plan)
{
    plan->descs.resize((size_t)nchunks);
    int filter = 0;
    int rc = single_filter(p, &filter);
    if (rc < 0) return rc;
    // lz4 and blosclz: bit-exact encoders.  lz4hc and zstd: FORMAT-VALID encoders whose bytes differ from liblz4's / libzstd's by
    // construction: lz4hc chunks are LZ4 blocks from the fast match finder at acceleration 1, zstd chunks are
    // frames built from the same matches (zstd_encode.h).  zlib: not built.
    if (p.compcode != CODEC_LZ4 && p.compcode != CODEC_BLOSCLZ && p.compcode != CODEC_LZ4HC && p.compcode != CODEC_ZSTD) return ERR_CODEC_SUPPORT;
    CodecParams& cp = plan->cp;
    cp.typesize = p.typesize > 255 ? 1 : p.typesize;
    cp.clevel = p.clevel;
    cp.compcode = p.compcode;
    cp.filter = filter;
    cp.accel = (p.compcode == CODEC_LZ4HC || p.compcode == CODEC_ZSTD) ? 1 : 10 - p.clevel;   // (the substitutes always search at acceleration 1)
    cp.max_blocksize = 0;
    cp.streams_per_block = 1;
    int32_t blk = 0;
    plan->lds_split = plan->lds_unsplit = 0;
    for (int i = 0; i < nchunks; i++) {
        ChunkDesc& d = plan->descs[(size_t)i];
        rc = plan_chunk(p, nbytes[i], destsize[i], &d);
        if (rc < 0) return rc;
        d.raw_off = raw_off[i];
        d.comp_off = comp_off[i];
        d.blk0 = blk;
        blk += d.nblocks;
        if (d.blocksize > cp.max_blocksize) cp.max_blocksize = d.blocksize;
        if (d.split && filter == FILTER_BITSHUFFLE) return ERR_CODEC_SUPPORT;   // bit rows are one stream (forced split: not built)
        if (d.split) cp.streams_per_block = cp.typesize;
        // every stream of the chunk in ONE encode launch (full blocks split into planes AND an unsplit leftover block: two launches)
        d.assemble = (!d.memcpyed && !(d.split && cp.typesize > 1 && d.leftover)) ? 1 : 0;
        if (!d.memcpyed) {
            const int nfull = d.leftover ? d.nblocks - 1 : d.nblocks;
            const bool multi = d.split && cp.typesize > 1;        // full blocks are cut into several planes
            if (multi && nfull > 0) plan->lds_split = imax(plan->lds_split, encode_lds_bytes(d.blocksize / cp.typesize, p.compcode));
            if (!multi && nfull > 0) plan->lds_unsplit = imax(plan->lds_unsplit, encode_lds_bytes(d.blocksize, p.compcode));
            if (d.leftover) plan->lds_unsplit = imax(plan->lds_unsplit, encode_lds_bytes(d.leftover, p.compcode));
            const int stream_max = multi ? imax(nfull > 0 ? d.blocksize / cp.typesize : 0, d.leftover) : d.blocksize;
            if (stream_max > (p.compcode == CODEC_BLOSCLZ ? (int)BLZ_MAX_INPUT : p.compcode == CODEC_ZSTD ? (int)ZSTD_ENC_MAX_INPUT : (int)LZ4_MAX_INPUT_U16)) return ERR_CODEC_SUPPORT;   // 32-bit position tables: not built
        }
    }
    if (plan->lds_split > MAX_LDS_BYTES || plan->lds_unsplit > MAX_LDS_BYTES) return ERR_CODEC_SUPPORT;
    plan->total_blocks = blk;
    plan->uniform_nblocks = uniform_blocks(plan->descs);
    cp.slot_bytes = (cp.max_blocksize + 63) & ~63;
    return 0;
}

struct DecodePlan {
    std::vector<ChunkDesc> descs;
    int32_t total_blocks = 0;
    int32_t lds_bytes = 0;
    int32_t lds_lean = 0;        // LDS of the lean launch: one plane of a 2-byte type (0: blocks too large for it)
    int32_t uniform_nblocks = 0;
};

inline int decode_lds_bound(int blocksize)
{
    int m = 0;
    for (int ts = 1; ts <= MAX_STREAMS; ts++) { const int v = decode_lds_bytes(blocksize, ts); if (v > m) m = v; }
    return m;
}

// nbytes / blocksize come from the chunk headers (the host reads them; blosc2_cbuffer_sizes)
// comp_size (optional): bytes each compressed buffer really holds; it travels in ChunkDesc::destsize (INT32_MAX = unknown) and
// the kernels refuse a header that claims more before they read anything behind the header
inline int plan_decode_batch(int nchunks, const int64_t* comp_off, const int32_t* nbytes, const int32_t* blocksize,
                             const int64_t* raw_off, DecodePlan* plan, const int32_t* comp_size = nullptr)
{
    plan->descs.resize((size_t)nchunks);
    int32_t blk = 0, lds = 0, max_bs = 0;
    for (int i = 0; i < nchunks; i++) {
        ChunkDesc& d = plan->descs[(size_t)i];
        d = ChunkDesc{};
        if (nbytes[i] < 0 || blocksize[i] <= 0 || (nbytes[i] > 0 && blocksize[i] > nbytes[i])) return ERR_INVALID_HEADER;
        d.raw_off = raw_off[i];
        d.comp_off = comp_off[i];
        d.nbytes = nbytes[i];
        d.destsize = comp_size ? comp_size[i] : 0x7fffffff;
        if (d.destsize < HEADER_LEN) return ERR_READ_BUFFER;
        d.blocksize = blocksize[i];
        d.nblocks = nbytes[i] / blocksize[i];
        d.leftover = nbytes[i] % blocksize[i];
        if (d.leftover) d.nblocks++;
        d.blk0 = blk;
        blk += d.nblocks;
        const int need = decode_lds_bound(d.blocksize);
        if (need > lds) lds = need;
        if (d.blocksize > max_bs) max_bs = d.blocksize;
    }
    if (lds > MAX_LDS_BYTES) return ERR_CODEC_SUPPORT;
    plan->total_blocks = blk;
    plan->lds_bytes = lds;
    plan->lds_lean = blz_region_stride(max_bs / 2) + 16;      // decode_lean_kernel.h: the one coded plane of a block (the larger, BloscLZ margin)
    plan->uniform_nblocks = uniform_blocks(plan->descs);
    return 0;
}

}  // namespace cimg
