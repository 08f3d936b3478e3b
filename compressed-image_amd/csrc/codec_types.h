// codec_types.h -- plain structs shared by the host engine, the gfx950 kernels and the emulator.
//
// Vocabulary follows the reference / c-blosc2: a *chunk* is one blosc2 compressed buffer
// (channel.h / schunk.h: up to 4 MiB of one channel), a chunk is cut into *blocks* (32 KiB,
// constants.h:11), a split block is `typesize` byte-plane *streams*, each stream is one LZ4 block.
#pragma once
#include <stdint.h>

namespace cimg {

enum : int {
    HEADER_LEN = 32,
    FLAG_SHUFFLE = 0x01, FLAG_MEMCPYED = 0x02, FLAG_BITSHUFFLE = 0x04, FLAG_DONT_SPLIT = 0x10,
    FILTER_NONE = 0, FILTER_SHUFFLE = 1, FILTER_BITSHUFFLE = 2,
    CODEC_BLOSCLZ = 0, CODEC_LZ4 = 1, CODEC_LZ4HC = 2, CODEC_ZLIB = 4, CODEC_ZSTD = 5,
    SPLIT_ALWAYS = 1, SPLIT_NEVER = 2, SPLIT_AUTO = 3, SPLIT_FORWARD_COMPAT = 4,
    OFF_FLAGS = 2, OFF_TYPESIZE = 3, OFF_NBYTES = 4, OFF_BLOCKSIZE = 8, OFF_CBYTES = 12,
    OFF_FILTERS = 16, OFF_COMPCODE = 22, OFF_FILTERS_META = 24, OFF_BLOSC2_FLAGS = 31,
    SPECIAL_ZERO = 1, SPECIAL_NAN = 2, SPECIAL_VALUE = 3, SPECIAL_UNINIT = 4,
    MAX_STREAMS = 16, MIN_BUFFERSIZE = 32,
    // stream record kinds written by the encode kernel
    REC_RUN = 0, REC_LZ4 = 1, REC_RAW = 2,
    // error codes (c-blosc2 values) surfaced per chunk
    ERR_FAILURE = -1, ERR_DATA = -3, ERR_READ_BUFFER = -5, ERR_WRITE_BUFFER = -6, ERR_CODEC_SUPPORT = -7,
    ERR_CODEC_PARAM = -8, ERR_VERSION_SUPPORT = -10, ERR_INVALID_HEADER = -11, ERR_INVALID_PARAM = -12,
    ERR_RUN_LENGTH = -17, ERR_MAX_BUFSIZE = -26,
    // internal, never returned to a caller: cimg_decode_blocks met a zstd chunk (codec format 4), which cimg_decode_zstd reads
    // (engine.hip: decompress_finish clears exactly these words before that launch)
    STATUS_ZSTD_PENDING = -1000,            // ... whose blocks are one stream each (one wave per block)
    STATUS_ZSTD_PENDING_SPLIT = -1001,      // ... whose blocks are split into one stream per byte of the element (two waves per block)
};

// one per chunk of a batch; built by the host (engine.cpp: plan_chunk)
struct ChunkDesc {
    int64_t raw_off;      // byte offset of the chunk's pixels in the uncompressed buffer
    int64_t comp_off;     // byte offset of the chunk in the compressed buffer
    int32_t nbytes;       // uncompressed bytes
    int32_t destsize;     // encode: capacity handed to blosc2_compress_ctx for this chunk; decode: bytes the compressed buffer holds (INT32_MAX: unknown)
    int32_t blocksize;    // effective block size
    int32_t nblocks;
    int32_t leftover;     // bytes in the last block if it is short, else 0
    int32_t blk0;         // index of the chunk's first block in the batch-wide block numbering
    int32_t flags;        // header flags byte the chunk starts with
    int32_t split;        // 1: full blocks are cut into `typesize` streams
    int32_t memcpyed;     // 1: clevel 0 or nbytes < 32 -> header + raw bytes
    int32_t nstreams;     // total stream count of the chunk
    int32_t assemble;     // encode: 1 = every stream of the chunk belongs to ONE encode launch, which then lays the chunk out and copies it into place itself
    int32_t pad_;
};

// batch-wide codec parameters
struct CodecParams {
    int32_t typesize;
    int32_t clevel;
    int32_t compcode;
    int32_t filter;        // FILTER_* applied to every block (the reference only uses FILTER_SHUFFLE)
    int32_t accel;         // LZ4 acceleration = 10 - clevel
    int32_t max_blocksize; // largest effective block size in the batch (sizes LDS and scratch slots)
    int32_t slot_bytes;    // scratch bytes reserved per block
    int32_t streams_per_block; // record slots per block (typesize if any chunk splits, else 1)
};

// per-stream record produced by the encode kernel, consumed by the layout kernel
struct StreamRec {
    int32_t kind;     // REC_*
    int32_t value;    // run byte (REC_RUN)
    int32_t csize;    // payload bytes in the scratch slot (REC_LZ4 / REC_RAW)
    int32_t need;     // smallest LZ4 budget under which the stream still compresses (REC_LZ4)
};

// per-chunk result of the layout kernel
struct ChunkLayout {
    int32_t cbytes;      // final chunk size (what blosc2_compress_ctx returns); 0 = does not fit
    int32_t mode;        // 0 regular, 1 memcpyed, 2 special-zero, 3 does not fit
};

}  // namespace cimg
