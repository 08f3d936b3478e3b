/*
 * oracle/orc.h -- CPU restatement of the c-blosc2 chunk codec path (TEST INFRASTRUCTURE).
 *
 * This directory is the *checker*, never the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped path is the HIP library in
 * compressed-image_amd/csrc (libcimg_hip.so) and it has no fallback onto this code.
 *
 * What it restates: everything the reference reaches through
 *   compressed_image/include/compressed/blosc2/wrapper.h:139,172  (blosc2_compress_ctx)
 *   compressed_image/include/compressed/blosc2/wrapper.h:246      (blosc2_decompress_ctx)
 *   compressed_image/include/compressed/blosc2/wrapper.h:453      (blosc2_cbuffer_sizes)
 * i.e. the c-blosc2 (>= 2.17.0, docs/developer/building.rst:17) chunk format: 32-byte extended
 * header, bstarts, per-block filter pipeline (byte shuffle / bit shuffle), typesize-way stream
 * split, run tokens, LZ4 block codec (vendored LZ4 `LZ4_compress_fast`, acceleration 10-clevel),
 * raw-stored streams, memcpyed fallback chunk and the all-zero special chunk.
 *
 * PARITY STATUS: c-blosc2 is an un-vendored, empty submodule in /root/reference
 * (thirdparty/c-blosc2, .gitmodules:1-3,16-18) and no reference test pins compressed bytes
 * (SURVEY.md section 8c), so *compressed-byte* parity against c-blosc2 itself is UNPINNED.
 * What IS pinned (tests/test_oracle_*.py):
 *   - the LZ4 block layer against system liblz4 1.9.3 output (tests/golden/lz4_kat.npz),
 *   - shuffle / split / framing mechanics against c-blosc 1.21 output (tests/golden/blosc1_kat.npz),
 *   - every decompressed-pixel known answer of the reference's OIIO-free tests.
 */
#ifndef ORC_H
#define ORC_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- codec / filter ids (public c-blosc2 values) ---- */
enum { ORC_BLOSCLZ = 0, ORC_LZ4 = 1, ORC_LZ4HC = 2, ORC_ZLIB = 4, ORC_ZSTD = 5 };
enum { ORC_NOFILTER = 0, ORC_SHUFFLE = 1, ORC_BITSHUFFLE = 2 };
enum { ORC_ALWAYS_SPLIT = 1, ORC_NEVER_SPLIT = 2, ORC_AUTO_SPLIT = 3, ORC_FORWARD_COMPAT_SPLIT = 4 };

/* ---- error codes (public c-blosc2 values) ---- */
enum {
    ORC_ERR_FAILURE = -1, ORC_ERR_DATA = -3, ORC_ERR_READ_BUFFER = -5, ORC_ERR_WRITE_BUFFER = -6,
    ORC_ERR_CODEC_SUPPORT = -7, ORC_ERR_CODEC_PARAM = -8, ORC_ERR_VERSION_SUPPORT = -10,
    ORC_ERR_INVALID_HEADER = -11, ORC_ERR_INVALID_PARAM = -12, ORC_ERR_RUN_LENGTH = -17,
    ORC_ERR_MAX_BUFSIZE = -26
};

#define ORC_HEADER_LEN 32
#define ORC_MAX_FILTERS 6

typedef struct {
    int32_t clevel;      /* 0..9 */
    int32_t typesize;    /* sizeof(T) */
    int32_t blocksize;   /* requested block size in bytes (0 unsupported: the reference always sets it) */
    int32_t compcode;    /* ORC_LZ4 / ORC_BLOSCLZ restated; ORC_ZSTD through the box's libzstd (zstd_dl.c); ORC_LZ4HC: the GPU path's substitute (LZ4 fast, acceleration 1), not LZ4_compress_HC */
    int32_t splitmode;   /* ORC_AUTO_SPLIT is what the reference uses (wrapper.h:328,353) */
    uint8_t filters[ORC_MAX_FILTERS];      /* default {0,0,0,0,0,ORC_SHUFFLE} */
    uint8_t filters_meta[ORC_MAX_FILTERS];
} orc_cparams;

void orc_cparams_default(orc_cparams* p);

/* ---- LZ4 block layer (lz4_block.c) ---- */
/* Byte-for-byte behaviour of LZ4_compress_fast(src,dst,n,cap,accel) for n < 65547 (byU16 table).
 * Returns compressed size, 0 if it does not fit `cap` (limited-output rules), <0 if unsupported.
 * If need != NULL and the call succeeds, *need = smallest cap for which the call still succeeds. */
int orc_lz4_compress_fast(const uint8_t* src, int n, uint8_t* dst, int cap, int accel, int* need);
/* Strict LZ4 block decoder: returns decoded size or <0. */
int orc_lz4_decompress_safe(const uint8_t* src, int csize, uint8_t* dst, int cap);

/* ---- BloscLZ stream layer (blosclz.c) ---- */
/* Byte-for-byte behaviour of blosclz_compress(clevel, src, n, dst, cap) of BloscLZ 2.3.0 (see blosclz.c for
 * what that pin means).  Returns compressed size, 0 if not worth it / does not fit.  *need as for LZ4. */
int orc_blosclz_compress(int clevel, const uint8_t* src, int n, uint8_t* dst, int cap, int* need);
int orc_blosclz_decompress(const uint8_t* src, int csize, uint8_t* dst, int cap);
/* the encoder's entropy probe, exposed for tests: counted bytes of the dry run / chosen ipshift (0 = give up) */
int orc_blosclz_probe(const uint8_t* src, int maxlen, int force_3b_shift);
int orc_blosclz_plan(int clevel, const uint8_t* src, int n);

/* ---- zstd streams through the box's own libzstd (zstd_dl.c; dlopen, nothing restated) ---- */
int orc_zstd_available(void);
const char* orc_zstd_version(void);
int orc_zstd_level_of_clevel(int clevel);
int orc_zstd_compress_stream(int clevel, const uint8_t* src, int n, uint8_t* dst, int maxout);   /* 0 = does not fit */
int orc_zstd_decompress_stream(const uint8_t* src, int csize, uint8_t* dst, int cap);

/* ---- filters (filters.c) ---- */
void orc_shuffle(int typesize, int bsize, const uint8_t* src, uint8_t* dst);
void orc_unshuffle(int typesize, int bsize, const uint8_t* src, uint8_t* dst);
void orc_bitshuffle(int typesize, int bsize, const uint8_t* src, uint8_t* dst);
void orc_bitunshuffle(int typesize, int bsize, const uint8_t* src, uint8_t* dst);

/* ---- chunk layer (chunk.c) ---- */
/* Serial (nthreads = 1) layout of blosc2_compress_ctx.  Returns cbytes, 0 if dest too small, <0 error. */
int orc_blosc2_compress(const orc_cparams* p, const void* src, int32_t nbytes, void* dst, int32_t destsize);
/* Same bytes, but blocks are encoded independently first (OpenMP over blocks when nthreads > 1) and
 * then laid out by a serial walk that re-applies the destsize rules from per-stream (size, need)
 * records.  This is the structure the GPU path uses; tests assert it equals orc_blosc2_compress. */
int orc_blosc2_compress_2phase(const orc_cparams* p, const void* src, int32_t nbytes, void* dst,
                               int32_t destsize, int nthreads);
int orc_blosc2_decompress(const void* src, int32_t srcsize, void* dst, int32_t destsize);
/* the same with the blocks of the chunk spread over nthreads OpenMP threads (bench.py's all-cores CPU baseline only) */
int orc_blosc2_decompress_mt(const void* src, int32_t srcsize, void* dst, int32_t destsize, int nthreads);
int orc_blosc2_cbuffer_sizes(const void* cbuffer, int32_t* nbytes, int32_t* cbytes, int32_t* blocksize);

/* Derived geometry, exposed for tests. */
typedef struct {
    int32_t blocksize, nblocks, leftover, split, nstreams_total, flags, memcpyed;
} orc_geometry;
int orc_chunk_geometry(const orc_cparams* p, int32_t nbytes, orc_geometry* g);

#ifdef __cplusplus
}
#endif
#endif
