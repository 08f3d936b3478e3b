/*
 * blosc2.h -- the slice of the c-blosc2 C API that EmilDohne/compressed-image binds, served by
 * libcimg_hip.so (MI355X / gfx950) instead of c-blosc2.
 *
 * The reference funnels every codec call through compressed/blosc2/wrapper.h (+ util.h, schunk.h:113,
 * lazyschunk.h:153,163) and uses exactly the symbols below.  Names, argument
 * meaning and error convention (negative int = BLOSC2_ERROR_*) follow c-blosc2 >= 2.17
 * (docs/developer/building.rst:17).  Struct layouts are this library's own: the reference is compiled
 * against THIS header, it does not need c-blosc2's binary layout.
 *
 *   reference call site                      symbol
 *   wrapper.h:139,172                        blosc2_compress_ctx
 *   wrapper.h:246                            blosc2_decompress_ctx
 *   wrapper.h:368,377                        blosc2_create_cctx
 *   wrapper.h:395,412                        blosc2_create_dctx
 *   wrapper.h:58                             blosc2_free_ctx
 *   wrapper.h:453                            blosc2_cbuffer_sizes
 *   wrapper.h:309                            blosc2_schunk_new
 *   wrapper.h:49                             blosc2_schunk_free
 *   wrapper.h:286, schunk.h:113              blosc2_schunk_append_chunk
 *   wrapper.h:34                             register_filters
 *   blosc2/util.h:18                         print_error
 */
#ifndef CIMG_BLOSC2_SHIM_H
#define CIMG_BLOSC2_SHIM_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLOSC2_VERSION_STRING "2.17.0-cimg-hip"
#define BLOSC2_MAX_FILTERS 6
#define BLOSC_MIN_HEADER_LENGTH 16
#define BLOSC_EXTENDED_HEADER_LENGTH 32
#define BLOSC2_MAX_OVERHEAD BLOSC_EXTENDED_HEADER_LENGTH
#define BLOSC2_MAX_BUFFERSIZE (INT32_MAX - BLOSC2_MAX_OVERHEAD)

enum { BLOSC_BLOSCLZ = 0, BLOSC_LZ4 = 1, BLOSC_LZ4HC = 2, BLOSC_ZLIB = 4, BLOSC_ZSTD = 5 };
enum { BLOSC_NOSHUFFLE = 0, BLOSC_NOFILTER = 0, BLOSC_SHUFFLE = 1, BLOSC_BITSHUFFLE = 2, BLOSC_DELTA = 3, BLOSC_TRUNC_PREC = 4 };
enum { BLOSC_ALWAYS_SPLIT = 1, BLOSC_NEVER_SPLIT = 2, BLOSC_AUTO_SPLIT = 3, BLOSC_FORWARD_COMPAT_SPLIT = 4 };

enum {
    BLOSC2_ERROR_SUCCESS = 0, BLOSC2_ERROR_FAILURE = -1, BLOSC2_ERROR_STREAM = -2, BLOSC2_ERROR_DATA = -3,
    BLOSC2_ERROR_MEMORY_ALLOC = -4, BLOSC2_ERROR_READ_BUFFER = -5, BLOSC2_ERROR_WRITE_BUFFER = -6,
    BLOSC2_ERROR_CODEC_SUPPORT = -7, BLOSC2_ERROR_CODEC_PARAM = -8, BLOSC2_ERROR_CODEC_DICT = -9,
    BLOSC2_ERROR_VERSION_SUPPORT = -10, BLOSC2_ERROR_INVALID_HEADER = -11, BLOSC2_ERROR_INVALID_PARAM = -12,
    BLOSC2_ERROR_FILE_READ = -13, BLOSC2_ERROR_FILE_WRITE = -14, BLOSC2_ERROR_FILE_OPEN = -15,
    BLOSC2_ERROR_NOT_FOUND = -16, BLOSC2_ERROR_RUN_LENGTH = -17, BLOSC2_ERROR_FILE_TRUNCATE = -18,
    BLOSC2_ERROR_THREAD_CREATE = -19, BLOSC2_ERROR_POSTFILTER = -20, BLOSC2_ERROR_FRAME_TYPE = -21,
    BLOSC2_ERROR_FILE_REMOVE = -22, BLOSC2_ERROR_NULL_POINTER = -23, BLOSC2_ERROR_INVALID_INDEX = -24,
    BLOSC2_ERROR_METALAYER_NOT_FOUND = -25, BLOSC2_ERROR_MAX_BUFSIZE_EXCEEDED = -26
};

typedef struct blosc2_context_s blosc2_context;
typedef struct blosc2_schunk blosc2_schunk;

typedef struct {
    uint8_t compcode;        /* BLOSC_* codec */
    uint8_t compcode_meta;
    uint8_t clevel;          /* 0..9 */
    int use_dict;            /* must be 0 */
    int32_t typesize;
    int16_t nthreads;        /* accepted, ignored: parallelism is the GPU's */
    int32_t blocksize;       /* 0 = automatic (not supported by this library: the reference always sets it) */
    int32_t splitmode;
    void* schunk;
    uint8_t filters[BLOSC2_MAX_FILTERS];
    uint8_t filters_meta[BLOSC2_MAX_FILTERS];
    void* prefilter;         /* must be NULL */
    void* preparams;
    void* tuner_params;
    int tuner_id;
    bool instr_codec;
    void* codec_params;
    void* filter_params[BLOSC2_MAX_FILTERS];
} blosc2_cparams;

typedef struct {
    int16_t nthreads;        /* accepted, ignored */
    void* schunk;
    void* postfilter;        /* must be NULL */
    void* postparams;
} blosc2_dparams;

typedef struct {
    bool contiguous;
    char* urlpath;           /* must be NULL: in-memory super-chunks only */
    blosc2_cparams* cparams;
    blosc2_dparams* dparams;
    void* io;
} blosc2_storage;

/* In-memory super-chunk: an ordered list of finished chunks (what schunk.h:107-121 exports to). */
struct blosc2_schunk {
    uint8_t version;
    uint8_t compcode;
    uint8_t clevel;
    int32_t typesize;
    int32_t blocksize;
    int32_t chunksize;       /* nbytes of the first chunk appended, -1 until then */
    int64_t nchunks;         /* test/src/test_schunk.cpp:66 */
    int64_t nbytes;          /* test/src/test_schunk.cpp:67 */
    int64_t cbytes;
    uint8_t** data;          /* chunk pointers, owned */
    size_t data_len;         /* capacity of `data` */
    blosc2_storage* storage;
};

static const blosc2_cparams BLOSC2_CPARAMS_DEFAULTS = {
    BLOSC_BLOSCLZ, 0, 5, 0, 8, 1, 0, BLOSC_FORWARD_COMPAT_SPLIT, NULL,
    {0, 0, 0, 0, 0, BLOSC_SHUFFLE}, {0, 0, 0, 0, 0, 0}, NULL, NULL, NULL, 0, false, NULL,
    {NULL, NULL, NULL, NULL, NULL, NULL}};
static const blosc2_dparams BLOSC2_DPARAMS_DEFAULTS = {1, NULL, NULL, NULL};
static const blosc2_storage BLOSC2_STORAGE_DEFAULTS = {false, NULL, NULL, NULL, NULL};

blosc2_context* blosc2_create_cctx(blosc2_cparams cparams);
blosc2_context* blosc2_create_dctx(blosc2_dparams dparams);
void blosc2_free_ctx(blosc2_context* context);

/* Compress `srcsize` bytes at host pointer `src` into one blosc2 chunk at host pointer `dest`
 * (capacity destsize).  Returns the chunk size, 0 if it does not fit, < 0 on error. */
int blosc2_compress_ctx(blosc2_context* context, const void* src, int32_t srcsize, void* dest, int32_t destsize);
/* Decompress one chunk; returns the number of decompressed bytes or < 0. */
int blosc2_decompress_ctx(blosc2_context* context, const void* src, int32_t srcsize, void* dest, int32_t destsize);
int blosc2_cbuffer_sizes(const void* cbuffer, int32_t* nbytes, int32_t* cbytes, int32_t* blocksize);

blosc2_schunk* blosc2_schunk_new(blosc2_storage* storage);
int blosc2_schunk_free(blosc2_schunk* schunk);
int64_t blosc2_schunk_append_chunk(blosc2_schunk* schunk, uint8_t* chunk, bool copy);

void register_filters(void);
const char* print_error(int rc);

#ifdef __cplusplus
}
#endif
#endif
