// blosc2_shim.cpp -- the eleven c-blosc2 entry points the reference binds (include/blosc2.h),
// implemented on the MI355X engine.  One chunk per call, host pointers in and out, exactly like
// c-blosc2; the batched calls in cimg_hip.h are what the re-shaped host loops use instead.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/blosc2.h"
#include "../../include/cimg_hip.h"

struct blosc2_context_s {
    bool compress;
    cimg_cparams cp;
    bool unsupported_params;     // prefilter / dict / non in-memory requests
};

namespace {

std::mutex g_mu;
cimg_engine* g_engine = nullptr;
std::string g_engine_error;

// one engine per process for the single-chunk API, on $CIMG_DEVICE or the current HIP device
cimg_engine* shared_engine()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_engine) return g_engine;
    int dev = -1;
    if (const char* s = getenv("CIMG_DEVICE")) dev = atoi(s);
    if (cimg_engine_create(dev, &g_engine) != 0) {
        g_engine_error = cimg_last_error(nullptr);
        g_engine = nullptr;
    }
    return g_engine;
}

int32_t rd32(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; }

}  // namespace

extern "C" {

cimg_engine* cimg_shared_engine(void) { return shared_engine(); }

int cimg_context_cparams(const blosc2_context_s* context, cimg_cparams* out)
{
    if (!context || !out) return BLOSC2_ERROR_NULL_POINTER;
    if (!context->compress) return BLOSC2_ERROR_INVALID_PARAM;
    *out = context->cp;
    return 0;
}

blosc2_context* blosc2_create_cctx(blosc2_cparams cparams)
{
    blosc2_context* c = new (std::nothrow) blosc2_context_s();
    if (!c) return nullptr;
    c->compress = true;
    cimg_cparams_init(&c->cp, cparams.typesize);
    c->cp.clevel = cparams.clevel;
    c->cp.blocksize = cparams.blocksize;
    c->cp.compcode = cparams.compcode;
    c->cp.splitmode = cparams.splitmode;
    memcpy(c->cp.filters, cparams.filters, BLOSC2_MAX_FILTERS);
    memcpy(c->cp.filters_meta, cparams.filters_meta, BLOSC2_MAX_FILTERS);
    c->unsupported_params = cparams.use_dict != 0 || cparams.prefilter != nullptr;
    return c;
}

blosc2_context* blosc2_create_dctx(blosc2_dparams dparams)
{
    blosc2_context* c = new (std::nothrow) blosc2_context_s();
    if (!c) return nullptr;
    c->compress = false;
    cimg_cparams_init(&c->cp, 1);
    c->unsupported_params = dparams.postfilter != nullptr;
    return c;
}

void blosc2_free_ctx(blosc2_context* context) { delete context; }

int blosc2_compress_ctx(blosc2_context* context, const void* src, int32_t srcsize, void* dest, int32_t destsize)
{
    if (!context || !src || !dest) return BLOSC2_ERROR_NULL_POINTER;
    if (!context->compress) return BLOSC2_ERROR_INVALID_PARAM;
    if (context->unsupported_params) return BLOSC2_ERROR_CODEC_SUPPORT;
    if (srcsize < 0 || srcsize > BLOSC2_MAX_BUFFERSIZE) return BLOSC2_ERROR_MAX_BUFSIZE_EXCEEDED;
    if (destsize < BLOSC2_MAX_OVERHEAD) return BLOSC2_ERROR_MAX_BUFSIZE_EXCEEDED;
    cimg_engine* e = shared_engine();
    if (!e) return BLOSC2_ERROR_FAILURE;
    const int64_t zero = 0;
    int32_t cbytes = 0;
    const int rc = cimg_compress_batch_host(e, &context->cp, 1, src, &zero, &srcsize, dest, &zero, &destsize, &cbytes);
    return rc < 0 ? rc : cbytes;
}

int blosc2_decompress_ctx(blosc2_context* context, const void* src, int32_t srcsize, void* dest, int32_t destsize)
{
    if (!context || !src || !dest) return BLOSC2_ERROR_NULL_POINTER;
    if (context->compress) return BLOSC2_ERROR_INVALID_PARAM;
    if (context->unsupported_params) return BLOSC2_ERROR_CODEC_SUPPORT;
    if (srcsize < BLOSC_MIN_HEADER_LENGTH) return BLOSC2_ERROR_READ_BUFFER;
    int32_t nbytes, cbytes, blocksize;
    int rc = blosc2_cbuffer_sizes(src, &nbytes, &cbytes, &blocksize);
    if (rc < 0) return rc;
    if (cbytes > srcsize) return BLOSC2_ERROR_READ_BUFFER;
    if (nbytes > destsize) return BLOSC2_ERROR_WRITE_BUFFER;
    if (nbytes == 0) return 0;
    cimg_engine* e = shared_engine();
    if (!e) return BLOSC2_ERROR_FAILURE;
    const int64_t zero = 0;
    int32_t status = 0;
    rc = cimg_decompress_batch_host(e, 1, src, &zero, dest, &zero, &destsize, &status);
    return rc < 0 ? rc : nbytes;
}

int blosc2_cbuffer_sizes(const void* cbuffer, int32_t* nbytes, int32_t* cbytes, int32_t* blocksize)
{
    const uint8_t* c = static_cast<const uint8_t*>(cbuffer);
    if (!c) return BLOSC2_ERROR_NULL_POINTER;
    if (c[0] > 5) {
        if (nbytes) *nbytes = 0;
        if (cbytes) *cbytes = 0;
        if (blocksize) *blocksize = 0;
        return BLOSC2_ERROR_VERSION_SUPPORT;
    }
    const int32_t nb = rd32(c + 4), bs = rd32(c + 8), cb = rd32(c + 12);
    if (nbytes) *nbytes = nb;
    if (cbytes) *cbytes = cb;
    if (blocksize) *blocksize = bs;
    if (cb < BLOSC_MIN_HEADER_LENGTH || bs <= 0 || (nb > 0 && bs > nb) || c[3] == 0) return BLOSC2_ERROR_INVALID_HEADER;
    return 0;
}

// ---- in-memory super-chunk: an append-only list of finished chunks ---------------------------------
blosc2_schunk* blosc2_schunk_new(blosc2_storage* storage)
{
    if (storage && storage->urlpath) return nullptr;          // on-disk frames are out of scope
    blosc2_schunk* s = static_cast<blosc2_schunk*>(calloc(1, sizeof(blosc2_schunk)));
    if (!s) return nullptr;
    s->version = 5;
    s->chunksize = -1;
    if (storage && storage->cparams) {
        s->compcode = storage->cparams->compcode;
        s->clevel = storage->cparams->clevel;
        s->typesize = storage->cparams->typesize;
        s->blocksize = storage->cparams->blocksize;
    }
    return s;
}

int blosc2_schunk_free(blosc2_schunk* schunk)
{
    if (!schunk) return 0;
    for (int64_t i = 0; i < schunk->nchunks; i++) free(schunk->data[i]);
    free(schunk->data);
    free(schunk);
    return 0;
}

int64_t blosc2_schunk_append_chunk(blosc2_schunk* schunk, uint8_t* chunk, bool copy)
{
    if (!schunk || !chunk) return BLOSC2_ERROR_NULL_POINTER;
    int32_t nbytes, cbytes, blocksize;
    const int rc = blosc2_cbuffer_sizes(chunk, &nbytes, &cbytes, &blocksize);
    if (rc < 0) return rc;
    if (schunk->chunksize < 0) schunk->chunksize = nbytes;
    if (nbytes > schunk->chunksize && schunk->nchunks > 0) return BLOSC2_ERROR_CODEC_PARAM;   // only the last chunk may be short
    if ((size_t)schunk->nchunks + 1 > schunk->data_len) {
        const size_t cap = schunk->data_len ? schunk->data_len * 2 : 16;
        uint8_t** d = static_cast<uint8_t**>(realloc(schunk->data, cap * sizeof(uint8_t*)));
        if (!d) return BLOSC2_ERROR_MEMORY_ALLOC;
        schunk->data = d;
        schunk->data_len = cap;
    }
    uint8_t* own = chunk;
    if (copy) {
        own = static_cast<uint8_t*>(malloc((size_t)cbytes));
        if (!own) return BLOSC2_ERROR_MEMORY_ALLOC;
        memcpy(own, chunk, (size_t)cbytes);
    }
    schunk->data[schunk->nchunks++] = own;
    schunk->nbytes += nbytes;
    schunk->cbytes += cbytes;
    return schunk->nchunks;
}

void register_filters(void) {}   // the reference calls this before every codec call (wrapper.h:30-36)

const char* print_error(int rc)
{
    switch (rc) {
    case BLOSC2_ERROR_SUCCESS: return "Success";
    case BLOSC2_ERROR_FAILURE: return "Generic failure";
    case BLOSC2_ERROR_STREAM: return "Bad stream";
    case BLOSC2_ERROR_DATA: return "Invalid data";
    case BLOSC2_ERROR_MEMORY_ALLOC: return "Memory alloc/realloc failure";
    case BLOSC2_ERROR_READ_BUFFER: return "Not enough space to read";
    case BLOSC2_ERROR_WRITE_BUFFER: return "Not enough space to write";
    case BLOSC2_ERROR_CODEC_SUPPORT: return "Codec not supported";
    case BLOSC2_ERROR_CODEC_PARAM: return "Invalid parameter supplied to codec";
    case BLOSC2_ERROR_CODEC_DICT: return "Codec dictionary error";
    case BLOSC2_ERROR_VERSION_SUPPORT: return "Version not supported";
    case BLOSC2_ERROR_INVALID_HEADER: return "Invalid value in header";
    case BLOSC2_ERROR_INVALID_PARAM: return "Invalid parameter supplied to function";
    case BLOSC2_ERROR_RUN_LENGTH: return "Bad run length encoding";
    case BLOSC2_ERROR_NULL_POINTER: return "Pointer is null";
    case BLOSC2_ERROR_INVALID_INDEX: return "Invalid index";
    case BLOSC2_ERROR_MAX_BUFSIZE_EXCEEDED: return "Maximum buffersize exceeded";
    default: return "Unknown error";
    }
}

}  // extern "C"
