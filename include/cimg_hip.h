/*
 * cimg_hip.h -- C ABI of libcimg_hip.so, the MI355X (gfx950) chunk codec engine.
 *
 * Two groups of entry points:
 *
 *  (1) include/blosc2.h -- the eleven c-blosc2 symbols the reference binds
 *      (compressed/blosc2/wrapper.h, blosc2/util.h, blosc2/schunk.h:113), same names and argument
 *      meaning, so the reference's own headers link against this library instead of c-blosc2.
 *
 *  (2) this header -- the *batched* extension.  One blosc2_compress_ctx call is one <= 4 MiB chunk
 *      (wrapper.h:139,172), far too little to fill 256 CUs; the reference's loops over chunks
 *      (schunk.h:85-104, schunk.h:130-138) and over channels (image.h:119-159, 1307-1315) become one
 *      call here.  Plain pointers and sizes only; no HIP or torch types.
 *
 * All functions return 0 (or a non-negative size) on success and a negative c-blosc2 error code
 * (include/blosc2.h, BLOSC2_ERROR_*) on failure; cimg_last_error() gives the text.  Every call takes
 * the engine's own (recursive) lock, so calls from several threads are serialised, not interleaved;
 * a _begin / _fetch pair is made one unit by holding cimg_engine_lock() across it.  (The reference's
 * contexts are not thread-safe at all, channel.h:511-513.)
 */
#ifndef CIMG_HIP_H
#define CIMG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cimg_engine cimg_engine;

/* Mirrors the blosc2_cparams fields the reference sets (wrapper.h:325-332, 350-356) plus the filter
 * pipeline it inherits from BLOSC2_CPARAMS_DEFAULTS. */
typedef struct cimg_cparams {
    int32_t typesize;        /* sizeof(T) */
    int32_t clevel;          /* 0..9 */
    int32_t blocksize;       /* bytes; constants.h:11 default 32768 */
    int32_t compcode;        /* BLOSC_LZ4 = 1, BLOSC_BLOSCLZ = 0: the CPU codec's bytes; BLOSC_LZ4HC = 2, BLOSC_ZSTD = 5: format-valid
                              * streams (any LZ4 / zstd decoder reads them), NOT liblz4-HC's / libzstd's bytes.  BLOSC_ZLIB: refused. */
    int32_t splitmode;       /* BLOSC_AUTO_SPLIT = 3 */
    uint8_t filters[6];      /* default {0,0,0,0,0,BLOSC_SHUFFLE} */
    uint8_t filters_meta[6];
} cimg_cparams;

void cimg_cparams_init(cimg_cparams* p, int32_t typesize);   /* the reference's defaults: lz4, level 9, 32 KiB blocks */

/* ---- engine ---------------------------------------------------------------------------------------- */
int  cimg_engine_create(int device, cimg_engine** out);      /* device < 0: current device */
void cimg_engine_destroy(cimg_engine* e);
const char* cimg_last_error(const cimg_engine* e);            /* e may be NULL: last create() failure */
int  cimg_engine_synchronize(cimg_engine* e);
/* Every batch call takes the engine's (recursive) lock for its own duration.  A caller that needs two calls to act as
 * one -- _begin followed by _fetch below -- brackets them with these, on the same thread. */
void cimg_engine_lock(cimg_engine* e);
void cimg_engine_unlock(cimg_engine* e);
/* The hipStream_t the engine's work is ordered on.  A batch call behaves as if everything it does were enqueued there, in order:
 * work the caller put on this stream before the call (a kernel producing the pixels, cimg_deinterleave_device) is seen by the
 * batch, work put there after a _begin call follows the batch.  (Internally a batch may run one of its launches on a second
 * stream beside the first -- the leftover blocks of chunks that are no multiple of the block size --; that stream is made to
 * wait for everything this one holds when the batch is enqueued, and this one for it before the batch ends.)
 * An encode launch is persistent: it is sized to fill the device, and it completes with any number of its workgroups resident
 * (two engines, or two processes, sharing one card slow each other down; they do not wait for each other). */
void* cimg_engine_stream(cimg_engine* e);

/* ---- device-resident batches -----------------------------------------------------------------------
 * Chunk i's pixels live at d_raw + raw_off[i] (nbytes[i] bytes), its blosc2 chunk at
 * d_comp + comp_off[i] with capacity destsize[i] (what the reference passes as the dest span:
 * nominal chunk size + BLOSC2_MAX_OVERHEAD, schunk.h:73).  Offset / size arrays are HOST arrays.
 * cbytes[i] receives what blosc2_compress_ctx would return for that chunk (0 = does not fit).
 * One encode launch per kind of block (split into byte planes / unsplit) does everything: the waves that encoded a chunk also
 * lay it out and copy it into place.
 * The call returns after the results are on the host (one stream sync). */
int cimg_compress_batch_device(cimg_engine* e, const cimg_cparams* p, int32_t nchunks,
                               const void* d_raw, const int64_t* raw_off, const int32_t* nbytes,
                               void* d_comp, const int64_t* comp_off, const int32_t* destsize,
                               int32_t* cbytes);
/* nbytes[i] / blocksize[i] are the values in chunk i's header (blosc2_cbuffer_sizes); status[i]
 * receives 0 or the blosc2 error code of that chunk. Returns the first non-zero status, or 0.
 * Chunks of codec format 0 (blosclz), 1 (lz4, lz4hc) and 4 (zstd: a slow path, launched only when such a
 * chunk is in the batch) decode; any other format is BLOSC2_ERROR_CODEC_SUPPORT for that chunk. */
int cimg_decompress_batch_device(cimg_engine* e, int32_t nchunks,
                                 const void* d_comp, const int64_t* comp_off,
                                 const int32_t* nbytes, const int32_t* blocksize,
                                 void* d_raw, const int64_t* raw_off, int32_t* status);

/* The same in two steps: _begin enqueues the kernels on the engine's stream and returns at once, _fetch waits for them
 * and hands over the results.  ONE compress batch and ONE decompress batch may be in flight together, and a decompress
 * _begin may follow the compress _begin of the chunks it reads directly (stream order: it sees them finished; nbytes /
 * blocksize are the caller's, nothing of the compress results is needed on the host) -- so the host's share of a round
 * trip (planning, launches, the wait) hides behind the kernels instead of leaving the GPU idle between the calls.
 * Each _fetch refers to the most recent _begin of its kind; a plain batch call of the same kind in between invalidates
 * it (error).  Threads sharing an engine hold cimg_engine_lock from _begin to _fetch. */
int cimg_compress_batch_device_begin(cimg_engine* e, const cimg_cparams* p, int32_t nchunks,
                                     const void* d_raw, const int64_t* raw_off, const int32_t* nbytes,
                                     void* d_comp, const int64_t* comp_off, const int32_t* destsize);
int cimg_compress_batch_device_fetch(cimg_engine* e, int32_t nchunks, int32_t* cbytes);
int cimg_decompress_batch_device_begin(cimg_engine* e, int32_t nchunks,
                                       const void* d_comp, const int64_t* comp_off,
                                       const int32_t* nbytes, const int32_t* blocksize,
                                       void* d_raw, const int64_t* raw_off);
int cimg_decompress_batch_device_fetch(cimg_engine* e, int32_t* status);
/* Device-resident decode for callers that know how many bytes each compressed buffer really holds (comp_size[i]; at least 32).
 * The reference passes INT32_MAX as srcsize (blosc2/wrapper.h:249), so the callee is the one that must be careful: a header
 * whose cbytes exceeds comp_size[i] gets BLOSC2_ERROR_READ_BUFFER for that chunk and the kernels read nothing of it behind
 * the 32-byte header (the unsized calls above trust the header's cbytes).  Otherwise as the unsized calls. */
int cimg_decompress_batch_device_sized(cimg_engine* e, int32_t nchunks,
                                       const void* d_comp, const int64_t* comp_off, const int32_t* comp_size,
                                       const int32_t* nbytes, const int32_t* blocksize,
                                       void* d_raw, const int64_t* raw_off, int32_t* status);
int cimg_decompress_batch_device_begin_sized(cimg_engine* e, int32_t nchunks,
                                             const void* d_comp, const int64_t* comp_off, const int32_t* comp_size,
                                             const int32_t* nbytes, const int32_t* blocksize,
                                             void* d_raw, const int64_t* raw_off);

/* ---- interleaved pixels -> planes (reference: image_algo::deinterleave, compressed/image_algo.h:84-111, the step between
 * reading scanlines and compressing them in the read path, image.h:1880) -------------------------------------------
 * d_interleaved holds npixels * nchannels elements of `typesize` bytes (R G B A R G B A ...); channel c comes out as
 * npixels elements at d_planar + c * plane_stride.  plane_stride is in bytes, a multiple of 16, at least npixels * typesize;
 * both buffers are 16-byte aligned device memory.  typesize is 1, 2, 4 or 8.  Returns when the kernel is enqueued on the
 * engine's stream (later calls on the same engine see its result). */
int cimg_deinterleave_device(cimg_engine* e, const void* d_interleaved, int32_t nchannels, int32_t typesize, int64_t npixels,
                             void* d_planar, int64_t plane_stride);
/* The producer path in one call: interleaved scanlines in HOST memory are uploaded once, split into planes on the device
 * and compressed from there.  Chunk i covers nbytes[i] bytes at offset raw_off[i] of the PLANAR layout, in which channel c
 * occupies [c * plane_stride, c * plane_stride + npixels * typesize) with plane_stride = npixels * typesize rounded up to
 * 16.  Otherwise as cimg_compress_batch_host_begin: the chunks stay on the device, cbytes[] comes back, and
 * cimg_compress_batch_host_fetch delivers them. */
int cimg_compress_batch_host_interleaved_begin(cimg_engine* e, const cimg_cparams* p, int32_t nchannels, int64_t npixels,
                                               const void* h_interleaved, int32_t nchunks, const int64_t* raw_off,
                                               const int32_t* nbytes, const int32_t* destsize, int32_t* cbytes);

/* ---- host-resident batches (H2D + kernels + D2H inside) ------------------------------------------ */
int cimg_compress_batch_host(cimg_engine* e, const cimg_cparams* p, int32_t nchunks,
                             const void* h_raw, const int64_t* raw_off, const int32_t* nbytes,
                             void* h_comp, const int64_t* comp_off, const int32_t* destsize,
                             int32_t* cbytes);
/* The same in two steps, for host code that wants to allocate the chunks' final storage with their exact sizes:
 * _begin uploads and compresses (the chunks stay in the engine's device staging area) and reports cbytes[];
 * _fetch copies chunk i (cbytes[i] bytes) to h_comp + comp_off[i] and synchronizes.  _fetch refers to the most
 * recent _begin on this engine; any other batch call in between invalidates it (error). */
int cimg_compress_batch_host_begin(cimg_engine* e, const cimg_cparams* p, int32_t nchunks,
                                   const void* h_raw, const int64_t* raw_off, const int32_t* nbytes,
                                   const int32_t* destsize, int32_t* cbytes);
int cimg_compress_batch_host_fetch(cimg_engine* e, int32_t nchunks, void* h_comp, const int64_t* comp_off);
/* The same in ONE call, for a caller that wants the chunks in host memory at exactly their compressed sizes (the host mirror's
 * schunk / image constructors: chunks are std::vector-like buffers of cbytes): the batch runs as a pipeline of groups of about
 * 16 MiB of pixels, and as soon as the sizes of a group are known the engine asks `alloc(user, bytes)` for a block of host memory
 * (page-locked for full PCIe speed: cimg_host_malloc) and sends the group's chunks there, packed back to back at 64-byte
 * boundaries, while the next group is being compressed.  chunk_ptr[i] = where chunk i went (NULL for a chunk that does not fit, cbytes
 * 0).  Returns when every chunk has arrived.  (cimg_compress_batch_host_begin + _fetch move the chunks only after the LAST group:
 * 4.2 against 3.0 ms for 4 x 4096^2 float16.) */
typedef void* (*cimg_alloc_fn)(void* user, size_t bytes);
int cimg_compress_batch_host_packed(cimg_engine* e, const cimg_cparams* p, int32_t nchunks, const void* h_raw, const int64_t* raw_off,
                                    const int32_t* nbytes, const int32_t* destsize, int32_t* cbytes, cimg_alloc_fn alloc, void* user, void** chunk_ptr);
/* Sizes are read from the chunk headers in host memory (the reference passes INT32_MAX as srcsize, wrapper.h:249,
 * so the header is all there is). */
int cimg_decompress_batch_host(cimg_engine* e, int32_t nchunks,
                               const void* h_comp, const int64_t* comp_off,
                               void* h_raw, const int64_t* raw_off, const int32_t* raw_capacity,
                               int32_t* status);
/* The same for callers that know how many bytes each chunk buffer really holds: a header that claims more
 * (truncated or hostile chunk) is refused with BLOSC2_ERROR_READ_BUFFER before anything is read past the buffer. */
int cimg_decompress_batch_host_sized(cimg_engine* e, int32_t nchunks,
                                     const void* h_comp, const int64_t* comp_off, const int32_t* comp_size,
                                     void* h_raw, const int64_t* raw_off, const int32_t* raw_capacity,
                                     int32_t* status);

/* ---- glue between the blosc2 shim and the batched calls ----------------------------------------------
 * The single-chunk blosc2_*_ctx calls run on one process-wide engine (device $CIMG_DEVICE, else the
 * current HIP device); cimg_shared_engine() hands it out so that host code holding blosc2 contexts
 * (the re-shaped schunk / image loops) can batch on the same stream.  NULL + cimg_last_error(NULL) on
 * failure.  cimg_context_cparams() copies the codec parameters a compression context was created with. */
struct blosc2_context_s;
cimg_engine* cimg_shared_engine(void);
int cimg_context_cparams(const struct blosc2_context_s* context, cimg_cparams* out);

/* ---- device memory without HIP headers ------------------------------------------------------------ */
void* cimg_device_malloc(cimg_engine* e, size_t bytes);
void  cimg_device_free(cimg_engine* e, void* p);
int   cimg_memcpy_h2d(cimg_engine* e, void* d_dst, const void* h_src, size_t bytes);
int   cimg_memcpy_d2h(cimg_engine* e, void* h_dst, const void* d_src, size_t bytes);

/* ---- page-locked host memory ---------------------------------------------------------------------------
 * The host-buffer batch calls move pixels and chunks over PCIe; from / to ordinary (pageable) memory the copy
 * is staged by the runtime and a freshly allocated destination is page-faulted in on the way (measured: 64 MiB
 * of pixels into a new numpy array 22 ms, into a recycled page-locked buffer 2 ms).  Host code that owns its
 * buffers (the Python binding's result arrays, the compress staging area) takes them from here.  NULL on failure. */
void* cimg_host_malloc(size_t bytes);
void  cimg_host_free(void* p);

/* ---- kernel timing (HIP events on the engine's stream) ------------------------------------------- */
enum { CIMG_K_ENCODE = 0, CIMG_K_LAYOUT = 1, CIMG_K_EMIT = 2, CIMG_K_DECODE = 3, CIMG_K_DEINTERLEAVE = 4, CIMG_K_DECODE_ZSTD = 5, CIMG_K_ENCODE_ZSTD = 6,
       /* CIMG_K_DECODE_ZSTD is the whole zstd read path of a batch; its launches are also timed one by one: */
       CIMG_K_ZSTD_WALK = 7, CIMG_K_ZSTD_REPLAY = 8, CIMG_K_ZSTD_FUSED = 9, CIMG_K_ZSTD_SEQ = 10, CIMG_K_ZSTD_LIT = 11, CIMG_K_COUNT = 12 };
/* on = 0: off; on = n > 0: the kernels of every n-th batch call are bracketed by events (1 = every call).  Each
 * event record costs about 5 us of stream time, so a throughput run samples (bench.py: every 4th batch). */
void cimg_engine_enable_timing(cimg_engine* e, int on);
void cimg_engine_reset_timing(cimg_engine* e);
/* total milliseconds and launch count of one kernel since the last reset (syncs the stream) */
int  cimg_engine_kernel_time(cimg_engine* e, int kernel, double* total_ms, int64_t* launches);
/* the duration of every timed launch since the last reset, in milliseconds, oldest first (for medians); returns how many were
 * copied (at most max_samples), < 0 on error */
int  cimg_engine_kernel_samples(cimg_engine* e, int kernel, float* ms, int max_samples);
/* decode batches since the engine was created: how many went through the lean launch (cimg_decode_lean), how many blocks of
 * those batches it left to cimg_decode_blocks (leftover blocks included), their block total, and how many batches needed
 * cimg_decode_zstd.  Any pointer may be NULL. */
void cimg_engine_decode_stats(cimg_engine* e, int64_t* lean_batches, int64_t* blocks_left_to_general, int64_t* blocks_total, int64_t* zstd_batches);
/* the zstd read path since the engine was created: batches that needed it, and blocks whose plan did not fit its slot (the walk
 * refused them and cimg_decode_zstd decoded them behind the other launches).  Any pointer may be NULL. */
void cimg_engine_zstd_stats(cimg_engine* e, int64_t* zstd_batches, int64_t* blocks_refused);
const char* cimg_kernel_name(int kernel);

/* ---- diagnostics: per-workgroup clock stamps of the most recent encode (0) / decode (1) launch ------
 * 16 uint64 per workgroup: four stamps of {shader clock, 100 MHz clock, HW_ID, XCC_ID} (start, stream 0
 * staged, stream 1 staged, end).  Off by
 * default; when on, the NEXT launches stamp (the dbg pointer is null otherwise and the kernels skip it). */
void cimg_engine_debug_stamps(cimg_engine* e, int on);
int  cimg_engine_read_stamps(cimg_engine* e, int which, uint64_t* out, int max_workgroups);

#ifdef __cplusplus
}
#endif
#endif
