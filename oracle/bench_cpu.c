/*
 * oracle/bench_cpu.c -- TEST INFRASTRUCTURE (see orc.h): C-level drivers for bench.py's cpu_baseline leg.
 *
 * Two call structures, both over the oracle's own codec:
 *   reference policy  chunks one after the other, exactly as the reference's loops do (schunk.h:85-94, 130-138);
 *                     compression with nthreads_blocks threads over the blocks of a chunk (channel.h:127:
 *                     hardware_concurrency() / 2 inside c-blosc2), decompression with ONE thread (wrapper.h:406).
 *   all cores         an OpenMP loop over chunks, one thread per chunk at a time (no Python in the timed region); with more
 *                     threads than chunks, NESTED: threads_over_chunks teams of threads_over_blocks threads each work on the
 *                     blocks of their chunk (two-phase compress; block-parallel decompress).
 */
#include "orc.h"
#include <omp.h>

/* compress nchunks chunks of chunk_bytes each; chunk i goes to dst + i * dst_stride.  Returns the sum of cbytes, < 0 on error. */
int64_t orc_bench_compress(const orc_cparams* p, const uint8_t* src, int nchunks, int32_t chunk_bytes, uint8_t* dst,
                           int64_t dst_stride, int32_t destsize, int32_t* cbytes, int threads_over_chunks, int threads_over_blocks)
{
    int64_t total = 0;
    int err = 0;
    if (threads_over_chunks > 1) {
        if (threads_over_blocks > 1) omp_set_max_active_levels(2);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads_over_chunks) reduction(+ : total)
        for (int i = 0; i < nchunks; i++) {
            const int r = threads_over_blocks > 1
                ? orc_blosc2_compress_2phase(p, src + (int64_t)i * chunk_bytes, chunk_bytes, dst + i * dst_stride, destsize, threads_over_blocks)
                : orc_blosc2_compress(p, src + (int64_t)i * chunk_bytes, chunk_bytes, dst + i * dst_stride, destsize);
            cbytes[i] = r;
            if (r <= 0) err = 1; else total += r;
        }
    } else {
        for (int i = 0; i < nchunks; i++) {
            const int r = threads_over_blocks > 1
                ? orc_blosc2_compress_2phase(p, src + (int64_t)i * chunk_bytes, chunk_bytes, dst + i * dst_stride, destsize, threads_over_blocks)
                : orc_blosc2_compress(p, src + (int64_t)i * chunk_bytes, chunk_bytes, dst + i * dst_stride, destsize);
            cbytes[i] = r;
            if (r <= 0) err = 1; else total += r;
        }
    }
    return err ? -1 : total;
}

int64_t orc_bench_decompress(const uint8_t* comp, int nchunks, int64_t comp_stride, const int32_t* cbytes, uint8_t* out,
                             int32_t chunk_bytes, int threads_over_chunks, int threads_over_blocks)
{
    int64_t total = 0;
    int err = 0;
    if (threads_over_chunks > 1 && threads_over_blocks > 1) omp_set_max_active_levels(2);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads_over_chunks) reduction(+ : total) if (threads_over_chunks > 1)
    for (int i = 0; i < nchunks; i++) {
        const int r = orc_blosc2_decompress_mt(comp + i * comp_stride, cbytes[i], out + (int64_t)i * chunk_bytes, chunk_bytes, threads_over_blocks);
        if (r != chunk_bytes) err = 1; else total += r;
    }
    return err ? -1 : total;
}
