/*
 * oracle/zstd_dl.c -- TEST INFRASTRUCTURE (see orc.h): the zstd leg of the checker.
 *
 * c-blosc2 codes every stream of a chunk written with BLOSC_ZSTD as ONE complete zstd frame (ZSTD_compressCCtx with
 * level = 2 * clevel - 1, clevel 8 -> ZSTD_maxCLevel() - 2, clevel 9 -> ZSTD_maxCLevel(); decode with ZSTD_decompressDCtx)
 * [UPSTREAM-RECALL of blosc/blosc2.c: zstd_wrap_compress / zstd_wrap_decompress].  Nothing of zstd is restated here: the
 * box's own libzstd.so.1 is dlopen()ed, so that (a) chunks as the reference would write them with enums::codec::zstd
 * (compressed_image/include/compressed/enums.h:18-24, blosc2/wrapper.h:74-119) can be MADE at full size for the decode
 * tests and bench.py --config 5, and (b) chunks the GPU zstd encoder makes can be decoded by the real library.
 * Where there is no libzstd the calls return ORC_ERR_CODEC_SUPPORT and the tests skip.
 */
#include "orc.h"
#include <dlfcn.h>
#include <stddef.h>

typedef size_t (*zstd_compress_fn)(void*, size_t, const void*, size_t, int);
typedef size_t (*zstd_decompress_fn)(void*, size_t, const void*, size_t);
typedef unsigned (*zstd_iserror_fn)(size_t);
typedef int (*zstd_maxlevel_fn)(void);
typedef const char* (*zstd_version_fn)(void);

static struct {
    int tried, ok;
    zstd_compress_fn compress;
    zstd_decompress_fn decompress;
    zstd_iserror_fn is_error;
    zstd_maxlevel_fn max_level;
    zstd_version_fn version;
} Z;

static int zstd_load(void)
{
    if (Z.tried) return Z.ok;
#pragma omp critical(orc_zstd_load)
    {
        if (!Z.tried) {
            void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                Z.compress = (zstd_compress_fn)dlsym(h, "ZSTD_compress");
                Z.decompress = (zstd_decompress_fn)dlsym(h, "ZSTD_decompress");
                Z.is_error = (zstd_iserror_fn)dlsym(h, "ZSTD_isError");
                Z.max_level = (zstd_maxlevel_fn)dlsym(h, "ZSTD_maxCLevel");
                Z.version = (zstd_version_fn)dlsym(h, "ZSTD_versionString");
                Z.ok = Z.compress && Z.decompress && Z.is_error && Z.max_level;
            }
            Z.tried = 1;
        }
    }
    return Z.ok;
}

int orc_zstd_available(void) { return zstd_load(); }
const char* orc_zstd_version(void) { return zstd_load() && Z.version ? Z.version() : ""; }

int orc_zstd_level_of_clevel(int clevel)
{
    if (!zstd_load()) return 0;
    if (clevel >= 9) return Z.max_level();
    if (clevel == 8) return Z.max_level() - 2;
    return 2 * clevel - 1;
}

/* one stream -> one frame; 0 = does not fit maxout (c-blosc2 then stores the stream raw), < 0 = no library */
int orc_zstd_compress_stream(int clevel, const uint8_t* src, int n, uint8_t* dst, int maxout)
{
    if (!zstd_load()) return ORC_ERR_CODEC_SUPPORT;
    const size_t r = Z.compress(dst, (size_t)maxout, src, (size_t)n, orc_zstd_level_of_clevel(clevel));
    if (Z.is_error(r)) return 0;
    return (int)r;
}

/* one frame -> exactly cap bytes, else ORC_ERR_DATA */
int orc_zstd_decompress_stream(const uint8_t* src, int csize, uint8_t* dst, int cap)
{
    if (!zstd_load()) return ORC_ERR_CODEC_SUPPORT;
    const size_t r = Z.decompress(dst, (size_t)cap, src, (size_t)csize);
    if (Z.is_error(r)) return ORC_ERR_DATA;
    return (int)r;
}
