/*
 * oracle/chunk.c -- TEST INFRASTRUCTURE (see orc.h).
 *
 * Restatement of the c-blosc2 chunk layer behind blosc2_compress_ctx / blosc2_decompress_ctx /
 * blosc2_cbuffer_sizes as the reference calls them (blosc2/wrapper.h:139,172,246,453) with the
 * cparams it builds (wrapper.h:338-359: BLOSC2_CPARAMS_DEFAULTS + blocksize, typesize,
 * BLOSC_AUTO_SPLIT, clevel, compcode; default filter pipeline = byte shuffle in the last slot).
 * c-blosc2 itself is not in /root/reference; this follows its published chunk format and the
 * nthreads = 1 block order.  Compressed-byte parity against a real
 * c-blosc2 build is UNPINNED (see orc.h).
 *
 * Layout of a regular chunk:
 *   [0]  version = 5   [1] versionlz = 1   [2] flags   [3] typesize
 *   [4:8] nbytes  [8:12] blocksize  [12:16] cbytes           (int32 LE)
 *   [16:22] filters  [22] compcode  [23] compcode_meta  [24:30] filters_meta  [30] 0  [31] blosc2_flags
 *   int32 bstarts[nblocks]      absolute offset of every block
 *   per block, per stream: int32 csize, payload
 *       csize == 0         -> stream is all zero bytes, no payload
 *       csize  < 0         -> stream is a run of byte (-csize); one token byte 0x01 follows
 *       csize == streamlen -> payload stored raw
 *       otherwise          -> LZ4 block (codec format 1: lz4, lz4hc) or BloscLZ stream (codec format 0)
 */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

enum { FLAG_SHUFFLE = 0x01, FLAG_MEMCPYED = 0x02, FLAG_BITSHUFFLE = 0x04, FLAG_DONT_SPLIT = 0x10 };
enum { MIN_BUFFERSIZE = 32, MAX_STREAMS = 16, MAX_TYPESIZE = 255, VERSION_FORMAT = 5 };
enum { SPECIAL_ZERO = 1, SPECIAL_NAN = 2, SPECIAL_VALUE = 3, SPECIAL_UNINIT = 4 };
enum { OFF_FLAGS = 2, OFF_TYPESIZE = 3, OFF_NBYTES = 4, OFF_BLOCKSIZE = 8, OFF_CBYTES = 12,
       OFF_FILTERS = 16, OFF_COMPCODE = 22, OFF_FILTERS_META = 24, OFF_BLOSC2_FLAGS = 31 };

static void put32(uint8_t* p, int32_t v) { uint32_t u = (uint32_t)v; p[0] = (uint8_t)u; p[1] = (uint8_t)(u >> 8); p[2] = (uint8_t)(u >> 16); p[3] = (uint8_t)(u >> 24); }
static int32_t get32(const uint8_t* p) { return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }

void orc_cparams_default(orc_cparams* p)
{
    memset(p, 0, sizeof(*p));
    p->clevel = 5;
    p->typesize = 8;
    p->compcode = ORC_BLOSCLZ;
    p->splitmode = ORC_FORWARD_COMPAT_SPLIT;
    p->filters[ORC_MAX_FILTERS - 1] = ORC_SHUFFLE;
}

static int compformat_of(int compcode)
{
    switch (compcode) {
    case ORC_BLOSCLZ: return 0;
    case ORC_LZ4: case ORC_LZ4HC: return 1;
    case ORC_ZLIB: return 3;
    case ORC_ZSTD: return 4;
    default: return -1;
    }
}

static int filter_flags_of(const orc_cparams* p)
{
    int f = 0;
    for (int i = 0; i < ORC_MAX_FILTERS; i++) {
        if (p->filters[i] == ORC_SHUFFLE) f |= FLAG_SHUFFLE;
        if (p->filters[i] == ORC_BITSHUFFLE) f |= FLAG_BITSHUFFLE;
    }
    return f;
}

static int wants_split(const orc_cparams* p, int typesize, int blocksize)
{
    if (p->splitmode == ORC_ALWAYS_SPLIT) return 1;
    if (p->splitmode == ORC_NEVER_SPLIT) return 0;
    const int fast = p->compcode == ORC_BLOSCLZ || p->compcode == ORC_LZ4 ||
                     (p->compcode == ORC_ZSTD && p->clevel <= 5);
    return fast && (filter_flags_of(p) & FLAG_SHUFFLE) && typesize <= MAX_STREAMS &&
           blocksize / typesize >= MIN_BUFFERSIZE;
}

int orc_chunk_geometry(const orc_cparams* p, int32_t nbytes, orc_geometry* g)
{
    memset(g, 0, sizeof(*g));
    if (nbytes < 0) return ORC_ERR_MAX_BUFSIZE;
    if (p->clevel < 0 || p->clevel > 9) return ORC_ERR_CODEC_PARAM;
    if (compformat_of(p->compcode) < 0) return ORC_ERR_CODEC_SUPPORT;
    int ts = p->typesize;
    if (ts <= 0) return ORC_ERR_INVALID_PARAM;
    if (ts > MAX_TYPESIZE) ts = 1;
    int bs;
    if (nbytes < ts) {
        bs = 1;
    } else {
        if (p->blocksize <= 0) return ORC_ERR_INVALID_PARAM;   /* automatic block size: not on the path */
        bs = p->blocksize;
        if (bs < MIN_BUFFERSIZE) bs = MIN_BUFFERSIZE;   /* SURVEY N2 [UPSTREAM-RECALL]: compute_blocksize raises a forced block size below
                                                        * BLOSC_MIN_BUFFERSIZE to it, then clips to nbytes, then rounds down to the typesize */
        if (bs > nbytes) bs = nbytes;
        if (bs > ts) bs = bs / ts * ts;
    }
    g->blocksize = bs;
    g->nblocks = nbytes / bs;
    g->leftover = nbytes % bs;
    if (g->leftover) g->nblocks++;
    g->memcpyed = (p->clevel == 0) || (nbytes < MIN_BUFFERSIZE);
    g->flags = FLAG_SHUFFLE | FLAG_BITSHUFFLE;      /* both set == "extended header" */
    /* ONE rule for byte 2 of the header (round 3): upstream's write_compression_header ORs the dont-split bit
     * (bit 4, from split_block) and the codec format (bits 5-7) into header_flags in one place, whether or not the chunk
     * was marked memcpyed up front (clevel 0, nbytes < 32) -- so an up-front memcpyed chunk carries them exactly like a
     * chunk that fell back to memcpyed after compression did not fit (finish_memcpyed keeps g->flags).  [UPSTREAM-RECALL]:
     * to be re-checked by tests/test_conformance_cblosc2.py the day a libblosc2 is present. */
    {
        const int split = wants_split(p, ts, bs);
        if (!split) g->flags |= FLAG_DONT_SPLIT;
        g->flags |= compformat_of(p->compcode) << 5;
        if (g->memcpyed) g->flags |= FLAG_MEMCPYED;
        else g->split = split;
    }
    if (g->split)
        g->nstreams_total = g->leftover ? (g->nblocks - 1) * ts + 1 : g->nblocks * ts;
    else
        g->nstreams_total = g->nblocks;
    return 0;
}

static void write_header(const orc_cparams* p, const orc_geometry* g, int32_t nbytes, uint8_t* dst)
{
    memset(dst, 0, ORC_HEADER_LEN);
    dst[0] = VERSION_FORMAT;
    dst[1] = 1;
    dst[OFF_FLAGS] = (uint8_t)g->flags;
    dst[OFF_TYPESIZE] = (uint8_t)(p->typesize > MAX_TYPESIZE ? 1 : p->typesize);
    put32(dst + OFF_NBYTES, nbytes);
    put32(dst + OFF_BLOCKSIZE, g->blocksize);
    for (int i = 0; i < ORC_MAX_FILTERS; i++) {
        dst[OFF_FILTERS + i] = p->filters[i];
        dst[OFF_FILTERS_META + i] = p->filters_meta[i];
    }
    dst[OFF_COMPCODE] = (uint8_t)p->compcode;
}

static int all_equal(const uint8_t* s, int n)
{
    if (n >= 16) {                                  /* s[0..n-8) == s[8..n): the period-8 test, then the first 8 bytes */
        if (memcmp(s, s + 8, (size_t)n - 8) != 0) return 0;
        n = 8;
    }
    for (int i = 1; i < n; i++) if (s[i] != s[0]) return 0;
    return 1;
}

/* forward filter pipeline of one block; returns pointer to the filtered bytes (a or b or src) */
static const uint8_t* filters_forward(const orc_cparams* p, int ts, int bsize, const uint8_t* src,
                                      uint8_t* a, uint8_t* b)
{
    const uint8_t* cur = src;
    uint8_t* out = a;
    for (int i = 0; i < ORC_MAX_FILTERS; i++) {
        if (p->filters[i] == ORC_SHUFFLE) orc_shuffle(ts, bsize, cur, out);
        else if (p->filters[i] == ORC_BITSHUFFLE) orc_bitshuffle(ts, bsize, cur, out);
        else continue;
        cur = out;
        out = (out == a) ? b : a;
    }
    return cur;
}

static int codec_compress(const orc_cparams* p, const uint8_t* s, int n, uint8_t* d, int maxout, int* need)
{
    if (p->compcode == ORC_LZ4) return orc_lz4_compress_fast(s, n, d, maxout, 10 - p->clevel, need);
    if (p->compcode == ORC_BLOSCLZ) return orc_blosclz_compress(p->clevel, s, n, d, maxout, need);
    if (p->compcode == ORC_ZSTD) {          /* the box's libzstd (zstd_dl.c): a frame fits iff the budget holds its bytes */
        const int r = orc_zstd_compress_stream(p->clevel, s, n, d, maxout);
        if (need && r > 0) *need = r;
        return r;
    }
    /* lz4hc: NOT LZ4_compress_HC.  The GPU path writes lz4hc chunks as LZ4 blocks from the fast match finder at acceleration 1
     * (format-valid -- codec format 1, any LZ4 decoder reads them -- but not liblz4's HC bytes), and this
     * is the checker's twin of THAT, so that the GPU's lz4hc chunks can be compared byte for byte with something. */
    if (p->compcode == ORC_LZ4HC) return orc_lz4_compress_fast(s, n, d, maxout, 1, need);
    return ORC_ERR_CODEC_SUPPORT;
}

/* one block, written at dst (= chunk + ntbytes).  Returns block bytes, 0 = does not fit, <0 error. */
static int encode_block(const orc_cparams* p, const orc_geometry* g, int ts, int bsize, int leftoverblock,
                        int32_t ntbytes, int32_t destsize, const uint8_t* src, uint8_t* dst,
                        uint8_t* tmpa, uint8_t* tmpb)
{
    const uint8_t* f = filters_forward(p, ts, bsize, src, tmpa, tmpb);
    const int nstreams = (g->split && !leftoverblock) ? ts : 1;
    const int neblock = bsize / nstreams;
    int ctbytes = 0;
    for (int j = 0; j < nstreams; j++) {
        const uint8_t* s = f + (size_t)j * neblock;
        dst += 4; ntbytes += 4; ctbytes += 4;
        if (all_equal(s, neblock)) {
            const int value = s[0];
            if (ntbytes > destsize) return 0;
            put32(dst - 4, -value);
            if (value > 0) {
                ntbytes += 1; ctbytes += 1;
                if (ntbytes > destsize) return 0;
                *dst++ = 0x01;
            }
            continue;
        }
        int maxout = neblock;
        if (ntbytes + maxout > destsize) {
            maxout = destsize - ntbytes;
            if (maxout <= 0) return 0;
        }
        int cbytes = codec_compress(p, s, neblock, dst, maxout, NULL);
        if (cbytes < 0) return cbytes;
        if (cbytes > maxout) return ORC_ERR_WRITE_BUFFER;
        if (cbytes == 0 || cbytes == neblock) {
            if (ntbytes + neblock > destsize) return 0;
            memcpy(dst, s, (size_t)neblock);
            cbytes = neblock;
        }
        put32(dst - 4, cbytes);
        dst += cbytes; ntbytes += cbytes; ctbytes += cbytes;
    }
    return ctbytes;
}

static int finish_memcpyed(const orc_geometry* g, const uint8_t* src, int32_t nbytes, uint8_t* dst, int32_t destsize)
{
    if ((int64_t)nbytes + ORC_HEADER_LEN > destsize) { put32(dst + OFF_CBYTES, 0); return 0; }
    memcpy(dst + ORC_HEADER_LEN, src, (size_t)nbytes);
    dst[OFF_FLAGS] = (uint8_t)(g->flags | FLAG_MEMCPYED);
    put32(dst + OFF_CBYTES, nbytes + ORC_HEADER_LEN);
    return nbytes + ORC_HEADER_LEN;
}

static int finish_regular(const orc_geometry* g, int32_t ntbytes, uint8_t* dst)
{
    if (ntbytes == ORC_HEADER_LEN + 4 * g->nblocks + 4 * g->nstreams_total) {
        dst[OFF_BLOSC2_FLAGS] |= SPECIAL_ZERO << 4;     /* every stream is a zero run */
        ntbytes = ORC_HEADER_LEN;
    }
    put32(dst + OFF_CBYTES, ntbytes);
    return ntbytes;
}

int orc_blosc2_compress(const orc_cparams* p, const void* src_, int32_t nbytes, void* dst_, int32_t destsize)
{
    const uint8_t* src = (const uint8_t*)src_;
    uint8_t* dst = (uint8_t*)dst_;
    orc_geometry g;
    int rc = orc_chunk_geometry(p, nbytes, &g);
    if (rc < 0) return rc;
    if (destsize < ORC_HEADER_LEN) return ORC_ERR_MAX_BUFSIZE;
    const int ts = p->typesize > MAX_TYPESIZE ? 1 : p->typesize;
    write_header(p, &g, nbytes, dst);
    if (g.memcpyed) return finish_memcpyed(&g, src, nbytes, dst, destsize);

    uint8_t* tmpa = (uint8_t*)malloc((size_t)g.blocksize * 2 + 16);
    if (!tmpa) return ORC_ERR_FAILURE;
    uint8_t* tmpb = tmpa + g.blocksize + 8;
    int32_t ntbytes = ORC_HEADER_LEN + 4 * g.nblocks;
    int fits = ntbytes <= destsize;                 /* bstarts[] must fit before anything is written */
    for (int j = 0; j < g.nblocks && fits; j++) {
        int bsize = g.blocksize, lo = 0;
        if (j == g.nblocks - 1 && g.leftover) { bsize = g.leftover; lo = 1; }
        put32(dst + ORC_HEADER_LEN + 4 * j, ntbytes);
        int cb = encode_block(p, &g, ts, bsize, lo, ntbytes, destsize, src + (size_t)j * g.blocksize,
                              dst + ntbytes, tmpa, tmpb);
        if (cb < 0) { free(tmpa); return cb; }
        if (cb == 0) { fits = 0; break; }
        ntbytes += cb;
    }
    free(tmpa);
    if (!fits) return finish_memcpyed(&g, src, nbytes, dst, destsize);
    return finish_regular(&g, ntbytes, dst);
}

/* ---------------------------------------------------------------------------------------------
 * Two-phase form: phase 1 encodes every block on its own against the *unclipped* stream budget
 * and records (kind, size, need) per stream; phase 2 walks the records serially and applies the
 * running-offset / destsize rules of encode_block above.  A clipped budget only ever turns an LZ4
 * success into "does not fit" (the emitted bytes never depend on the budget), so `need` (the
 * smallest budget under which LZ4 still succeeds) makes phase 2 exact.
 * ------------------------------------------------------------------------------------------- */
typedef struct { int32_t kind, value, csize, need; } stream_rec;   /* kind: 0 run, 1 lz4, 2 raw */

int orc_blosc2_compress_2phase(const orc_cparams* p, const void* src_, int32_t nbytes, void* dst_,
                               int32_t destsize, int nthreads)
{
    const uint8_t* src = (const uint8_t*)src_;
    uint8_t* dst = (uint8_t*)dst_;
    orc_geometry g;
    int rc = orc_chunk_geometry(p, nbytes, &g);
    if (rc < 0) return rc;
    if (destsize < ORC_HEADER_LEN) return ORC_ERR_MAX_BUFSIZE;
    const int ts = p->typesize > MAX_TYPESIZE ? 1 : p->typesize;
    write_header(p, &g, nbytes, dst);
    if (g.memcpyed) return finish_memcpyed(&g, src, nbytes, dst, destsize);
    if (p->compcode != ORC_LZ4 && p->compcode != ORC_BLOSCLZ && p->compcode != ORC_ZSTD && p->compcode != ORC_LZ4HC) return ORC_ERR_CODEC_SUPPORT;

    const int maxstreams = g.split ? ts : 1;
    const size_t slot = (size_t)g.blocksize + 16;
    stream_rec* recs = (stream_rec*)calloc((size_t)g.nblocks * maxstreams, sizeof(stream_rec));
    uint8_t* scratch = (uint8_t*)malloc((size_t)g.nblocks * slot);      /* LZ4 payloads, stream-major per block */
    uint8_t* filtered = (uint8_t*)malloc((size_t)g.nblocks * slot);     /* filtered blocks (raw payload source) */
    if (!recs || !scratch || !filtered) { free(recs); free(scratch); free(filtered); return ORC_ERR_FAILURE; }
    int err = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) if (nthreads > 1)
    for (int j = 0; j < g.nblocks; j++) {
        int bsize = g.blocksize, lo = 0;
        if (j == g.nblocks - 1 && g.leftover) { bsize = g.leftover; lo = 1; }
        uint8_t* tmp = (uint8_t*)malloc((size_t)bsize * 2 + 16);
        const uint8_t* f = filters_forward(p, ts, bsize, src + (size_t)j * g.blocksize, tmp, tmp + bsize + 8);
        uint8_t* fdst = filtered + (size_t)j * slot;
        memcpy(fdst, f, (size_t)bsize);
        free(tmp);
        const int nstreams = (g.split && !lo) ? ts : 1;
        const int neblock = bsize / nstreams;
        for (int s = 0; s < nstreams; s++) {
            stream_rec* r = &recs[(size_t)j * maxstreams + s];
            const uint8_t* sp = fdst + (size_t)s * neblock;
            if (all_equal(sp, neblock)) { r->kind = 0; r->value = sp[0]; continue; }
            int need = 0;
            int cb = codec_compress(p, sp, neblock, scratch + (size_t)j * slot + (size_t)s * neblock, neblock, &need);
            if (cb < 0) { err = cb; continue; }
            if (cb == 0 || cb == neblock) { r->kind = 2; r->csize = neblock; }
            else { r->kind = 1; r->csize = cb; r->need = need; }
        }
    }
    int32_t ntbytes = ORC_HEADER_LEN + 4 * g.nblocks;
    int fits = !err && ntbytes <= destsize;
    for (int j = 0; j < g.nblocks && fits; j++) {
        int bsize = g.blocksize, lo = 0;
        if (j == g.nblocks - 1 && g.leftover) { bsize = g.leftover; lo = 1; }
        const int nstreams = (g.split && !lo) ? ts : 1;
        const int neblock = bsize / nstreams;
        put32(dst + ORC_HEADER_LEN + 4 * j, ntbytes);
        for (int s = 0; s < nstreams && fits; s++) {
            const stream_rec* r = &recs[(size_t)j * maxstreams + s];
            ntbytes += 4;
            if (r->kind == 0) {
                if (ntbytes > destsize) { fits = 0; break; }
                put32(dst + ntbytes - 4, -r->value);
                if (r->value > 0) {
                    ntbytes += 1;
                    if (ntbytes > destsize) { fits = 0; break; }
                    dst[ntbytes - 1] = 0x01;
                }
                continue;
            }
            int maxout = neblock;
            if (ntbytes + maxout > destsize) {
                maxout = destsize - ntbytes;
                if (maxout <= 0) { fits = 0; break; }
            }
            if (r->kind == 1 && r->need <= maxout) {
                put32(dst + ntbytes - 4, r->csize);
                memcpy(dst + ntbytes, scratch + (size_t)j * slot + (size_t)s * neblock, (size_t)r->csize);
                ntbytes += r->csize;
            } else {
                if (ntbytes + neblock > destsize) { fits = 0; break; }
                put32(dst + ntbytes - 4, neblock);
                memcpy(dst + ntbytes, filtered + (size_t)j * slot + (size_t)s * neblock, (size_t)neblock);
                ntbytes += neblock;
            }
        }
    }
    free(recs); free(scratch); free(filtered);
    if (err) return err;
    if (!fits) return finish_memcpyed(&g, src, nbytes, dst, destsize);
    return finish_regular(&g, ntbytes, dst);
}

int orc_blosc2_cbuffer_sizes(const void* cbuffer, int32_t* nbytes, int32_t* cbytes, int32_t* blocksize)
{
    const uint8_t* c = (const uint8_t*)cbuffer;
    if (c[0] > VERSION_FORMAT) { if (nbytes) *nbytes = 0; if (cbytes) *cbytes = 0; if (blocksize) *blocksize = 0; return ORC_ERR_VERSION_SUPPORT; }
    const int32_t nb = get32(c + OFF_NBYTES), bs = get32(c + OFF_BLOCKSIZE), cb = get32(c + OFF_CBYTES);
    if (nbytes) *nbytes = nb;
    if (cbytes) *cbytes = cb;
    if (blocksize) *blocksize = bs;
    if (cb < 16 || bs <= 0 || (nb > 0 && bs > nb) || c[OFF_TYPESIZE] == 0) return ORC_ERR_INVALID_HEADER;
    return 0;
}

/* one block of a regular chunk: streams -> tmpa, backward filter pipeline -> dst + j * blocksize.  0 or an error code. */
static int decode_one_block(const uint8_t* src, int32_t cbytes, int32_t blocksize, int nblocks, int leftover, int ts, int dont_split,
                            int compformat, int j, uint8_t* dst, uint8_t* tmpa, uint8_t* tmpb)
{
    const uint8_t* filters = src + OFF_FILTERS;
    int bsize = blocksize, lo = 0;
    if (j == nblocks - 1 && leftover) { bsize = leftover; lo = 1; }
    const int32_t bstart = get32(src + ORC_HEADER_LEN + 4 * j);
    if (bstart < ORC_HEADER_LEN + 4 * nblocks || bstart > cbytes) return ORC_ERR_DATA;
    const uint8_t* ip = src + bstart;
    int32_t left = cbytes - bstart;
    const int nstreams = (!dont_split && !lo) ? ts : 1;
    const int neblock = bsize / nstreams;
    for (int s = 0; s < nstreams; s++) {
        uint8_t* out = tmpa + (size_t)s * neblock;
        if (left < 4) return ORC_ERR_READ_BUFFER;
        int32_t cs = get32(ip); ip += 4; left -= 4;
        if (cs == 0) { memset(out, 0, (size_t)neblock); continue; }
        if (cs < 0) {
            if (left < 1) return ORC_ERR_READ_BUFFER;
            const int token = *ip++; left--;
            if (!(token & 1) || cs < -255) return ORC_ERR_RUN_LENGTH;
            memset(out, -cs, (size_t)neblock);
            continue;
        }
        if (cs > left) return ORC_ERR_READ_BUFFER;
        if (cs == neblock) memcpy(out, ip, (size_t)neblock);
        else if ((compformat == 1 ? orc_lz4_decompress_safe(ip, cs, out, neblock)
                  : compformat == 4 ? orc_zstd_decompress_stream(ip, cs, out, neblock)
                                    : orc_blosclz_decompress(ip, cs, out, neblock)) != neblock) return ORC_ERR_DATA;
        ip += cs; left -= cs;
    }
    /* backward filter pipeline */
    uint8_t* cur = tmpa; uint8_t* other = tmpb;
    int last = -1;
    for (int i = 0; i < ORC_MAX_FILTERS; i++)
        if (filters[i] == ORC_SHUFFLE || filters[i] == ORC_BITSHUFFLE) last = i;
    uint8_t* final_out = dst + (size_t)j * blocksize;
    if (last < 0) memcpy(final_out, cur, (size_t)bsize);
    for (int i = ORC_MAX_FILTERS - 1; i >= 0; i--) {
        if (filters[i] != ORC_SHUFFLE && filters[i] != ORC_BITSHUFFLE) continue;
        int first = 1;
        for (int k = 0; k < i; k++) if (filters[k] == ORC_SHUFFLE || filters[k] == ORC_BITSHUFFLE) first = 0;
        uint8_t* out = first ? final_out : other;
        if (filters[i] == ORC_SHUFFLE) orc_unshuffle(ts, bsize, cur, out);
        else orc_bitunshuffle(ts, bsize, cur, out);
        other = cur; cur = out;
    }
    return 0;
}

int orc_blosc2_decompress(const void* src_, int32_t srcsize, void* dst_, int32_t destsize)
{
    return orc_blosc2_decompress_mt(src_, srcsize, dst_, destsize, 1);
}

int orc_blosc2_decompress_mt(const void* src_, int32_t srcsize, void* dst_, int32_t destsize, int nthreads)
{
    const uint8_t* src = (const uint8_t*)src_;
    uint8_t* dst = (uint8_t*)dst_;
    if (srcsize < ORC_HEADER_LEN) return ORC_ERR_READ_BUFFER;
    int32_t nbytes, cbytes, blocksize;
    int rc = orc_blosc2_cbuffer_sizes(src, &nbytes, &cbytes, &blocksize);
    if (rc < 0) return rc;
    if (cbytes > srcsize) return ORC_ERR_READ_BUFFER;
    if (nbytes > destsize) return ORC_ERR_WRITE_BUFFER;
    if (nbytes == 0) return 0;
    const int flags = src[OFF_FLAGS];
    const int ts = src[OFF_TYPESIZE];
    if ((flags & (FLAG_SHUFFLE | FLAG_BITSHUFFLE)) != (FLAG_SHUFFLE | FLAG_BITSHUFFLE))
        return ORC_ERR_VERSION_SUPPORT;        /* Blosc1-style short header: not produced by the path */
    const int special = (src[OFF_BLOSC2_FLAGS] >> 4) & 7;
    if (special == SPECIAL_ZERO) { memset(dst, 0, (size_t)nbytes); return nbytes; }
    if (special == SPECIAL_UNINIT) return nbytes;
    if (special == SPECIAL_VALUE) {
        if (cbytes < ORC_HEADER_LEN + ts || nbytes % ts) return ORC_ERR_DATA;
        for (int32_t i = 0; i < nbytes; i += ts) memcpy(dst + i, src + ORC_HEADER_LEN, (size_t)ts);
        return nbytes;
    }
    if (special == SPECIAL_NAN) {
        if (ts == 4) { const uint32_t q = 0x7FC00000u; for (int32_t i = 0; i + 4 <= nbytes; i += 4) memcpy(dst + i, &q, 4); return nbytes; }
        if (ts == 8) { const uint64_t q = 0x7FF8000000000000ull; for (int32_t i = 0; i + 8 <= nbytes; i += 8) memcpy(dst + i, &q, 8); return nbytes; }
        return ORC_ERR_DATA;
    }
    if (special != 0) return ORC_ERR_DATA;
    if (flags & FLAG_MEMCPYED) {
        if (cbytes != nbytes + ORC_HEADER_LEN) return ORC_ERR_DATA;
        memcpy(dst, src + ORC_HEADER_LEN, (size_t)nbytes);
        return nbytes;
    }
    const int compformat = flags >> 5;
    if (compformat != 0 && compformat != 1 && compformat != 4) return ORC_ERR_CODEC_SUPPORT;   /* blosclz; lz4 and lz4hc share format 1; zstd = 4 */
    if (compformat == 4 && !orc_zstd_available()) return ORC_ERR_CODEC_SUPPORT;
    const int dont_split = (flags & FLAG_DONT_SPLIT) != 0;
    int nblocks = nbytes / blocksize;
    const int leftover = nbytes % blocksize;
    if (leftover) nblocks++;
    if (cbytes < ORC_HEADER_LEN + 4 * nblocks) return ORC_ERR_READ_BUFFER;
    rc = nbytes;
    if (nthreads > 1 && nblocks > 1) {
        /* blocks are independent (bstarts[]): the all-cores CPU baseline of bench.py spreads them over threads, like c-blosc2's
         * parallel_blosc does for nthreads > 1 (the reference itself decodes with ONE thread, blosc2/wrapper.h:406) */
        int err = 0;
#pragma omp parallel num_threads(nthreads)
        {
            uint8_t* ta = (uint8_t*)malloc((size_t)blocksize * 2 + 16);
#pragma omp for schedule(dynamic, 4)
            for (int j = 0; j < nblocks; j++) {
                const int r = ta ? decode_one_block(src, cbytes, blocksize, nblocks, leftover, ts, dont_split, compformat, j, dst, ta, ta + blocksize + 8)
                                 : ORC_ERR_FAILURE;
                if (r < 0) {
#pragma omp critical
                    { if (!err) err = r; }
                }
            }
            free(ta);
        }
        return err ? err : rc;
    }
    uint8_t* tmpa = (uint8_t*)malloc((size_t)blocksize * 2 + 16);
    if (!tmpa) return ORC_ERR_FAILURE;
    uint8_t* tmpb = tmpa + blocksize + 8;
    for (int j = 0; j < nblocks; j++) {
        const int r = decode_one_block(src, cbytes, blocksize, nblocks, leftover, ts, dont_split, compformat, j, dst, tmpa, tmpb);
        if (r < 0) { rc = r; break; }
    }
    free(tmpa);
    return rc;
}
