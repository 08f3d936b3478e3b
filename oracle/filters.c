/*
 * oracle/filters.c -- TEST INFRASTRUCTURE (see orc.h).
 *
 * Byte shuffle and bit shuffle of one block, as c-blosc2's generic (non-SIMD) filters define
 * them.  The reference only ever uses the byte shuffle
 * (BLOSC2_CPARAMS_DEFAULTS filters, never touched: blosc2/wrapper.h:325-332,350-356); the bit
 * shuffle is the BASELINE.json config-3 extension.
 */
#include "orc.h"
#include <string.h>

/* out[j*ne + i] = in[i*ts + j]; the bsize % ts tail bytes are copied verbatim. */
void orc_shuffle(int ts, int bsize, const uint8_t* src, uint8_t* dst)
{
    if (ts <= 1) { memcpy(dst, src, (size_t)bsize); return; }
    const int ne = bsize / ts;
    if (ts == 2) {                                  /* written so that the compiler vectorises it */
        uint8_t* restrict d0 = dst; uint8_t* restrict d1 = dst + ne;
        for (int i = 0; i < ne; i++) { d0[i] = src[2 * i]; d1[i] = src[2 * i + 1]; }
    } else if (ts == 4) {
        uint8_t* restrict d0 = dst; uint8_t* restrict d1 = dst + ne; uint8_t* restrict d2 = dst + 2 * (size_t)ne; uint8_t* restrict d3 = dst + 3 * (size_t)ne;
        for (int i = 0; i < ne; i++) { d0[i] = src[4 * i]; d1[i] = src[4 * i + 1]; d2[i] = src[4 * i + 2]; d3[i] = src[4 * i + 3]; }
    } else {
        for (int j = 0; j < ts; j++)
            for (int i = 0; i < ne; i++)
                dst[(size_t)j * ne + i] = src[(size_t)i * ts + j];
    }
    const int done = ne * ts;
    memcpy(dst + done, src + done, (size_t)(bsize - done));
}

void orc_unshuffle(int ts, int bsize, const uint8_t* src, uint8_t* dst)
{
    if (ts <= 1) { memcpy(dst, src, (size_t)bsize); return; }
    const int ne = bsize / ts;
    if (ts == 2) {
        const uint8_t* restrict s0 = src; const uint8_t* restrict s1 = src + ne;
        for (int i = 0; i < ne; i++) { dst[2 * i] = s0[i]; dst[2 * i + 1] = s1[i]; }
    } else if (ts == 4) {
        const uint8_t* restrict s0 = src; const uint8_t* restrict s1 = src + ne; const uint8_t* restrict s2 = src + 2 * (size_t)ne; const uint8_t* restrict s3 = src + 3 * (size_t)ne;
        for (int i = 0; i < ne; i++) { dst[4 * i] = s0[i]; dst[4 * i + 1] = s1[i]; dst[4 * i + 2] = s2[i]; dst[4 * i + 3] = s3[i]; }
    } else {
        for (int i = 0; i < ne; i++)
            for (int j = 0; j < ts; j++)
                dst[(size_t)i * ts + j] = src[(size_t)j * ne + i];
    }
    const int done = ne * ts;
    memcpy(dst + done, src + done, (size_t)(bsize - done));
}

/*
 * Bit shuffle: only ne8 = ne - ne % 8 elements take part.  Bit-row r = 8*j + k (byte j of the
 * element, bit k of that byte, LSB = 0) holds bit k of byte j of every element, element i at byte
 * i/8, bit i%8 of the row.  Remaining bsize - ne8*ts bytes are copied verbatim.
 */
void orc_bitshuffle(int ts, int bsize, const uint8_t* src, uint8_t* dst)
{
    const int ne = bsize / ts;
    const int ne8 = ne - ne % 8;
    const int rowbytes = ne8 / 8;
    memset(dst, 0, (size_t)ne8 * ts);
    for (int i = 0; i < ne8; i++)
        for (int j = 0; j < ts; j++) {
            const unsigned b = src[(size_t)i * ts + j];
            for (int k = 0; k < 8; k++)
                if (b & (1u << k))
                    dst[(size_t)(8 * j + k) * rowbytes + (i >> 3)] |= (uint8_t)(1u << (i & 7));
        }
    const int done = ne8 * ts;
    memcpy(dst + done, src + done, (size_t)(bsize - done));
}

void orc_bitunshuffle(int ts, int bsize, const uint8_t* src, uint8_t* dst)
{
    const int ne = bsize / ts;
    const int ne8 = ne - ne % 8;
    const int rowbytes = ne8 / 8;
    memset(dst, 0, (size_t)ne8 * ts);
    for (int i = 0; i < ne8; i++)
        for (int j = 0; j < ts; j++) {
            unsigned b = 0;
            for (int k = 0; k < 8; k++)
                if (src[(size_t)(8 * j + k) * rowbytes + (i >> 3)] & (1u << (i & 7))) b |= 1u << k;
            dst[(size_t)i * ts + j] = (uint8_t)b;
        }
    const int done = ne8 * ts;
    memcpy(dst + done, src + done, (size_t)(bsize - done));
}
