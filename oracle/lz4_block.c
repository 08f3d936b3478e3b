/*
 * oracle/lz4_block.c -- TEST INFRASTRUCTURE (see orc.h).
 *
 * Restatement of the LZ4 *block* layer as c-blosc2 drives it: one
 * LZ4_compress_fast(stream, dest, n, maxout, accel = 10 - clevel) per split stream and one
 * LZ4_decompress_safe per stream.  LZ4 is a third-party
 * dependency of c-blosc2 (vendored lz4 1.10.0 at c-blosc2 2.17) and is not in /root/reference;
 * this file follows the published block format and the published fast-encoder behaviour and is
 * pinned byte-for-byte against system liblz4 1.9.3 by tests/test_oracle_lz4.py
 * (tests/golden/lz4_kat.npz).  Only the byU16 regime (n < 65547) is restated: the reference's
 * block size is 32 KiB (constants.h:11), so every stream it ever produces is in that regime.
 */
#include "orc.h"
#include <string.h>

enum {
    MINMATCH = 4, MFLIMIT = 12, LASTLITERALS = 5, MINLENGTH = MFLIMIT + 1,
    ML_BITS = 4, ML_MASK = 15, RUN_MASK = 15, SKIP_TRIGGER = 6,
    LIMIT_64K = 65536 + MFLIMIT - 1, HASH_LOG_U16 = 13
};

static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t hash_u16(uint32_t v) { return (v * 2654435761u) >> (32 - HASH_LOG_U16); }

static inline int max_i(int a, int b) { return a > b ? a : b; }

/* length of the common prefix of a[0..) and b[0..), a limited to a_end */
static int common_len(const uint8_t* a, const uint8_t* b, const uint8_t* a_end)
{
    const uint8_t* s = a;
    while (a + 8 <= a_end) {                       /* eight bytes at a time (same answer as the byte loop) */
        uint64_t x, y;
        memcpy(&x, a, 8); memcpy(&y, b, 8);
        if (x != y) return (int)(a - s) + (__builtin_ctzll(x ^ y) >> 3);
        a += 8; b += 8;
    }
    while (a < a_end && *a == *b) { a++; b++; }
    return (int)(a - s);
}

int orc_lz4_compress_fast(const uint8_t* src, int n, uint8_t* dst, int cap, int accel, int* need_out)
{
    if (accel < 1) accel = 1;
    if (accel > 65537) accel = 65537;
    if (n < 0) return 0;
    if (n >= LIMIT_64K) return ORC_ERR_CODEC_SUPPORT; /* byU32 regime: not on the reference's path */
    const int limited = cap < n + n / 255 + 16;
    if (n == 0) {
        if (limited && cap <= 0) return 0;
        dst[0] = 0;
        if (need_out) *need_out = 1;
        return 1;
    }

    uint16_t table[1 << HASH_LOG_U16];
    memset(table, 0, sizeof(table));

    int ip = 0, anchor = 0, op = 0, need = 0;
    const int iend = n;
    const int mflimit_p1 = n - MFLIMIT + 1;
    const int matchlimit = n - LASTLITERALS;

    if (n >= MINLENGTH) {
        table[hash_u16(rd32(src))] = 0;
        ip = 1;
        for (;;) {
            int match, token;
            /* --- search: probe positions with the skip schedule --- */
            {
                int probe = ip, step = 1, nb = accel << SKIP_TRIGGER;
                for (;;) {
                    const int cur = probe;
                    const int next = cur + step;
                    step = nb++ >> SKIP_TRIGGER;
                    if (next > mflimit_p1) goto last_literals;
                    const uint32_t h = hash_u16(rd32(src + cur));
                    match = table[h];
                    table[h] = (uint16_t)cur;
                    if (rd32(src + match) == rd32(src + cur)) { ip = cur; break; }
                    probe = next;
                }
            }
            /* --- extend backwards --- */
            while (ip > anchor && match > 0 && src[ip - 1] == src[match - 1]) { ip--; match--; }

            /* --- literals --- */
            {
                const int lit = ip - anchor;
                token = op++;
                const int lhs = op + lit + (2 + 1 + LASTLITERALS) + lit / 255;
                if (limited && lhs > cap) return 0;
                need = max_i(need, lhs);
                if (lit >= RUN_MASK) {
                    int len = lit - RUN_MASK;
                    dst[token] = (uint8_t)(RUN_MASK << ML_BITS);
                    for (; len >= 255; len -= 255) dst[op++] = 255;
                    dst[op++] = (uint8_t)len;
                } else {
                    dst[token] = (uint8_t)(lit << ML_BITS);
                }
                memcpy(dst + op, src + anchor, (size_t)lit);
                op += lit;
            }
        next_match:
            /* --- offset + match length --- */
            {
                const int off = ip - match;
                dst[op++] = (uint8_t)(off & 0xFF);
                dst[op++] = (uint8_t)(off >> 8);
                int mcode = common_len(src + ip + MINMATCH, src + match + MINMATCH, src + matchlimit);
                ip += mcode + MINMATCH;
                const int lhs = op + (1 + LASTLITERALS) + (mcode + 240) / 255;
                if (limited && lhs > cap) return 0;
                need = max_i(need, lhs);
                if (mcode >= ML_MASK) {
                    dst[token] += ML_MASK;
                    mcode -= ML_MASK;
                    for (; mcode >= 255; mcode -= 255) dst[op++] = 255;
                    dst[op++] = (uint8_t)mcode;
                } else {
                    dst[token] += (uint8_t)mcode;
                }
            }
            anchor = ip;
            if (ip >= mflimit_p1) break;

            /* --- refill table, test the position right after the match --- */
            table[hash_u16(rd32(src + ip - 2))] = (uint16_t)(ip - 2);
            {
                const uint32_t h = hash_u16(rd32(src + ip));
                match = table[h];
                table[h] = (uint16_t)ip;
                if (rd32(src + match) == rd32(src + ip)) {
                    token = op++;
                    dst[token] = 0;
                    goto next_match;
                }
            }
            ip++;
        }
    }
last_literals:
    {
        const int run = iend - anchor;
        const int lhs = op + run + 1 + (run + 255 - RUN_MASK) / 255;
        if (limited && lhs > cap) return 0;
        need = max_i(need, lhs);
        if (run >= RUN_MASK) {
            int acc = run - RUN_MASK;
            dst[op++] = (uint8_t)(RUN_MASK << ML_BITS);
            for (; acc >= 255; acc -= 255) dst[op++] = 255;
            dst[op++] = (uint8_t)acc;
        } else {
            dst[op++] = (uint8_t)(run << ML_BITS);
        }
        memcpy(dst + op, src + anchor, (size_t)run);
        op += run;
    }
    if (need_out) *need_out = need;
    return op;
}

int orc_lz4_decompress_safe(const uint8_t* src, int csize, uint8_t* dst, int cap)
{
    if (csize <= 0 || cap < 0) return -1;
    int ip = 0, op = 0;
    for (;;) {
        if (ip >= csize) return -1;
        const unsigned token = src[ip++];
        int lit = (int)(token >> ML_BITS);
        if (lit == RUN_MASK) {
            unsigned b;
            do {
                if (ip >= csize) return -1;
                b = src[ip++];
                lit += (int)b;
            } while (b == 255);
        }
        if (lit > csize - ip || lit > cap - op) return -1;
        memcpy(dst + op, src + ip, (size_t)lit);
        ip += lit; op += lit;
        if (ip == csize) break;                 /* a block ends with a literal-only sequence */
        if (csize - ip < 2) return -1;
        const int off = src[ip] | (src[ip + 1] << 8);
        ip += 2;
        if (off == 0 || off > op) return -1;
        int mlen = (int)(token & ML_MASK);
        if (mlen == ML_MASK) {
            unsigned b;
            do {
                if (ip >= csize) return -1;
                b = src[ip++];
                mlen += (int)b;
            } while (b == 255);
        }
        mlen += MINMATCH;
        if (mlen > cap - op) return -1;
        if (off >= mlen) memcpy(dst + op, dst + op - off, (size_t)mlen);
        else if (off == 1) memset(dst + op, dst[op - 1], (size_t)mlen);
        else for (int k = 0; k < mlen; k++) dst[op + k] = dst[op + k - off];   /* overlap replicates */
        op += mlen;
    }
    return op;
}
