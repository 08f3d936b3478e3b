/*
 * oracle/blosclz.c -- TEST INFRASTRUCTURE (see orc.h).
 *
 * CPU restatement of the BloscLZ stream codec that c-blosc2 calls per stream when the reference
 * asks for enums::codec::blosclz (compressed_image/include/compressed/enums.h:18-24, mapped to
 * BLOSC_BLOSCLZ at blosc2/wrapper.h:74-119; used by the reference's own tests at
 * test/src/test_channel.cpp:74-87 and python/test/test_channel.py:50-69).
 *
 * PIN: the algorithm is the one published as blosc/blosclz.c (functions get_csize, get_run / get_match,
 * blosclz_compress, blosclz_decompress).  c-blosc2 is absent from /root/reference (empty submodule), so
 * the constants and the order of every check below were pinned against the only BloscLZ build present in
 * this image: BloscLZ 2.3.0 inside the system library /opt/conda/lib/libblosc.so.1 (c-blosc 1.21.0),
 * byte for byte on the vectors of tests/golden/blosclz_kat.npz (tests/golden/make_blosclz_golden.py).
 * c-blosc2 >= 2.17 vendors a LATER BloscLZ (2.5.x: other hash log, probe length and thresholds), so
 * compressed bytes are "pinned to c-blosc1's blosclz 2.3.0, unverified vs c-blosc2"; the stream FORMAT
 * (what blosclz_decompress accepts) is the same in both, so decoding is format-defined.
 *
 * Stream format (FastLZ level-2 lineage):
 *   ctrl < 32           : literal run of ctrl + 1 bytes follows (first ctrl byte of a stream has bit 5 set
 *                         as a marker and is read as ctrl & 31)
 *   ctrl >= 32          : match.  len = (ctrl >> 5) - 1, if that is 6 length bytes follow (sum, each 255
 *                         continues); one distance byte follows; copy length = len + 3;
 *                         distance = ((ctrl & 31) << 8) + byte + 1.  If (ctrl & 31) == 31 and byte == 255 two
 *                         more bytes follow: distance = (b0 << 8) + b1 + 8191 + 1 ("far" match).
 */
#include "orc.h"
#include <string.h>

enum { MAX_COPY = 32, MAX_DISTANCE = 8191, MAX_FARDISTANCE = 65535 + 8191 - 1, HASH_LOG = 12, HASH_LOG2 = 12 };

static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint32_t hashf(uint32_t seq, int hashlog) { return (seq * 2654435761u) >> (32 - hashlog); }

/* get_run / get_match of blosclz.c, both: ip advances over bytes equal to the ones at ref and stops one
 * PAST the first difference, but never beyond ip_bound (the run variant compares with ip[-1]; at distance 1
 * that is the same sequence of comparisons). */
static int run_or_match(const uint8_t* b, int ip, int ip_bound, int ref)
{
    while (ip + 8 < ip_bound) {                    /* eight bytes at a time (same answer as the byte loop below) */
        uint64_t x, y;
        memcpy(&x, b + ip, 8); memcpy(&y, b + ref, 8);
        if (x != y) return ip + (__builtin_ctzll(x ^ y) >> 3) + 1;
        ip += 8; ref += 8;
    }
    while (ip < ip_bound) {
        const int same = b[ref] == b[ip];
        ref++; ip++;
        if (!same) break;
    }
    return ip;
}

/* get_csize(ibase, maxlen, force_3b_shift): dry run over the first maxlen bytes that only counts output
 * bytes.  Quirks kept on purpose: it starts at byte 0 with the counters of "4 literals already copied"
 * (so those bytes are counted twice), its minimum match length is 3 (4 for far matches) whatever the
 * compression level, and it always enters both hashes after a match. */
int orc_blosclz_probe(const uint8_t* b, int maxlen, int force_3b_shift)
{
    uint32_t htab[1 << HASH_LOG2];
    const int ip_bound = maxlen - 1, ip_limit = maxlen - 12;
    const int ipshift = force_3b_shift ? 3 : 4;
    int ip = 0, oc = 5, copy = 4;
    memset(htab, 0, sizeof htab);
    while (ip < ip_limit) {
        const int anchor = ip;
        const uint32_t seq = rd32(b + ip);
        const uint32_t hv = hashf(seq, HASH_LOG2);
        const int ref = (int)htab[hv];
        unsigned distance = (unsigned)(anchor - ref);
        htab[hv] = (uint32_t)anchor;
        int is_match = !(distance == 0 || distance >= MAX_FARDISTANCE) && rd32(b + ref) == seq;
        int len = 0, nip = 0;
        if (is_match) {
            distance--;
            nip = run_or_match(b, anchor + 4, ip_bound, ref + 4) - ipshift;
            len = nip - anchor;
            if (len < (distance >= MAX_DISTANCE ? 4 : 3)) is_match = 0;
        }
        if (!is_match) {
            oc++; ip = anchor + 1; copy++;
            if (copy == MAX_COPY) { copy = 0; oc++; }
            continue;
        }
        if (!copy) oc--;
        copy = 0;
        if (len >= 7) oc += (len - 7) / 255 + 1;
        oc += distance < MAX_DISTANCE ? 2 : 4;
        {
            uint32_t s2 = rd32(b + nip);
            htab[hashf(s2, HASH_LOG2)] = (uint32_t)nip;
            s2 >>= 8;
            htab[hashf(s2, HASH_LOG2)] = (uint32_t)(nip + 1);
            ip = nip + 2;
        }
        oc++;
    }
    if (!copy) oc--;
    return oc;
}

static const uint8_t k_hashlog[10] = {0, HASH_LOG - 2, HASH_LOG - 1, HASH_LOG, HASH_LOG, HASH_LOG, HASH_LOG, HASH_LOG, HASH_LOG, HASH_LOG};
static const int k_minlen[10] = {0, 12, 12, 11, 10, 9, 8, 7, 6, 5};
static const double k_cratio[10] = {0, 2, 2, 2, 2, 1.8, 1.6, 1.4, 1.2, 1.1};

/* entropy probe of blosclz_compress: decides ipshift (clevel 9: whichever shift counts fewer bytes) and
 * whether the stream is worth compressing at all.  Returns 0 = give up, else the ipshift to use. */
int orc_blosclz_plan(int clevel, const uint8_t* b, int length)
{
    int ipshift = 4;
    double cratio = 0;
    if (clevel >= 1 && clevel <= 8) {
        const int maxlen = length / 8;
        cratio = (double)maxlen / (double)orc_blosclz_probe(b, maxlen, 0);
    } else if (clevel == 9) {
        const int maxlen = length / 8;
        const int c3 = orc_blosclz_probe(b, maxlen, 1), c4 = orc_blosclz_probe(b, maxlen, 0);
        ipshift = c3 < c4 ? 3 : 4;
        cratio = (double)maxlen / (double)(c3 < c4 ? c3 : c4);
    }
    if (cratio < k_cratio[clevel < 0 || clevel > 9 ? 0 : clevel]) return 0;
    return ipshift;
}

/* blosclz_compress(clevel, input, length, output, maxout).  Returns the compressed size, 0 when the stream
 * is not worth it / does not fit.  *need (optional): smallest maxout for which this call still succeeds --
 * every budget check of the encoder is "op + k <= op_limit" with a left-hand side that never decreases
 * along the stream, so the last check (op + 2 before the final literal) decides: need = max(66, size + 1). */
int orc_blosclz_compress(int clevel, const uint8_t* b, int length, uint8_t* out, int maxout, int* need)
{
    uint32_t htab[1 << HASH_LOG];
    if (clevel < 0 || clevel > 9) return ORC_ERR_CODEC_PARAM;
    if (length < 16 || maxout < 66) return 0;
    const int hashlog = k_hashlog[clevel];
    const int ipshift = orc_blosclz_plan(clevel, b, length);
    if (!ipshift) return 0;
    const int minlen = clevel == 9 ? ipshift : k_minlen[clevel];
    const int ip_bound = length - 1, ip_limit = length - 12;
    int ip = 4, op = 5, copy = 4;
    memset(htab, 0, sizeof(uint32_t) << hashlog);
    out[0] = MAX_COPY - 1;
    memcpy(out + 1, b, 4);

#define LITERAL_()  do { if (op + 2 > maxout) return 0; out[op++] = b[anchor]; ip = anchor + 1; copy++; \
                         if (copy == MAX_COPY) { copy = 0; out[op++] = MAX_COPY - 1; } } while (0)
    while (ip < ip_limit) {
        const int anchor = ip;
        const uint32_t seq = rd32(b + ip);
        const uint32_t hv = hashf(seq, hashlog);
        const int ref = (int)htab[hv];
        unsigned distance = (unsigned)(anchor - ref);
        htab[hv] = (uint32_t)anchor;
        if (distance == 0 || distance >= MAX_FARDISTANCE || rd32(b + ref) != seq) { LITERAL_(); continue; }
        distance--;
        const int nip = run_or_match(b, anchor + 4, ip_bound, ref + 4) - ipshift;
        unsigned len = (unsigned)(nip - anchor);
        if ((int)len < minlen || (len <= 5 && distance >= MAX_DISTANCE)) { LITERAL_(); continue; }
        if (copy) out[op - copy - 1] = (uint8_t)(copy - 1); else op--;
        copy = 0;
        const int far = distance >= MAX_DISTANCE;
        if (far) distance -= MAX_DISTANCE;
        if (len < 7) {
            if (op + (far ? 4 : 2) > maxout) return 0;
            out[op++] = (uint8_t)((len << 5) + (far ? 31 : (distance >> 8)));
        } else {
            if (op + 1 > maxout) return 0;
            out[op++] = (uint8_t)((7u << 5) + (far ? 31 : (distance >> 8)));
            for (len -= 7; len >= 255; len -= 255) { if (op + 1 > maxout) return 0; out[op++] = 255; }
            if (op + (far ? 4 : 2) > maxout) return 0;
            out[op++] = (uint8_t)len;
        }
        if (far) { out[op++] = 255; out[op++] = (uint8_t)(distance >> 8); }
        out[op++] = (uint8_t)(distance & 255);
        {
            uint32_t s2 = rd32(b + nip);
            htab[hashf(s2, hashlog)] = (uint32_t)nip;
            s2 >>= 8;
            htab[hashf(s2, hashlog)] = (uint32_t)(nip + 1);
            ip = nip + 2;
        }
        if (op + 1 > maxout) return 0;
        out[op++] = MAX_COPY - 1;
    }
    while (ip <= ip_bound) {
        if (op + 2 > maxout) return 0;
        out[op++] = b[ip++];
        copy++;
        if (copy == MAX_COPY) { copy = 0; out[op++] = MAX_COPY - 1; }
    }
#undef LITERAL_
    if (copy) out[op - copy - 1] = (uint8_t)(copy - 1); else op--;
    out[0] |= 1u << 5;
    if (need) *need = op + 1 > 66 ? op + 1 : 66;
    return op;
}

/* blosclz_decompress(input, length, output, maxout): returns the number of bytes produced, 0 on any
 * malformed input.  Kept quirk: when a match is the last thing in the input the decoder stops BEFORE copying
 * it (the encoder never ends a stream with a match: the last 12 bytes are always literals). */
int orc_blosclz_decompress(const uint8_t* in, int length, uint8_t* out, int maxout)
{
    if (length == 0) return 0;
    int ip = 0, op = 0;
    uint32_t ctrl = in[ip++] & 31u;
    for (;;) {
        if (ctrl >= 32) {
            int len = (int)(ctrl >> 5) - 1;
            int ofs = (int)(ctrl & 31u) << 8;
            uint8_t code;
            if (len == 7 - 1) {
                do {
                    if (ip + 1 >= length) return 0;
                    code = in[ip++];
                    len += code;
                } while (code == 255);
            } else if (ip + 1 >= length) return 0;
            code = in[ip++];
            len += 3;
            int64_t ref = (int64_t)op - ofs - code;
            if (code == 255 && ofs == (31 << 8)) {
                if (ip + 1 >= length) return 0;
                ofs = in[ip] << 8; ofs += in[ip + 1]; ip += 2;
                ref = (int64_t)op - ofs - MAX_DISTANCE;
            }
            if (op + len > maxout) return 0;
            if (ref - 1 < 0) return 0;
            if (ip >= length) break;
            ctrl = in[ip++];
            ref--;
            if (op - ref >= len) memcpy(out + op, out + ref, (size_t)len);
            else if (op - ref == 1) memset(out + op, out[ref], (size_t)len);
            else for (int k = 0; k < len; k++) out[op + k] = out[ref + k];
            op += len;
        } else {
            const int n = (int)ctrl + 1;
            if (op + n > maxout) return 0;
            if (ip + n > length) return 0;
            memcpy(out + op, in + ip, (size_t)n);
            op += n; ip += n;
            if (ip >= length) break;
            ctrl = in[ip++];
        }
    }
    return op;
}
