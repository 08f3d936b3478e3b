// zstd_encode.h -- a Zstandard FRAME encoder for one stream, by one wave: the write side of blosc2's codec format 4.
//
// The reference offers enums::codec::zstd (compressed_image/include/compressed/enums.h:18-24, blosc2/wrapper.h:74-119); c-blosc2
// then stores every stream of a block as ONE complete zstd frame (ZSTD_compressCCtx).  libzstd's output at its levels (clevel 9 =
// ZSTD_maxCLevel(), btultra2) cannot be reproduced bit for bit on a GPU and nothing in this image could pin it anyway, so this
// encoder is FORMAT-VALID, NOT BYTE-PINNED: what it writes is a frame every zstd decoder reads (RFC 8878) -- checked in the
// tests by decoding every output with the box's own libzstd and with this repository's cimg_decode_zstd -- and its bytes differ
// from libzstd's by construction (DESIGN.md section 2 says so; the compression ratio is reported beside the rate).
//
// Shape of a frame: magic, frame header (single segment, content size), ONE compressed block:
//     literals section   Raw_Literals_Block: the literal runs of all sequences back to back, then the bytes behind the last match
//     sequences section  the (literal length, offset, match length) triples, FSE-coded with the three PREDEFINED distributions
//                        (no table description in the stream); an offset equal to the previous sequence's is the first repeat offset
// The triples come from the wave's LZ4 match finder (encode_kernel.h: lz4_encode_body<true>, windows of 64 probes, table in LDS)
// at acceleration 1 -- the blosc2 level only decides, as in c-blosc2, whether a block is split into byte planes (clevel <= 5).
// A frame that does not come out smaller than its input is not written: the stream is then stored raw, as c-blosc2 does when
// ZSTD_compress does not fit maxout.
//
// FSE coding walks the sequences from the last to the first (the decoder reads the bit stream backwards), one sequence at a
// time, wave-uniform scalar code (like the decoder in zstd_decode.h: all lanes run it with the same data); the bit stream is
// collected in LDS, in the hash table's place behind the FSE tables (about 15 KiB: a stream with more sequence bits than that is
// not worth a frame) -- NOT in the plane's: the plane is still needed when the frame turns out not to be smaller than its input
// and the stream is stored raw.
#pragma once
#include "codec_types.h"
#include "wave.h"
#include "zstd_decode.h"      // zstd_default_freq, zstd_ll_base / _bits, zstd_ml_base / _bits, zstd_highbit

namespace cimg {

enum : int { ZSTD_ENC_MAX_INPUT = 65536,         // a 64 KiB block as one stream: offsets and match lengths stay below 2^16 (a match starts behind a first byte)
              ZSTD_ENC_HASH_BITS = 12 };         // the match finder's table: 4096 slots of 16 bits (8 KiB; afterwards the place of the FSE tables)

// FSE compression tables of the three predefined distributions (literal lengths: 36 symbols, log 6; offsets: 29, log 5; match
// lengths: 53, log 6), built once on the host the way FSE_buildCTable builds them: state table + per symbol (deltaNbBits,
// deltaFindState).  8-byte aligned, copied into LDS by the kernel.
struct ZstdEncTables {
    uint16_t ll_state[64], ml_state[64], of_state[32];
    int32_t ll_dnb[36], ll_dfs[36], of_dnb[32], of_dfs[32], ml_dnb[56], ml_dfs[56];
};

inline void zstd_build_one_ctable(int which, int nsym, int log, uint16_t* state_table, int32_t* dnb, int32_t* dfs)
{
    const int size = 1 << log, step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    uint8_t table_symbol[64];
    int cumul[64], high = size - 1;
    cumul[0] = 0;
    for (int s = 0; s < nsym; s++) {
        const int f = zstd_default_freq(which, s);
        if (f == -1) { cumul[s + 1] = cumul[s] + 1; table_symbol[high--] = (uint8_t)s; }
        else cumul[s + 1] = cumul[s] + f;
    }
    int pos = 0;
    for (int s = 0; s < nsym; s++) {
        const int f = zstd_default_freq(which, s);
        for (int i = 0; i < f; i++) {
            table_symbol[pos] = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    }
    int next[64];
    for (int s = 0; s < nsym; s++) next[s] = cumul[s];
    for (int u = 0; u < size; u++) state_table[next[table_symbol[u]]++] = (uint16_t)(size + u);
    int total = 0;
    for (int s = 0; s < nsym; s++) {
        const int f = zstd_default_freq(which, s);
        if (f == 0) { dnb[s] = ((log + 1) << 16) - size; dfs[s] = 0; }
        else if (f == -1 || f == 1) { dnb[s] = (log << 16) - size; dfs[s] = total - 1; total++; }
        else {
            const int max_bits = log - zstd_highbit((uint32_t)(f - 1));
            dnb[s] = (max_bits << 16) - (f << max_bits);
            dfs[s] = total - f;
            total += f;
        }
    }
}

inline void zstd_build_enc_tables(ZstdEncTables* t)
{
    memset(t, 0, sizeof(*t));
    zstd_build_one_ctable(0, 36, 6, t->ll_state, t->ll_dnb, t->ll_dfs);
    zstd_build_one_ctable(1, 29, 5, t->of_state, t->of_dnb, t->of_dfs);
    zstd_build_one_ctable(2, 53, 6, t->ml_state, t->ml_dnb, t->ml_dfs);
}

CIMG_DEV int zstd_ll_code(int ll) { return ll < 16 ? ll : ll < 24 ? 16 + ((ll - 16) >> 1) : ll < 32 ? 20 + ((ll - 24) >> 2) : ll < 48 ? 22 + ((ll - 32) >> 3) : ll < 64 ? 24 : zstd_highbit((uint32_t)ll) + 19; }
CIMG_DEV int zstd_ml_code(int ml)     // ml = the match length itself (>= 3)
{
    if (ml < 35) return ml - 3;
    if (ml < 43) return 32 + ((ml - 35) >> 1);
    if (ml < 51) return 36 + ((ml - 43) >> 2);
    if (ml < 67) return 38 + ((ml - 51) >> 3);
    if (ml < 99) return 40 + ((ml - 67) >> 4);
    if (ml < 131) return 42;
    return zstd_highbit((uint32_t)(ml - 3)) + 36;
}

// bytes of the frame in front of the literals: magic (4), frame header descriptor + content size, block header (3), literals
// section header (3: raw literals, 20-bit size)
CIMG_HD int zstd_frame_prefix(int n) { return 4 + 1 + (n < 256 ? 1 : n <= 65791 ? 2 : 4) + 3 + 3; }

// the sink the LZ4 match finder writes to instead of LZ4 bytes (encode_kernel.h: lz4_encode_body<true>)
struct SeqSink {
    uint32_t* seq;        // two dwords per sequence: literal length; (match length << 16) | offset
    int nseq;
    int lit_total;        // literal bytes written to the literal area (incl. the bytes behind the last match)
};

// Writes the np parked sequences (lane k = k-th) as records and appends their literal runs to the literal area at lit_out[lit_pos..).
// false: the literals do not fit lit_cap (the stream is not worth a frame).
CIMG_DEV bool zstd_take_parked(const uint8_t* in, cimg_global_u8p lit_out, int lit_cap, int& lit_pos, SeqSink& sink, int np,
                               const LV<int>& P_anchor, const LV<int>& P_lit, const LV<int>& P_off, const LV<int>& P_mcode)
{
    LV<int> size, start;
    FOR_LANES(l) { size[l] = l < np ? P_lit[l] : 0; }
    int total;
    wave_exscan(size, start, total);
    if (lit_pos + total > lit_cap) return false;
    uint32_t* rec = sink.seq + 2 * (size_t)sink.nseq;
    FOR_LANES(l) {
        if (l < np) {
            rec[2 * l] = (uint32_t)P_lit[l];
            rec[2 * l + 1] = ((uint32_t)(P_mcode[l] + 4) << 16) | (uint32_t)P_off[l];
        }
    }
    LV<bool> shortlit, longlit;
    FOR_LANES(l) { shortlit[l] = l < np && P_lit[l] > 0 && P_lit[l] <= 8; longlit[l] = l < np && P_lit[l] > 8; }
    if (ballot(shortlit)) {
        for (int r = 0; r < 8; ++r) {                              // short literal runs: one lane per sequence
            LV<bool> more;
            FOR_LANES(l) {
                if (shortlit[l] && r < P_lit[l]) lit_out[lit_pos + start[l] + r] = in[P_anchor[l] + r];
                more[l] = shortlit[l] && r + 1 < P_lit[l];
            }
            if (!ballot(more)) break;
        }
    }
    uint64_t longm = ballot(longlit);
    while (longm) {                                                // long literal runs: the whole wave copies one
        const int k = ctz64(longm);
        longm &= longm - 1;
        const int from = readlane(P_anchor, k), to = lit_pos + readlane(start, k), count = readlane(P_lit, k);
#ifndef CIMG_EMULATE
#pragma unroll 1
#endif
        for (int c = 0; c < count; c += 64) { FOR_LANES(l) { if (c + l < count) lit_out[to + c + l] = in[from + c + l]; } }
    }
    lit_pos += total;
    sink.nseq += np;
    return true;
}

// the bit stream under construction: bits are appended at the low end of a 64-bit container and leave it bytewise, straight to
// their place in the frame (stores nobody waits for; the plane in LDS stays intact for the raw store of a frame that does not pay)
struct ZstdBitWriter {
    cimg_global_u8p frame;
    int pos, limit;          // next byte of the frame, first byte not to be written
    uint64_t cont;
    int nbits;
    bool overflow;
    CIMG_DEV void init(cimg_global_u8p f, int begin, int end) { frame = f; pos = begin; limit = end; cont = 0; nbits = 0; overflow = false; }
    CIMG_DEV void add(uint32_t value, int n) { cont |= (uint64_t)(value & (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u))) << nbits; nbits += n; }
    // whole bytes of the container -> frame (at most 7 bits stay); call before the container could pass 64 bits
    CIMG_DEV void flush()
    {
        const int nb = nbits >> 3;
        if (pos + 8 > limit) { overflow = true; return; }
        const uint64_t c = cont;
        FOR_LANES_W(l) { if (l < nb) frame[pos + l] = (uint8_t)(c >> (8 * (l & 7))); }
        pos += nb;
        cont = nb >= 8 ? 0 : (cont >> (8 * nb));
        nbits &= 7;
    }
};

CIMG_DEV void zstd_fse_put(ZstdBitWriter& bw, uint32_t& state, const uint16_t* state_table, int dnb, int dfs)
{
    const uint32_t nb = (uint32_t)((int)state + dnb) >> 16;
    bw.add(state, (int)nb);
    state = (uint32_t)uni((int)state_table[(int)(state >> nb) + dfs]);
}
CIMG_DEV uint32_t zstd_fse_init(const uint16_t* state_table, int dnb, int dfs)
{
    const uint32_t nb = (uint32_t)(dnb + (1 << 15)) >> 16;
    const uint32_t value = (nb << 16) - (uint32_t)dnb;
    return (uint32_t)uni((int)state_table[(int)(value >> nb) + dfs]);
}

// The frame around what the match finder left: lds = the wave's LDS (plane at 0, n bytes, dead by now; table area at tab_off,
// dead too: the FSE tables' place), out = the stream's output (cap bytes; the literal area at out + zstd_frame_prefix(n) is filled already),
// sink = the sequences.  Returns the frame size, 0 when it does not fit cap (or is not smaller than n).
CIMG_DEV int zstd_finish_frame(uint8_t* lds, int tab_off, int n, cimg_global_u8p out, int cap, const SeqSink& sink, const ZstdEncTables* tabs_global)
{
    const int nseq = sink.nseq, lit_total = sink.lit_total;
    if (nseq <= 0) return 0;
    // ---- tables -> LDS (the hash table's place) --------------------------------------------------------------------------
    ZstdEncTables* T = reinterpret_cast<ZstdEncTables*>(lds + tab_off);
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(tabs_global);
        uint32_t* dst = reinterpret_cast<uint32_t*>(T);
        const int words = (int)(sizeof(ZstdEncTables) / 4);
        for (int w0 = 0; w0 < words; w0 += 64) { FOR_LANES(l) { if (w0 + l < words) dst[w0 + l] = src[w0 + l]; } }
    }
    // ---- sequences section header ------------------------------------------------------------------------------------------
    const int prefix = zstd_frame_prefix(n);
    const int seq_at = prefix + lit_total;                       // offset of the sequences section in the frame
    const int nseq_bytes = nseq < 128 ? 1 : nseq < 0x7F00 ? 2 : 3;
    const int stream_at = seq_at + nseq_bytes + 1;               // the FSE bit stream (behind the symbol-compression-modes byte)
    if (stream_at + 8 >= cap) return 0;
    ZstdBitWriter bw;
    bw.init(out, stream_at, cap);
    // ---- FSE: the last sequence first ---------------------------------------------------------------------------------------
    uint32_t st_ll = 0, st_of = 0, st_ml = 0;
#ifdef CIMG_ABL_ZSTD_NO_FSE      /* timing experiment only: the sequences are not coded */
    for (int base = -1; base >= 0; base -= 64) {
#else
    for (int base = ((nseq - 1) >> 6) << 6; base >= 0; base -= 64) {
#endif          // batches of 64 records, highest first
        // (rp: the record BEFORE each one -- a sequence whose offset repeats the previous sequence's, and that has literals, is
        // coded as the first repeat offset, Offset_Value 1: no extra bits.  The decoder's first repeat offset always IS the previous
        // sequence's offset here -- a real offset becomes it, a repeat leaves it -- and starts at 1 for the first sequence; the
        // second and third repeat offsets are never used, so their bookkeeping cannot matter.  A sequence WITHOUT literals reads
        // Offset_Value 1 as the second repeat offset: it keeps its real offset.)
        LV<uint32_t> r0, r1, rp;
        FOR_LANES(l) {
            const int i = imin(base + l, nseq - 1);
            r0[l] = sink.seq[2 * (size_t)i];
            r1[l] = sink.seq[2 * (size_t)i + 1];
            rp[l] = i > 0 ? (sink.seq[2 * (size_t)i - 1] & 0xFFFFu) : 1u;
        }
        const int top = imin(63, nseq - 1 - base);
        for (int k = top; k >= 0; --k) {
            const uint32_t w0 = readlane(r0, k), w1 = readlane(r1, k);
            const int ll = (int)w0, ml = (int)(w1 >> 16);
            const int off_base = ((w1 & 0xFFFF) == readlane(rp, k) && ll > 0) ? 1 : (int)(w1 & 0xFFFF) + 3;
            const int llc = zstd_ll_code(ll), mlc = zstd_ml_code(ml), ofc = zstd_highbit((uint32_t)off_base);
            const int llb = zstd_ll_bits(llc), mlb = zstd_ml_bits(mlc);
            if (base + k == nseq - 1) {
                st_ml = zstd_fse_init(T->ml_state, uni(T->ml_dnb[mlc]), uni(T->ml_dfs[mlc]));
                st_of = zstd_fse_init(T->of_state, uni(T->of_dnb[ofc]), uni(T->of_dfs[ofc]));
                st_ll = zstd_fse_init(T->ll_state, uni(T->ll_dnb[llc]), uni(T->ll_dfs[llc]));
            } else {
                zstd_fse_put(bw, st_of, T->of_state, uni(T->of_dnb[ofc]), uni(T->of_dfs[ofc]));
                zstd_fse_put(bw, st_ml, T->ml_state, uni(T->ml_dnb[mlc]), uni(T->ml_dfs[mlc]));
                zstd_fse_put(bw, st_ll, T->ll_state, uni(T->ll_dnb[llc]), uni(T->ll_dfs[llc]));
                bw.flush();                                       // <= 7 + 5 + 6 + 6 bits so far
            }
            bw.add((uint32_t)(ll - zstd_ll_base(llc)), llb);
            bw.add((uint32_t)(ml - zstd_ml_base(mlc)), mlb);
            bw.flush();                                           // <= 7 + 16 + 16
            bw.add((uint32_t)off_base - (1u << ofc), ofc);
            bw.flush();
            if (bw.overflow) return 0;
        }
    }
    bw.add(st_ml, 6);
    bw.add(st_of, 5);
    bw.add(st_ll, 6);
    bw.add(1, 1);                                                 // the end mark the decoder looks for
    bw.flush();
    if (bw.overflow) return 0;
    int stream_bytes = bw.pos - stream_at;
    if (bw.nbits > 0) {
        const uint64_t c = bw.cont;
        FOR_LANES_W(l) { if (l == 0) out[stream_at + stream_bytes] = (uint8_t)c; }       // (flush kept eight bytes of room)
        stream_bytes += 1;
    }
    const int total = stream_at + stream_bytes;
    if (total > cap || total >= n) return 0;
    // ---- the fixed parts ---------------------------------------------------------------------------------------------------------
    {
        const int fcs_bytes = n < 256 ? 1 : n <= 65791 ? 2 : 4;
        const int fcs_flag = n < 256 ? 0 : n <= 65791 ? 1 : 2;
        const uint32_t fcs = n < 256 ? (uint32_t)n : n <= 65791 ? (uint32_t)(n - 256) : (uint32_t)n;
        const int bh_at = 5 + fcs_bytes;
        const uint32_t bsize = (uint32_t)(total - bh_at - 3);                    // Block_Content size
        const uint32_t bh = 1u | (2u << 1) | (bsize << 3);                       // last block, Compressed_Block
        const uint32_t lh = 0u | (3u << 2) | ((uint32_t)lit_total << 4);         // Raw_Literals_Block, 20-bit size
        FOR_LANES(l) {
            int v = -1;
            if (l == 0) v = 0x28; else if (l == 1) v = 0xB5; else if (l == 2) v = 0x2F; else if (l == 3) v = 0xFD;
            else if (l == 4) v = (fcs_flag << 6) | (1 << 5);                     // Single_Segment, no checksum, no dictionary
            else if (l >= 5 && l < 5 + fcs_bytes) v = (int)((fcs >> (8 * (l - 5))) & 0xFF);
            else if (l >= bh_at && l < bh_at + 3) v = (int)((bh >> (8 * (l - bh_at))) & 0xFF);
            else if (l >= bh_at + 3 && l < bh_at + 6) v = (int)((lh >> (8 * (l - bh_at - 3))) & 0xFF);
            if (v >= 0) out[l] = (uint8_t)v;
        }
        FOR_LANES(l) {
            int v = -1;
            if (nseq_bytes == 1) { if (l == 0) v = nseq; }
            else if (nseq_bytes == 2) { if (l == 0) v = (nseq >> 8) + 0x80; else if (l == 1) v = nseq & 0xFF; }
            else { if (l == 0) v = 0xFF; else if (l == 1) v = (nseq - 0x7F00) & 0xFF; else if (l == 2) v = ((nseq - 0x7F00) >> 8) & 0xFF; }
            if (l == nseq_bytes) v = 0;                                          // symbol compression modes: all three predefined
            if (v >= 0) out[seq_at + l] = (uint8_t)v;
        }
    }
    return total;
}

}  // namespace cimg
