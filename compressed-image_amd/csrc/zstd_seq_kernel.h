// zstd_seq_kernel.h -- the sequences of MANY blocks at once, one lane per block (round 4).
//
// Decoding FSE-coded sequences is a serial walk down a bit stream: as wave-uniform scalar code (zstd_decode.h: zstd_sequences) it
// runs at what a SIMD issues for one wave, about a sequence per 800 cycles and SIMD however many waves share it.  The walk of one
// block is independent of every other block's, though, and what it needs is small -- three tables (5 KiB), a bit position, three
// states, three offsets.  cimg_zstd_seq gives every lane of a wave a block of its own: lane l of workgroup g takes the jobs the
// walker left in the plan of block g * lanes + l (ZstdSeqJob: the block's tables copied to the plan, the bit stream where it lies in
// the chunk), its tables in the lane's 5 KiB of LDS, its bit stream read through a 128-bit window in registers with the next 64
// bits requested one step ahead, and writes the 8-byte records cimg_zstd_replay executes.  A wave decodes `lanes` sequences per
// step instead of one.
#pragma once
#include "zstd_walk_kernel.h"

namespace cimg {

// a lane's LDS: its job's tables, and ZSTD_SEQ_STREAM bytes of the job's bit stream -- the END of it first (a stream is read from its
// last byte down), brought over in one wide copy by the whole wave, and the ZSTD_SEQ_STREAM bytes below the reader again whenever it
// gets to the bottom of what is there (a stream read where it lies costs a trip to global memory per step, and the wave waits for
// every lane's)
#ifndef CIMG_ZSTD_SEQ_STREAM
#define CIMG_ZSTD_SEQ_STREAM 4096      /* (tests build the emulator once more with 256: every stream is refilled many times) */
#endif
enum : int { ZSTD_SEQ_STREAM = CIMG_ZSTD_SEQ_STREAM, ZSTD_SEQ_PAD = 32, ZSTD_SEQ_LANE_BYTES = ZSTD_JOB_TABLE_BYTES + ZSTD_SEQ_PAD + ZSTD_SEQ_STREAM };   // (PAD zero bytes in front of the stream: its window may begin below bit 0)
// (behind the lanes' areas: what the length codes mean, base | extra bits << 24 -- 36 literal length codes, 53 match length codes)
enum : int { ZSTD_SEQ_CODES_BYTES = 4 * (36 + 53 + 7) };
CIMG_HD int zstd_seq_lds_bytes(int lanes) { return lanes * ZSTD_SEQ_LANE_BYTES + ZSTD_SEQ_CODES_BYTES + 64; }

// bits [p, p + 64) of a stream of bl bytes, p a multiple of 8 and any value; bits outside the stream read as zero
CIMG_DEV uint64_t zstd_window64(const uint8_t* bs, int bl, int p)
{
    const int b = p >> 3;
    uint64_t v = 0;
    if (b >= 0 && b + 8 <= bl) { memcpy(&v, bs + b, 8); return v; }
    for (int k = 0; k < 8; k++) { const int i = b + k; if (i >= 0 && i < bl) v |= (uint64_t)bs[i] << (8 * k); }
    return v;
}

struct ZstdSeqLanes {
    const DecodeArgs& a;
    uint8_t* lds;
    int g;
    CIMG_DEV ZstdSeqLanes(const DecodeArgs& a_, uint8_t* lds_, int g_) : a(a_), lds(lds_), g(g_) {}

    CIMG_DEV void run()
    {
        const int lanes = a.zlanes;
        // the meaning of the length codes, once per workgroup: a lookup in the loop instead of a chain of comparisons per lane
        uint32_t* const codes = reinterpret_cast<uint32_t*>(lds + lanes * ZSTD_SEQ_LANE_BYTES);
        FOR_LANES_W(l) {
            if (l < 36) codes[l] = (uint32_t)zstd_ll_base(l) | ((uint32_t)zstd_ll_bits(l) << 24);
            if (l < 53) codes[36 + l] = (uint32_t)zstd_ml_base(l) | ((uint32_t)zstd_ml_bits(l) << 24);
        }
        LV<uint8_t*> slot;                 // the lane's plan
        LV<int> njobs, job_i, rem, err;
        FOR_LANES(l) {
            const int k = g * lanes + l;   // block of the group
            const bool mine = l < lanes && k < a.zblocks;
            slot[l] = a.zplan + (int64_t)(mine ? k : 0) * a.zplan_stride;
            const int32_t* const head = reinterpret_cast<const int32_t*>(slot[l]);
            njobs[l] = (mine && head[0] == ZPLAN_READY) ? head[3] : 0;
            if (njobs[l] < 0 || njobs[l] > ZSTD_PLAN_JOBS) njobs[l] = 0;
            job_i[l] = 0; rem[l] = 0; err[l] = 0;
        }
        // per lane: the job being decoded
        LV<const uint8_t*> bs;
        LV<uint64_t*> rec;
        LV<int> bl, off, lo, sl, so, sm, r0, r1, r2, llm, ofm, mlm;
        LV<uint64_t> c0, c1;               // bits [lo, lo + 64), [lo + 64, lo + 128) of the stream: read anew for every sequence
        LV<int> w0;                        // the lane's LDS holds bytes [w0, w0 + ZSTD_SEQ_STREAM) of its stream (zero bytes in front of byte 0)
        FOR_LANES(l) { bs[l] = nullptr; rec[l] = nullptr; bl[l] = 0; off[l] = 0; lo[l] = 0; sl[l] = so[l] = sm[l] = 0; r0[l] = 1; r1[l] = 4; r2[l] = 8;
                       llm[l] = ofm[l] = mlm[l] = 0; c0[l] = c1[l] = 0; w0[l] = 0; }
        const int cap = (a.zcap + 15) & ~15;
        // bytes [from, from + ZSTD_SEQ_STREAM) of lane t's stream (as many as it has) into the lane's LDS: 16-byte units wide, the last
        // bytes one by one; the zero bytes in front (they are read as the bits below bit 0 when from == 0)
        auto stage_stream = [&](int t, const uint8_t* jbs, int jbl, int from) {
            const int at = t * ZSTD_SEQ_LANE_BYTES + ZSTD_JOB_TABLE_BYTES + ZSTD_SEQ_PAD;
            const int n = imin((int)ZSTD_SEQ_STREAM, jbl - from);
            FOR_LANES_W(l) { if (l < ZSTD_SEQ_PAD) lds[at - ZSTD_SEQ_PAD + l] = 0; }
            wave_copy_g2l(jbs + from, lds, at, n & ~15);
            FOR_LANES_W(l) { if (l < (n & 15)) lds[at + (n & ~15) + l] = jbs[from + (n & ~15) + l]; }
        };
        for (int guard = 0; guard < (1 << 30); ++guard) {
            // ---- lanes between jobs: the next job's tables into the lane's LDS, its reader and states set up
            LV<bool> want;
            FOR_LANES(l) { want[l] = rem[l] == 0 && err[l] == 0 && job_i[l] < njobs[l]; }
            uint64_t need = ballot(want);
            while (need) {
                const int t = ctz64(need);
                need &= need - 1;
                const uint8_t* const sl_t = a.zplan + (int64_t)(g * lanes + t) * a.zplan_stride;          // (a lane that wants a job has a block)
                const int ji = uni(readlane(job_i, t));
                const uint32_t* const jq = reinterpret_cast<const uint32_t*>(sl_t + ZSTD_PLAN_JOBS_AT + 32 * ji);
                const uint32_t tab = uni(jq[5]);
                int bad = 0;
                if (tab > (uint32_t)(ZSTD_PLAN_TABLE_BYTES - ZSTD_JOB_TABLE_BYTES) || (tab & 15)) bad = 1;
                if (!bad) wave_copy_g2l(sl_t + zstd_plan_tabs_at(cap) + tab, lds, t * ZSTD_SEQ_LANE_BYTES, ZSTD_JOB_TABLE_BYTES);
                // the bit stream beside them when it fits: whole 16-byte units wide, the last bytes one by one
                const uint32_t jbl = uni(jq[2]);
                const uint8_t* const jbs = reinterpret_cast<const uint8_t*>((uintptr_t)((uint64_t)uni(jq[0]) | ((uint64_t)uni(jq[1]) << 32)));
                const bool sane = !bad && jbl >= 1 && jbl <= (1u << 24);
                const int from0 = sane ? imax(0, (int)jbl - (int)ZSTD_SEQ_STREAM) : 0;
                if (sane) stage_stream(t, jbs, (int)jbl, from0);
                FOR_LANES(l) {
                    if (l == t) {
                        const ZstdSeqJob* const job = reinterpret_cast<const ZstdSeqJob*>(slot[l] + ZSTD_PLAN_JOBS_AT) + job_i[l];
                        const ZstdSeqJob j = *job;
                        job_i[l] += 1;
                        const int nrecs = zstd_plan_records(cap);
                        if (bad || j.bl < 1 || j.bl > (1u << 24) || j.nseq < 1 || j.nseq > (uint32_t)nrecs || j.rec > (uint32_t)nrecs - j.nseq || j.ll_log > 9 || j.of_log > 8 || j.ml_log > 9) err[l] = ERR_DATA;
                        else {
                            bs[l] = reinterpret_cast<const uint8_t*>((uintptr_t)j.bs);
                            w0[l] = from0;
                            bl[l] = (int)j.bl;
                            rec[l] = reinterpret_cast<uint64_t*>(slot[l] + ZSTD_PLAN_HEAD) + j.rec;
                            rem[l] = (int)j.nseq;
                            llm[l] = (1 << j.ll_log) - 1; ofm[l] = (1 << j.of_log) - 1; mlm[l] = (1 << j.ml_log) - 1;
                            if (j.first) { r0[l] = 1; r1[l] = 4; r2[l] = 8; }
                            const int last = bs[l][bl[l] - 1];
                            if (last == 0) { err[l] = ERR_DATA; rem[l] = 0; }
                            else {
                                off[l] = 8 * bl[l] - (8 - zstd_highbit((uint32_t)last));
                                lo[l] = 8 * bl[l] - 128;
                                c1[l] = zstd_window64(bs[l], bl[l], lo[l] + 64);
                                c0[l] = zstd_window64(bs[l], bl[l], lo[l]);
                                // the three initial states: literal lengths, offsets, match lengths (at most 9 + 8 + 9 bits: inside the window)
                                sl[l] = (int)take(c0[l], c1[l], lo[l], off[l], j.ll_log);
                                so[l] = (int)take(c0[l], c1[l], lo[l], off[l], j.of_log);
                                sm[l] = (int)take(c0[l], c1[l], lo[l], off[l], j.ml_log);
                            }
                        }
                    }
                }
            }
            LV<bool> act;
            FOR_LANES(l) { act[l] = rem[l] > 0 && err[l] == 0; }
            if (!ballot(act)) break;
            // ---- one sequence per lane and step, until a lane is through with its job
            for (;;) {
                // a lane whose reader has got to the bottom of what its LDS holds of the stream: the ZSTD_SEQ_STREAM bytes that end where
                // it stands, by the whole wave (where the stream lies is read from the lane's job again: nothing of it is kept in
                // registers)
                {
                    LV<bool> low;
                    FOR_LANES(l) { low[l] = rem[l] > 0 && err[l] == 0 && w0[l] > 0 && ((off[l] + 7) >> 3) - 16 < w0[l]; }
                    uint64_t refill = ballot(low);
                    while (refill) {
                        const int t = ctz64(refill);
                        refill &= refill - 1;
                        const uint8_t* const sl_t = a.zplan + (int64_t)(g * lanes + t) * a.zplan_stride;
                        const uint32_t* const jq = reinterpret_cast<const uint32_t*>(sl_t + ZSTD_PLAN_JOBS_AT + 32 * (uni(readlane(job_i, t)) - 1));
                        const uint8_t* const jbs = reinterpret_cast<const uint8_t*>((uintptr_t)((uint64_t)uni(jq[0]) | ((uint64_t)uni(jq[1]) << 32)));
                        const int jbl = (int)uni(jq[2]);
                        const int top = (uni(readlane(off, t)) + 7) >> 3;                       // the byte the reader stands in (exclusive end)
                        const int from = imax(0, imin(top, jbl) - (int)ZSTD_SEQ_STREAM);
                        stage_stream(t, jbs, jbl, from);
                        FOR_LANES(l) { if (l == t) w0[l] = from; }
                    }
                }
                LV<bool> through;
                FOR_LANES(l) {
                    through[l] = false;
                    if (rem[l] > 0 && err[l] == 0) {
                        // The window: the 128 bits that end with the byte the reader stands in -- a sequence takes at most 31 + 16 + 16
                        // bits of lengths and offset and 26 bits of states, 89 of the 121 it can count on.  Read anew for every sequence
                        // (two LDS reads beside the three of the table entries, no bookkeeping); below bit 0 lie zero bytes.
                        {
                            const int B = ((off[l] + 7) >> 3) - 16;
                            lo[l] = 8 * B;
                            int rel = B - w0[l];                               // (>= 0 after the refill step above, or w0 == 0: the zero bytes)
                            rel = rel < -ZSTD_SEQ_PAD ? -ZSTD_SEQ_PAD : (rel > ZSTD_SEQ_STREAM - 16 ? ZSTD_SEQ_STREAM - 16 : rel);
                            cimg_lds_cu8p const q = CIMG_AS_LDS_CU8(lds + l * ZSTD_SEQ_LANE_BYTES + ZSTD_JOB_TABLE_BYTES + ZSTD_SEQ_PAD) + rel;
                            uint64_t lo64, hi64;
                            __builtin_memcpy(&lo64, q, 8);
                            __builtin_memcpy(&hi64, q + 8, 8);
                            c0[l] = lo64; c1[l] = hi64;
                        }
                        cimg_lds_cu32p const T = CIMG_AS_LDS_CU32(lds + l * ZSTD_SEQ_LANE_BYTES);
                        const uint32_t pl = T[sl[l] & llm[l]], pm = T[512 + (sm[l] & mlm[l])], po = T[1024 + (so[l] & ofm[l])];
                        const int ls = (int)(pl & 0xFF), lnb = (int)((pl >> 8) & 0xFF), lbase = (int)(pl >> 16);
                        const int os = (int)(po & 0xFF), onb = (int)((po >> 8) & 0xFF), obase = (int)(po >> 16);
                        const int ms = (int)(pm & 0xFF), mnb = (int)((pm >> 8) & 0xFF), mbase = (int)(pm >> 16);
                        if (os > 31 || ls > 35 || ms > 52 || lnb > 9 || onb > 8 || mnb > 9) { err[l] = ERR_DATA; }
                        else {
                            const uint32_t xo = take(c0[l], c1[l], lo[l], off[l], os);
                            // (most sequences of an image have a literal length below 16 and a match length below 35: coded directly)
                            int llen = ls, mlen = ms + 3;
                            if (ls >= 16 || ms >= 32) {
                                cimg_lds_cu32p const C = CIMG_AS_LDS_CU32(lds + lanes * ZSTD_SEQ_LANE_BYTES);
                                const uint32_t lc = C[ls], mc = C[36 + ms];
                                const uint32_t xm = take(c0[l], c1[l], lo[l], off[l], (int)(mc >> 24));
                                const uint32_t xl = take(c0[l], c1[l], lo[l], off[l], (int)(lc >> 24));
                                llen = (int)(lc & 0xFFFFFF) + (int)xl; mlen = (int)(mc & 0xFFFFFF) + (int)xm;
                            }
                            const uint32_t ov = (1u << os) + xo;
                            if (rem[l] > 1) {
                                const uint32_t V = take(c0[l], c1[l], lo[l], off[l], lnb + mnb + onb);
                                so[l] = obase + (int)(V & ((1u << onb) - 1));
                                sm[l] = mbase + (int)((V >> onb) & ((1u << mnb) - 1));
                                sl[l] = lbase + (int)(V >> (onb + mnb));
                            }
                            // the history of repeat offsets (RFC 8878 3.1.1.5) as selects: a new offset pushes the three down; a repeat
                            // code picks one (one further when the sequence has no literals; the fourth choice is the first minus one)
                            // and moves it to the front
                            const bool rep = ov <= 3;
                            const int idx = (int)ov + (llen == 0 ? 1 : 0);
                            const int offset = !rep ? (int)(ov - 3) : idx == 1 ? r0[l] : idx == 2 ? r1[l] : idx == 3 ? r2[l] : r0[l] - 1;
                            r2[l] = (!rep || idx >= 3) ? r1[l] : r2[l];
                            r1[l] = (!rep || idx >= 2) ? r0[l] : r1[l];
                            r0[l] = offset;
                            if (off[l] < 0 || offset <= 0 || offset >= (1 << 22) || llen >= (1 << 21) || mlen >= (1 << 21)) err[l] = ERR_DATA;
                            else {
                                *rec[l] = zstd_record((uint32_t)llen, (uint32_t)mlen, (uint32_t)offset);
                                rec[l] += 1;
                                rem[l] -= 1;
                                if (rem[l] == 0 && off[l] != 0) err[l] = ERR_DATA;       // a stream ends at its first bit
                            }
                        }
                        through[l] = rem[l] == 0 || err[l] != 0;
                    }
                }
                if (ballot(through)) break;
            }
        }
        // a lane that met damage says so in its block's plan (the replay reports it)
        FOR_LANES_W(l) {
            if (err[l] != 0) { int32_t* const head = reinterpret_cast<int32_t*>(slot[l]); head[0] = err[l]; }
        }
    }

    // n bits (n <= 32) below position `off` of the 128-bit window c1:c0 = bits [lo, lo + 128); off moves down.  No branches: a
    // position outside the window (a damaged stream only: the loop's window rule keeps a sound one inside) is pulled into it and the
    // lane reads rubbish -- its stream then ends below bit 0 or above it, which is what the caller checks.
    static CIMG_DEV uint32_t take(uint64_t c0, uint64_t c1, int lo, int& off, int n)
    {
        off -= n;
        int s = off - lo;
        s = s < 0 ? 0 : (s > 127 ? 127 : s);
        const uint64_t a = s < 64 ? c0 : c1, b = s < 64 ? c1 : 0ull;
        const int sh = s & 63;
        const uint64_t v = (a >> sh) | ((b << 1) << (63 - sh));
        return (uint32_t)v & (uint32_t)((1ull << (n < 0 ? 0 : (n > 32 ? 32 : n))) - 1);
    }
};

}  // namespace cimg
