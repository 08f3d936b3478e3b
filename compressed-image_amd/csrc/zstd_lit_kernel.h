// zstd_lit_kernel.h -- the Huffman-coded literals of MANY blocks at once, one lane per stream (round 4).
//
// A literals section is one or four Huffman streams, each a serial walk (a table entry per symbol, the entry says how far to move).
// Inside the walker (zstd_decode.h: zstd_huf4_run, four lanes of a wave busy) it costs about 1500 cycles a symbol and is 3/4 of what
// a photograph's chunk takes to read.  cimg_zstd_lit gives lanes 4k .. 4k+3 of a wave the (up to) four streams of block k's job --
// sixteen blocks a wave, the blocks' decoding tables (4 KiB each, copied to the plan by the walker) in LDS, the streams read where
// they lie in the chunk, eight bytes per five symbols (a code has at most 11 bits: 55 of the 57 the window can count on), the
// symbols stored as bytes in the plan's literal area, where cimg_zstd_replay picks them up.
#pragma once
#include "zstd_seq_kernel.h"

namespace cimg {

enum : int { ZSTD_LIT_BLOCKS = 8 };
CIMG_HD int zstd_lit_lds_bytes() { return ZSTD_LIT_BLOCKS * ZSTD_HUF_TABLE_BYTES + 64; }

struct ZstdLitLanes {
    const DecodeArgs& a;
    uint8_t* lds;
    int g;
    CIMG_DEV ZstdLitLanes(const DecodeArgs& a_, uint8_t* lds_, int g_) : a(a_), lds(lds_), g(g_) {}

    CIMG_DEV void run()
    {
        const int cap = (a.zcap + 15) & ~15;
        LV<uint8_t*> slot;
        LV<int> njobs, job_i, rem, err;                      // per lane; the four lanes of a block move through its jobs together
        FOR_LANES(l) {
            const int k = g * ZSTD_LIT_BLOCKS + (l >> 2);
            const bool mine = (l >> 2) < ZSTD_LIT_BLOCKS && k < a.zblocks;
            slot[l] = a.zplan + (int64_t)(mine ? k : 0) * a.zplan_stride;
            const int32_t* const head = reinterpret_cast<const int32_t*>(slot[l]);
            njobs[l] = (mine && head[0] == ZPLAN_READY) ? *reinterpret_cast<const int32_t*>(slot[l] + ZSTD_PLAN_NLIT_AT) : 0;
            if (njobs[l] < 0 || njobs[l] > ZSTD_PLAN_LITJOBS) njobs[l] = 0;
            job_i[l] = 0; rem[l] = 0; err[l] = 0;
        }
        LV<const uint8_t*> bs;
        LV<uint8_t*> out;
        LV<int> size, top, log;
        FOR_LANES(l) { bs[l] = nullptr; out[l] = nullptr; size[l] = 0; top[l] = 0; log[l] = 1; }
        for (int guard = 0; guard < (1 << 30); ++guard) {
            // ---- blocks between jobs (all four lanes of a block are through): the next job's table into the block's LDS
            LV<bool> idle, want;
            FOR_LANES(l) { idle[l] = rem[l] == 0; }
            const uint64_t idle_m = ballot(idle);
            FOR_LANES(l) {
                const uint64_t four = (idle_m >> (l & ~3)) & 15;
                want[l] = (l & 3) == 0 && four == 15 && job_i[l] < njobs[l];
            }
            uint64_t need = ballot(want);
            while (need) {
                const int t = ctz64(need);                     // lane 4k: block k of the group
                need &= need - 1;
                const uint8_t* const sl_t = a.zplan + (int64_t)(g * ZSTD_LIT_BLOCKS + (t >> 2)) * a.zplan_stride;
                const int ji = uni(readlane(job_i, t));
                const uint32_t* const jq = reinterpret_cast<const uint32_t*>(sl_t + ZSTD_PLAN_LITJOBS_AT + 64 * ji);
                const uint32_t tab = uni(jq[5]), jlog = uni(jq[6]);
                const bool bad = tab > (uint32_t)(ZSTD_PLAN_TABLE_BYTES - ZSTD_HUF_TABLE_BYTES) || (tab & 15) || jlog < 1 || jlog > (uint32_t)ZSTD_HUF_LOG_MAX;
                if (!bad) wave_copy_g2l(sl_t + zstd_plan_tabs_at(cap) + tab, lds, (t >> 2) * ZSTD_HUF_TABLE_BYTES, (2 << jlog) < 16 ? 16 : (2 << jlog));
                FOR_LANES(l) {
                    if ((l & ~3) == t) {
                        const ZstdLitJob* const job = reinterpret_cast<const ZstdLitJob*>(slot[l] + ZSTD_PLAN_LITJOBS_AT) + job_i[l];
                        const uint64_t jsrc = job->src;
                        const int lsz = (int)job->lsz, regen = (int)job->regen, jout = (int)job->out, streams = (int)job->streams;
                        job_i[l] += 1;
                        const int j = l & 3;
                        const uint8_t* const p = reinterpret_cast<const uint8_t*>((uintptr_t)jsrc);
                        int e = 0;
                        bs[l] = nullptr; size[l] = 0; rem[l] = 0;
                        if (bad || lsz < 1 || lsz > (1 << 24) || regen < 0 || jout < 0 || regen > cap - jout || (streams != 1 && streams != 4)) e = ERR_DATA;
                        else if (streams == 1) {
                            if (j == 0) { bs[l] = p; size[l] = lsz; rem[l] = regen; out[l] = slot[l] + zstd_plan_lits_at(cap) + jout; }
                        } else if (lsz < 6) e = ERR_DATA;
                        else {
                            const int s1 = p[0] | (p[1] << 8), s2 = p[2] | (p[3] << 8), s3 = p[4] | (p[5] << 8);
                            const int s4 = lsz - 6 - s1 - s2 - s3;
                            const int per = (regen + 3) / 4;
                            if (s4 < 1 || s1 < 1 || s2 < 1 || s3 < 1 || 3 * per > regen) e = ERR_DATA;
                            else {
                                bs[l] = p + 6 + (j > 0 ? s1 : 0) + (j > 1 ? s2 : 0) + (j > 2 ? s3 : 0);
                                size[l] = j == 0 ? s1 : j == 1 ? s2 : j == 2 ? s3 : s4;
                                rem[l] = j == 3 ? regen - 3 * per : per;
                                out[l] = slot[l] + zstd_plan_lits_at(cap) + jout + j * per;
                            }
                        }
                        if (e == 0 && rem[l] > 0) {
                            const int last = bs[l][size[l] - 1];
                            if (last == 0) e = ERR_DATA;
                            else top[l] = 8 * size[l] - (8 - zstd_highbit((uint32_t)last));
                            log[l] = (int)job->log;
                        } else if (e == 0 && size[l] > 0 && bs[l] != nullptr && rem[l] == 0) {
                            // (a stream that regenerates nothing must be the one bit that ends it)
                            if (bs[l][size[l] - 1] != 1 || size[l] != 1) e = ERR_DATA;
                        }
                        if (e) { err[l] = e; rem[l] = 0; }
                    }
                }
            }
            LV<bool> act;
            FOR_LANES(l) { act[l] = rem[l] > 0; }
            if (!ballot(act)) break;
            // ---- up to five symbols per lane and step, until the four lanes of some block are through with their job
            for (;;) {
                LV<bool> now_idle;
                FOR_LANES(l) {
                    if (rem[l] > 0) {
                        // The 128 bits that end with the byte the reader stands in, in ONE read -- ten symbols a step: five from the
                        // upper 64 bits (at least 57 of them below the reader, a code has at most 11), then the 64 bits that end with
                        // the byte the reader has reached by then (cut out of the same 128), five more.
                        const int B = ((top[l] + 7) >> 3) - 16;
                        uint64_t c0, c1;
                        if (B >= 0) { __builtin_memcpy(&c0, bs[l] + B, 8); __builtin_memcpy(&c1, bs[l] + B + 8, 8); }
                        else { c0 = zstd_window64(bs[l], size[l], 8 * B); c1 = zstd_window64(bs[l], size[l], 8 * B + 64); }
                        cimg_lds_cu16p const T = CIMG_AS_LDS_CU16(lds + (l >> 2) * ZSTD_HUF_TABLE_BYTES);
                        const int lg = log[l];
                        const uint32_t mask = (1u << lg) - 1;
                        int t = top[l], n = rem[l];
                        uint8_t* o = out[l];
                        uint64_t c = c1;
                        int lo = 8 * B + 64;
                        CIMG_UNROLL
                        for (int half = 0; half < 2; half++) {
                            CIMG_UNROLL
                            for (int k = 0; k < 5; k++) {
                                if (n > 0) {
                                    const int s = t - lg - lo;                 // (>= 0: lo <= t - 57 at the top of the half, at most 44 bits taken since)
                                    const uint32_t idx = (s >= 0 ? (uint32_t)(c >> s) : (uint32_t)(c << -s)) & mask;
                                    const uint32_t e = T[idx];
                                    *o++ = (uint8_t)e;
                                    t -= (int)(e >> 8);
                                    n -= 1;
                                }
                            }
                            if (half == 0) {
                                int d = ((t + 7) >> 3) - 8 - B;                // bytes the second window lies above the start of the 128 bits: 1 .. 8
                                d = d < 0 ? 0 : (d > 8 ? 8 : d);               // (a damaged stream only)
                                c = d >= 8 ? c1 : (d <= 0 ? c0 : (c0 >> (8 * d)) | (c1 << (64 - 8 * d)));
                                lo = 8 * (B + d);
                            }
                        }
                        top[l] = t; rem[l] = n; out[l] = o;
                        if (t < 0 || (n == 0 && t != 0)) { err[l] = ERR_DATA; rem[l] = 0; }      // a stream ends at its first bit, with its last symbol
                    }
                    now_idle[l] = rem[l] == 0;
                }
                const uint64_t m = ballot(now_idle);
                // (a block whose four lanes are idle and that has another job: back to the loading step)
                LV<bool> reload;
                FOR_LANES(l) { reload[l] = (l & 3) == 0 && ((m >> (l & ~3)) & 15) == 15 && job_i[l] < njobs[l]; }
                if (ballot(reload) || m == ~0ull) break;
            }
        }
        FOR_LANES_W(l) {
            if (err[l] != 0) { int32_t* const head = reinterpret_cast<int32_t*>(slot[l]); head[0] = err[l]; }
        }
    }
};

}  // namespace cimg
