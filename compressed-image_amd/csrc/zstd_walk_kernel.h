// zstd_walk_kernel.h -- the zstd read path as TWO launches (round 4).
//
// cimg_decode_zstd (zstd_kernel.h) keeps a block's output, the frame stage and the entropy tables in LDS together: 52.5 KiB, three
// waves a CU -- and the entropy decoder, a serial scalar walk down a bit stream, runs at what ONE wave per SIMD issues.  The walk
// does not need the output.  So:
//   cimg_zstd_walk     one wave per block, LDS = stage + tables = 20 KiB: EIGHT waves a CU.  Reads what the block's frames say into
//                      the block's PLAN in global memory (zstd_decode.h: ZstdOp, 8-byte sequence records, the Huffman-coded
//                      literals decoded; raw literals and raw blocks stay where they lie in the chunk)
//   cimg_zstd_replay   one wave per block, LDS = the block's planes: four waves a CU.  Replays the plan -- literals to the end of
//                      the plane, records executed 64 at a time by the batch executor -- then the filter stage and the stores
// A plan that does not fit its slot (a block with more than area / 8 sequences, more than 62 ops, more coded literals than the
// block is long) is marked, counted, and the block is decoded by cimg_decode_zstd behind the two launches.
#pragma once
#include "zstd_kernel.h"

namespace cimg {

enum : int { ZSTD_PLAN_OPS = 62, ZSTD_PLAN_NLIT_AT = 2000, ZSTD_PLAN_JOBS_AT = 2048, ZSTD_PLAN_JOBS = 32, ZSTD_PLAN_LITJOBS_AT = 3072, ZSTD_PLAN_LITJOBS = 16, ZSTD_PLAN_HEAD = 4096,
             ZSTD_PLAN_TABLES = 4, ZSTD_PLAN_TABLE_BYTES = ZSTD_PLAN_TABLES * (ZSTD_JOB_TABLE_BYTES + ZSTD_HUF_TABLE_BYTES) };   // (tables of four compressed blocks: FSE and Huffman)
enum : int { ZFALL_PENDING = 1, ZFALL_DONE = 2 };        // DecodeArgs::skipped[1 + block] while the read path runs: the replay left the block to cimg_decode_zstd / that kernel took it
enum : int { ZPLAN_NOT_OURS = 0, ZPLAN_READY = 1, ZPLAN_FALLBACK = 2 };          // a plan's first word; negative: the block's error code
static_assert(16 + ZSTD_PLAN_OPS * sizeof(ZstdOp) <= ZSTD_PLAN_NLIT_AT && ZSTD_PLAN_JOBS_AT + ZSTD_PLAN_JOBS * sizeof(ZstdSeqJob) <= ZSTD_PLAN_LITJOBS_AT &&
              ZSTD_PLAN_LITJOBS_AT + ZSTD_PLAN_LITJOBS * sizeof(ZstdLitJob) <= ZSTD_PLAN_HEAD, "status, counts, ops and jobs in the head of a slot");
// a slot: head (status, ops, records, jobs | the ops | the jobs), 2 x `cap` bytes of records, `cap` bytes of literals, and -- jobs for the
// lane decoders -- room for the tables of ZSTD_PLAN_TABLES compressed blocks
// (cap = the 16-byte multiple above the block area.  Records: 2 x cap bytes = a sequence per four bytes of the block -- what byte-wide
// photographs at level 22 come to; denser blocks are refused.  Literals: cap bytes.)
CIMG_HD int zstd_plan_records(int cap) { return (2 * cap) >> 3; }
CIMG_HD int64_t zstd_plan_lits_at(int cap) { return ZSTD_PLAN_HEAD + 2 * (int64_t)cap; }
CIMG_HD int64_t zstd_plan_tabs_at(int cap) { return ZSTD_PLAN_HEAD + 3 * (int64_t)cap; }
CIMG_HD int64_t zstd_plan_stride(int cap, bool jobs) { return zstd_plan_tabs_at((cap + 15) & ~15) + (jobs ? ZSTD_PLAN_TABLE_BYTES : 0); }
CIMG_HD int zstd_walk_lds_bytes(int stage = ZSTD_KERNEL_STAGE) { return ((stage + 15) & ~15) + zstd_work_bytes() + 64; }
CIMG_HD int zstd_replay_lds_bytes(int max_blocksize) { return zstd_kernel_area(max_blocksize) + 64; }

struct ZstdWalkBlock {
    const DecodeArgs& a;
    uint8_t* lds;
    int b;
    CIMG_DEV ZstdWalkBlock(const DecodeArgs& a_, uint8_t* lds_, int b_) : a(a_), lds(lds_), b(b_) {}

    CIMG_DEV void say(int32_t* head, int status, int nops, int nrecs, int njobs = 0, int nlit = 0)
    {
        FOR_LANES_W(l) { if (l == 0) { head[1] = nops; head[2] = nrecs; head[3] = njobs; head[ZSTD_PLAN_NLIT_AT / 4] = nlit; head[0] = status; } }
    }
    CIMG_DEV void run()
    {
        uint8_t* const slot = a.zplan + (int64_t)(b - a.blk_first) * a.zplan_stride;
        int32_t* const head = reinterpret_cast<int32_t*>(slot);
        ZstdBlockGeom g;
        const int ours = g.parse(a, b, a.zarea, 0);
        if (ours == 0) { say(head, ZPLAN_NOT_OURS, 0, 0); return; }
        if (ours < 0) { say(head, ours, 0, 0); return; }
        const int cap = (a.zcap + 15) & ~15;
        uint8_t* const stage = lds;
        const int stage_cap = (a.lds_bytes - zstd_work_bytes() - 64) & ~15;       // zstd_walk_lds_bytes, read backwards
        ZstdWork* w = reinterpret_cast<ZstdWork*>(lds + stage_cap);
        FOR_LANES_W(l) {
            w->stage = stage; w->stage_cap = stage_cap; w->tail = 1;
            w->mem_lo = lds; w->mem_hi = lds + (a.lds_bytes & ~3);                    // ("this is the kernel": tables and stage are LDS)
            w->ops = reinterpret_cast<ZstdOp*>(slot + 16); w->op_cap = ZSTD_PLAN_OPS; w->op_n = 0;
            w->recs = reinterpret_cast<uint64_t*>(slot + ZSTD_PLAN_HEAD); w->rec_cap = zstd_plan_records(cap); w->rec_n = 0;
            w->lits = slot + zstd_plan_lits_at(cap); w->lit_cap = cap; w->lit_n = 0;
            w->stream = 0;
            w->defer = a.zlanes > 0 ? 1 : 0;
            w->jobs = reinterpret_cast<ZstdSeqJob*>(slot + ZSTD_PLAN_JOBS_AT); w->job_cap = ZSTD_PLAN_JOBS; w->job_n = 0;
            w->litjobs = reinterpret_cast<ZstdLitJob*>(slot + ZSTD_PLAN_LITJOBS_AT); w->litjob_cap = ZSTD_PLAN_LITJOBS; w->litjob_n = 0; w->huf_tab = 0;
            w->tabs = slot + zstd_plan_tabs_at(cap); w->tab_cap = a.zlanes > 0 ? ZSTD_PLAN_TABLE_BYTES : 0; w->tab_n = 0; w->frame_jobs = 0;
        }
        int pos = g.bstart;
        for (int s = 0; s < g.ns; s++) {
            int cs, payload;
            const int hrc = g.stream_header(pos, cs, payload);
            if (hrc < 0) { say(head, hrc, 0, 0); return; }
            if (cs > 0 && cs < g.neblock) {
                FOR_LANES_W(l) { w->stream = s; }
                const int r = zstd_decode_frame(g.c + pos, cs, nullptr, g.neblock, w);
                if (r == ZSTD_WALK_OVERFLOW) {
                    say(head, ZPLAN_FALLBACK, 0, 0);
                    FOR_LANES_W(l) { if (l == 0) atomic_count(a.skipped); }
                    return;
                }
#if !defined(CIMG_ABL_ZSTD_NO_SEQ) && !defined(CIMG_ABL_ZSTD_NO_EXEC)
                if (r != g.neblock) { say(head, r < 0 ? r : ERR_DATA, 0, 0); return; }
#endif
            } else if (cs > g.neblock) { say(head, ERR_DATA, 0, 0); return; }
            pos += payload;                                      // (run tokens and stored streams: the replay reads them in the chunk)
        }
        say(head, ZPLAN_READY, zstd_field(&w->op_n), zstd_field(&w->rec_n), zstd_field(&w->job_n), zstd_field(&w->litjob_n));
    }
};

struct ZstdReplayBlock {
    const DecodeArgs& a;
    uint8_t* lds;
    int b;
    CIMG_DEV ZstdReplayBlock(const DecodeArgs& a_, uint8_t* lds_, int b_) : a(a_), lds(lds_), b(b_) {}
    CIMG_DEV void fail(int chunk, int code) { FOR_LANES_W(l) { if (l == 0) a.status[chunk] = code; } }

    CIMG_DEV void run()
    {
        const uint8_t* const slot = a.zplan + (int64_t)(b - a.blk_first) * a.zplan_stride;
        const int32_t* const head = reinterpret_cast<const int32_t*>(slot);
        const int status = (int)uni((uint32_t)head[0]), nops = (int)uni((uint32_t)head[1]), nrecs = (int)uni((uint32_t)head[2]);
        if (status == ZPLAN_NOT_OURS) return;
        if (status == ZPLAN_FALLBACK) {
            // (a.tune & 2: blocks so large that cimg_decode_zstd cannot take over -- its LDS holds output AND tables: the chunk says so)
            if (a.tune & 2) fail(find_chunk(a.descs, a.nchunks, b, a.uniform_nblocks), ERR_CODEC_SUPPORT);
            // The block is NOT decoded yet, and says so where the host looks: its own word behind the refusal counter (page-locked
            // host memory, as the status words are) becomes ZFALL_PENDING; cimg_decode_zstd turns it into ZFALL_DONE, and the host
            // fails every chunk that still has a pending block when the call ends (the counter alone -- one word, bumped
            // by an atomic across PCIe -- was all that stood between a lost count and pixels that were never written).
            else if (a.skipped) { FOR_LANES_W(l) { if (l == 0) a.skipped[1 + b] = ZFALL_PENDING; } }
            return;
        }
        if (status != ZPLAN_READY) { fail(find_chunk(a.descs, a.nchunks, b, a.uniform_nblocks), status < 0 ? status : ERR_FAILURE); return; }
        const int area = (a.lds_bytes - 64) & ~63;
        ZstdBlockGeom g;
        const int ours = g.parse(a, b, area, 0);
        if (ours <= 0) { if (ours < 0) fail(g.chunk, ours); return; }            // (the walker saw the same header)
        const int cap = (a.zcap + 15) & ~15;
        if (nops < 0 || nops > ZSTD_PLAN_OPS || nrecs < 0 || nrecs > zstd_plan_records(cap)) { fail(g.chunk, ERR_FAILURE); return; }
        const ZstdOp* const ops = reinterpret_cast<const ZstdOp*>(slot + 16);
        const uint64_t* const recs = reinterpret_cast<const uint64_t*>(slot + ZSTD_PLAN_HEAD);
        int pos = g.bstart, opi = 0;
        for (int s = 0; s < g.ns; s++) {
            int cs, payload;
            const int hrc = g.stream_header(pos, cs, payload);
            if (hrc < 0) { fail(g.chunk, hrc); return; }
            uint8_t* plane = lds + s * g.neblock;
            if (cs <= 0) {
                if (cs < 0 && (!(g.c[pos] & 1) || cs < -255)) { fail(g.chunk, ERR_RUN_LENGTH); return; }
                const uint8_t v = (uint8_t)((uint32_t)(-cs) & 0xFF);
                for (int i = 0; i < g.neblock; i += 64) { FOR_LANES_W(l) { if (i + l < g.neblock) plane[i + l] = v; } }
            } else if (cs == g.neblock) {
                if (((s * g.neblock) & 15) == 0) wave_copy_g2l(g.c + pos, lds, s * g.neblock, g.neblock);
                else for (int i = 0; i < g.neblock; i += 64) { FOR_LANES_W(l) { if (i + l < g.neblock) plane[i + l] = g.c[pos + i + l]; } }
            } else if (cs > g.neblock) {
                fail(g.chunk, ERR_DATA); return;
            } else {
                const int r = zstd_replay_frame<cimg_lds_u8p>(ops, opi, nops, s, recs, nrecs, plane, g.neblock, lds, lds + (a.lds_bytes & ~3), &opi);
#if !defined(CIMG_ABL_ZSTD_NO_SEQ) && !defined(CIMG_ABL_ZSTD_NO_EXEC)
                if (r != g.neblock) { fail(g.chunk, r < 0 ? r : ERR_DATA); return; }
#endif
            }
            pos += payload;
        }
        DecodeBlock fb(a, lds, b);
        fb.chunk = g.chunk; fb.j = g.j; fb.bsize = g.bsize; fb.ns = g.ns; fb.neblock = g.neblock; fb.rs = g.neblock; fb.ts = g.ts;
        fb.filter = g.filter; fb.mode = 0; fb.c = g.c; fb.out = g.out;
        for (int q = 0; q < 4; ++q) fb.phase_b(q);
    }
};

}  // namespace cimg
