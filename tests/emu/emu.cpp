// tests/emu/emu.cpp -- TEST INFRASTRUCTURE: the gfx950 kernel bodies compiled for the host with the
// 64-lane emulator of csrc/wave.h, so the kernel *logic* (window commit rules, in-place LZ4 decode,
// layout walk) can be checked against the oracle in a container that has no GPU.  Never linked into
// libcimg_hip.so; nothing in the product loads it.
#define CIMG_EMULATE 1
#include "plan.h"
#include "zstd_kernel.h"
#include "zstd_walk_kernel.h"
#include "zstd_seq_kernel.h"
#include "zstd_lit_kernel.h"
#include "deinterleave_kernel.h"
#include "assemble_kernel.h"
#include "blosclz_kernel.h"
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

// the workgroup's LDS is a host buffer of exactly the launch size (+ slack in the normal test build; the sanitizer
// build uses 0 so that any access past the allocation a real launch would get is reported)
#ifndef EMU_LDS_SLACK
#define EMU_LDS_SLACK 64
#endif

namespace cimg {
int g_emu_zstd_take = 1 << 30; long g_emu_zx_batches = 0, g_emu_zx_rounds = 0, g_emu_zx_par = 0, g_emu_zx_serial = 0, g_emu_zx_longlit = 0; }
namespace cimg { long g_emu_d2[8] = {0, 0, 0, 0, 0, 0, 0, 0}; }
namespace cimg { long g_emu_dec_par = 0, g_emu_dec_serial = 0, g_emu_dec_batches = 0; int g_emu_write_order = 0; long g_emu_windows = 0, g_emu_matches = 0, g_emu_collisions = 0; }
using namespace cimg;
static int g_emu_zstd_plan_cap = -1;    // -1: the block area (the engine's default); 0: no plans (fused kernels only); n: n bytes of records / literals a plan
static long g_emu_zstd_refused = 0;
static int g_emu_zstd_lanes = 8;         // blocks a wave of the lane decoder takes (0: the walkers decode sequences themselves)
static int g_emu_block_items = 1;     // tests also run the one-item-per-plane form
static int g_emu_enc_rt = 0;          // LZ4 / LZ4HC streams through the register-table form of the encoder (CIMG_ENC_RT=1); 0: the LDS-table form, the engine's default

extern "C" {

void emu_set_write_order(int o) { g_emu_write_order = o; }
void emu_dec_stats(long* out) { out[0] = g_emu_dec_par; out[1] = g_emu_dec_serial; out[2] = g_emu_dec_batches; g_emu_dec_par = g_emu_dec_serial = g_emu_dec_batches = 0; }
void emu_stats(long* out, int reset) { out[0] = g_emu_windows; out[1] = g_emu_matches; out[2] = g_emu_collisions; if (reset) g_emu_windows = g_emu_matches = g_emu_collisions = 0; }

struct EmuCParams {
    int32_t typesize, clevel, blocksize, compcode, splitmode;
    uint8_t filters[6], filters_meta[6];
};

static HostCParams to_host(const EmuCParams* p)
{
    HostCParams h;
    h.typesize = p->typesize; h.clevel = p->clevel; h.blocksize = p->blocksize;
    h.compcode = p->compcode; h.splitmode = p->splitmode;
    memcpy(h.filters, p->filters, 6); memcpy(h.filters_meta, p->filters_meta, 6);
    return h;
}

int g_emu_fold = 1;            // 0: the two assembly kernels behind the encode launches even where the launch could assemble
int g_emu_folded = 0;          // chunks of the last batch that were assembled inside an encode launch
extern "C" void emu_set_fold(int on) { g_emu_fold = on; }
extern "C" int emu_last_folded(void) { return g_emu_folded; }

int emu_compress_batch(const EmuCParams* p, int nchunks, const uint8_t* raw, const int64_t* raw_off,
                       const int32_t* nbytes, uint8_t* comp, const int64_t* comp_off, const int32_t* destsize,
                       int32_t* cbytes)
{
    EncodePlan plan;
    int rc = plan_encode_batch(to_host(p), nchunks, raw_off, nbytes, comp_off, destsize, &plan);
    if (rc < 0) return rc;
    std::vector<uint8_t> scratch((size_t)plan.total_blocks * plan.cp.slot_bytes + 64, 0xEE);
    std::vector<StreamRec> recs((size_t)plan.total_blocks * plan.cp.streams_per_block);
    std::vector<ChunkLayout> layout((size_t)nchunks), layout_host((size_t)nchunks + 1);
    // as the engine does (engine.hip: compress_launch): chunks whose streams all belong to one launch are assembled inside it,
    // the two assembly kernels run behind the launches for the others
    const bool fold = g_emu_fold != 0;
    bool leftovers = false;
    if (!fold) for (ChunkDesc& d : plan.descs) d.assemble = 0;
    int folded = 0;
    for (const ChunkDesc& d : plan.descs) { if (!d.assemble) leftovers = true; else folded++; }
    std::vector<uint32_t> chunk_count((size_t)nchunks, 0), ready((size_t)nchunks, 3);
    const uint32_t gen = 5;
    g_emu_folded = folded;
    for (int split = 1; split >= 0; split--) {
        int lds_bytes = split ? plan.lds_split : plan.lds_unsplit;
        if (!lds_bytes) continue;
        const bool rt = g_emu_enc_rt && (plan.cp.compcode == CODEC_LZ4 || plan.cp.compcode == CODEC_LZ4HC);
        if (rt) lds_bytes = encode_lds_bytes_rt(lds_bytes - LZ4_HASH_BYTES);   // (as engine.hip: the plane and its margin, no table)
        std::vector<uint8_t> lds((size_t)lds_bytes + EMU_LDS_SLACK);
        std::vector<uint32_t> queue((size_t)ENC_NQ * ENC_QSTRIDE, 0), queue_next((size_t)ENC_NQ * ENC_QSTRIDE, 77);
        bool block_items = split && g_emu_block_items && !rt;
        if (block_items)
            for (const ChunkDesc& d : plan.descs)
                if (!d.memcpyed && d.split && !encode_block_items_ok(plan.cp.typesize, plan.cp.filter, d.blocksize)) { block_items = false; break; }
        // 1: every block whole; 2: the first half whole, the rest plane by plane (the mixed queue of a small batch)
        const int whole_blocks = !block_items ? 0 : (g_emu_block_items == 2 ? plan.total_blocks / 2 : plan.total_blocks);
        const int zstride = 2 * (plan.cp.max_blocksize / 4 + 64);
        std::vector<uint32_t> zseq((size_t)zstride * 80, 0xA5A5A5A5u);       // (a region per wave: up to 75 waves below)
        static ZstdEncTables ztabs;
        zstd_build_enc_tables(&ztabs);
        // (the waves run one after the other here: the first drains every sub-queue -- its own, then the others as a thief --, the
        // rest find them dry.  Every item is popped by a running wave, so a wave that waits for a chunk never waits for a wave
        // that has not started.  The wave count rotates, so that different waves' "own" sub-queues come first.)
        static int emu_wave_rot = 0;
        const int emu_waves = 1 + (emu_wave_rot++ % 3) * 37;
        std::vector<int32_t> next_item((size_t)encode_items(plan.total_blocks, plan.cp.streams_per_block, split != 0, whole_blocks) + 1, -7);
        EncodeArgs ea{plan.descs.data(), nchunks, plan.cp, raw, scratch.data(), recs.data(), lds_bytes, plan.total_blocks, split, nullptr, queue.data(), plan.uniform_nblocks, whole_blocks,
                      zseq.data(), zstride, &ztabs,
                      queue_next.data(), emu_waves, fold ? 1 : 0, comp, layout.data(), layout_host.data(), chunk_count.data(), ready.data(), next_item.data(), gen};
        for (int w = emu_waves - 1; w >= 0; w--) {  // persistent waves, the highest-numbered first (so that wave 0, which zeroes the next launch's heads, is not the one that encodes)
            memset(lds.data(), 0xCD, lds.size());
            if (plan.cp.compcode == CODEC_BLOSCLZ) { EncodeStream<CODEC_BLOSCLZ> es(&ea, lds.data(), w); es.run(); }
            else if (plan.cp.compcode == CODEC_ZSTD) { EncodeStream<CODEC_ZSTD> es(&ea, lds.data(), w); es.run(); }
            else if (rt) { EncodeStream<CODEC_LZ4_RT> es(&ea, lds.data(), w); es.run(); }
            else { EncodeStream<CODEC_LZ4> es(&ea, lds.data(), w); es.run(); }
        }
        for (int q = 0; q < ENC_NQ; q++) if (queue_next[(size_t)q * ENC_QSTRIDE] != 0) return -1;   // the next launch's heads were zeroed
    }
    if (layout_host[(size_t)nchunks].cbytes < 0) return -1;                      // a wave waited in vain
    for (int c = 0; c < nchunks; c++) if (chunk_count[(size_t)c] != 0) return -1;   // every closer puts its count back to zero
    for (int c = 0; c < nchunks; c++) if (plan.descs[(size_t)c].assemble && ready[(size_t)c] != gen) return -1;
    if (leftovers) {
        AssembleArgs aa{plan.descs.data(), nchunks, plan.cp, raw, scratch.data(), recs.data(), comp, layout.data(), plan.uniform_nblocks, nullptr, 1};
        for (int c = 0; c < nchunks; c++) { LayoutChunk lc(aa, c); lc.run(); }
        for (int b = 0; b < plan.total_blocks; b++) { EmitBlock eb(aa, b); for (int w = 0; w < 4; w++) eb.run(w); }
    }
    for (int c = 0; c < nchunks; c++) cbytes[c] = layout[(size_t)c].cbytes;
    return 0;
}

extern "C" void emu_set_block_items(int on) { g_emu_block_items = on; }
extern "C" void emu_set_enc_rt(int on) { g_emu_enc_rt = on; }
int g_emu_lean_shape = -1;      // <= 0: rotate the number of persistent lean waves (1, 3, 7, one per block), else that many
extern "C" void emu_set_lean_shape(int shape) { g_emu_lean_shape = shape; }
int g_emu_lean = 1;            // tests switch the lean kernel off to cover the general one on every block
long g_emu_lean_blocks = 0;    // blocks the lean kernel produced since the last emu_stats reset
extern "C" void emu_set_lean(int on) { g_emu_lean = on; }
extern "C" void emu_set_zstd_plan(int cap) { g_emu_zstd_plan_cap = cap; }
static int g_emu_zstd_lose_fused = 0;   // test hook: the fused pass behind the replay does not happen (what a lost refusal count did before round 5)
extern "C" void emu_set_zstd_lose_fused(int on) { g_emu_zstd_lose_fused = on; }
extern "C" void emu_set_zstd_lanes(int n) { g_emu_zstd_lanes = n; }
extern "C" long emu_zstd_refused() { const long r = g_emu_zstd_refused; g_emu_zstd_refused = 0; return r; }
extern "C" long emu_lean_blocks(void) { const long n = g_emu_lean_blocks; g_emu_lean_blocks = 0; return n; }

int emu_decompress_batch(int nchunks, const uint8_t* comp, const int64_t* comp_off, const int32_t* nbytes,
                         const int32_t* blocksize, uint8_t* raw, const int64_t* raw_off, int32_t* status)
{
    DecodePlan plan;
    int rc = plan_decode_batch(nchunks, comp_off, nbytes, blocksize, raw_off, &plan);
    if (rc < 0) return rc;
    memset(status, 0, sizeof(int32_t) * (size_t)nchunks);
    std::vector<uint8_t> lds((size_t)plan.lds_bytes + EMU_LDS_SLACK);
    // as the engine does: the lean kernel over every block, then the general kernel over what it left
    std::vector<uint32_t> done((size_t)plan.total_blocks, 0);
    const uint32_t gen = 7;
    if (g_emu_lean) {
        // cimg_decode_lean: persistent waves; wave w of G walks blocks w, w + G, ... with the loads of the next two blocks issued
        // ahead (here they simply happen early).  G rotates so that one-, two- and many-block walks are all covered.
        std::vector<uint32_t> left((size_t)plan.total_blocks + 1, 0);          // one word per wave: the blocks it left to the general kernel
        DecodeArgs la{plan.descs.data(), nchunks, comp, raw, status, plan.lds_lean, nullptr, plan.uniform_nblocks, done.data(), gen, left.data(), plan.total_blocks};
        std::vector<uint8_t> llds((size_t)plan.lds_lean + EMU_LDS_SLACK);
        static int rot = 0;
        const int choices[4] = {1, 3, 7, plan.total_blocks > 0 ? plan.total_blocks : 1};
        const int G = g_emu_lean_shape > 0 ? g_emu_lean_shape : choices[rot++ & 3];
        for (int w = 0; w < G && w < plan.total_blocks; w++) {
            memset(llds.data(), 0xCD, llds.size());
            DecodeLeanWave wave(la, llds.data());
            wave.run(w, G);
        }
        int lean_done = 0;
        for (int b = 0; b < plan.total_blocks; b++) if (done[(size_t)b] == gen) lean_done++;
        g_emu_lean_blocks += lean_done;
        // what the host adds up (engine.hip: decompress_finish) is exactly what the lean waves did not decode
        uint32_t left_sum = 0;
        for (uint32_t v : left) left_sum += v;
        if ((int)left_sum != plan.total_blocks - lean_done) return -1;
    }
    DecodeArgs da{plan.descs.data(), nchunks, comp, raw, status, plan.lds_bytes, nullptr, plan.uniform_nblocks, done.data(), gen, nullptr, plan.total_blocks};
    for (int b = 0; b < plan.total_blocks; b++) {
        memset(lds.data(), 0xCD, lds.size());
        DecodeBlock blk(da, lds.data(), b);
        // each wave keeps its own copy of the uniform walk results on the GPU; emulate that
        DecodeBlock w0 = blk, w1 = blk, w2 = blk, w3 = blk;
        DecodeBlock* ws[4] = {&w0, &w1, &w2, &w3};
        for (int w = 0; w < 4; w++) ws[w]->phase_a(w);
        for (int w = 0; w < 4; w++) ws[w]->phase_b(w);
    }
    // as the engine does (decompress_finish): chunks nobody read yet go to the zstd kernel
    // (one wave per block for chunks of one stream per block, two waves for split ones; each launch reads its own kind only)
    bool unread[2] = {false, false};
    for (int i = 0; i < nchunks; i++) if (status[i] == STATUS_ZSTD_PENDING || status[i] == STATUS_ZSTD_PENDING_SPLIT) { unread[status[i] == STATUS_ZSTD_PENDING_SPLIT] = true; status[i] = 0; }
    int max_bs = 0;
    for (const ChunkDesc& d : plan.descs) max_bs = d.blocksize > max_bs ? d.blocksize : max_bs;
    // The engine's order (engine.hip: decompress_finish): the walk launch (plans in "global memory"), the replay launch, and behind
    // them the fused kernels for the blocks whose plan did not fit its slot -- everything, when the planner is switched off
    // (emu_set_zstd_plan: cap 0 = fused only; a small cap makes plans overflow).
    std::vector<uint8_t> zplan;
    std::vector<uint32_t> fallwords((size_t)plan.total_blocks + 2, 0);   // [0]: the refusal counter, [1 + b]: block b left to / taken by the fused kernel
    uint32_t& refused = fallwords[0];
    bool no_fused = false;
    DecodeArgs zb = da;
    zb.done = nullptr;
    const bool planned = (unread[0] || unread[1]) && g_emu_zstd_plan_cap != 0;
    if (planned) {
        const int area = zstd_kernel_area(max_bs);
        const int cap = g_emu_zstd_plan_cap > 0 ? g_emu_zstd_plan_cap : area;
        const int lanes = g_emu_zstd_lanes;
        const int64_t stride = zstd_plan_stride(cap, lanes > 0);
        zplan.assign((size_t)plan.total_blocks * (size_t)stride, 0xCD);
        zb.skipped = fallwords.data();
        zb.zplan = zplan.data(); zb.zplan_stride = stride; zb.zcap = cap; zb.zarea = area; zb.blk_first = 0; zb.zlanes = lanes; zb.zblocks = plan.total_blocks;
        DecodeArgs wa = zb;
        wa.lds_bytes = zstd_walk_lds_bytes();
        std::vector<uint8_t> wl((size_t)wa.lds_bytes + EMU_LDS_SLACK);
        for (int b = 0; b < plan.total_blocks; b++) {
            memset(wl.data(), 0xCD, wl.size());
            ZstdWalkBlock blk(wa, wl.data(), b);
            blk.run();
        }
        if (lanes > 0) {
            DecodeArgs sa = zb;
            sa.lds_bytes = zstd_seq_lds_bytes(lanes);
            {
                DecodeArgs la = sa;
                la.lds_bytes = zstd_lit_lds_bytes();
                std::vector<uint8_t> ll((size_t)la.lds_bytes);
                for (int g = 0; g * ZSTD_LIT_BLOCKS < plan.total_blocks; g++) {
                    memset(ll.data(), 0xCD, ll.size());
                    ZstdLitLanes w(la, ll.data(), g);
                    w.run();
                }
            }
            std::vector<uint8_t> sl((size_t)sa.lds_bytes);
            for (int g = 0; g * lanes < plan.total_blocks; g++) {
                memset(sl.data(), 0xCD, sl.size());
                ZstdSeqLanes w(sa, sl.data(), g);
                w.run();
            }
        }
        DecodeArgs ra = zb;
        const bool fused_fits = zstd_kernel_lds_bytes(max_bs, 1) <= 163840;          // (engine.hip: blocks the fused kernel's LDS cannot hold)
        ra.tune = fused_fits ? 0 : 2;
        if (!fused_fits) no_fused = true;
        ra.lds_bytes = zstd_replay_lds_bytes(max_bs);
        std::vector<uint8_t> rl((size_t)ra.lds_bytes);            // (exactly: the replay's dword fetches stay inside the launch's LDS)
        for (int b = 0; b < plan.total_blocks; b++) {
            memset(rl.data(), 0xCD, rl.size());
            ZstdReplayBlock blk(ra, rl.data(), b);
            blk.run();
        }
        g_emu_zstd_refused += refused;
    }
    // as the engine does: what the replay left over is read from the blocks' own words, the counter is a second witness
    int pending = 0;
    for (int b = 0; b < plan.total_blocks; b++) pending += fallwords[(size_t)b + 1] == ZFALL_PENDING;
    if (planned && pending != (no_fused ? 0 : (int)refused)) return -1;           // (every refused plan is a pending block, and nothing else is)
    const bool run_fused = (unread[0] || unread[1]) && (!planned || (pending > 0 && !no_fused)) && !(planned && g_emu_zstd_lose_fused);
    if (!run_fused) { unread[0] = false; unread[1] = false; }
    const bool two_waves = unread[1] && zstd_kernel_lds_bytes(max_bs, 2) <= 163840;       // (engine.hip: decompress_finish)
    if (!two_waves) { unread[0] = unread[0] || unread[1]; unread[1] = false; }
    for (int kind = 0; kind < 2; kind++) {
        if (!unread[kind]) continue;
        const int nw = kind ? 2 : 1;
        DecodeArgs za = zb;
        za.lds_bytes = zstd_kernel_lds_bytes(max_bs, nw);
        za.done = nullptr;
        za.tune = two_waves ? 1 : 0;
        std::vector<uint8_t> zl((size_t)za.lds_bytes + EMU_LDS_SLACK);
        for (int b = 0; b < plan.total_blocks; b++) {
            memset(zl.data(), 0xCD, zl.size());
            DecodeZstdBlock w0(za, zl.data(), b, nw), w1(za, zl.data(), b, nw);
            DecodeZstdBlock* ws[2] = {&w0, &w1};
            w0.init();
            // the waves of a block run one after the other here.  Every other block whoever goes first gets every stream from the
            // counter; in the blocks between, a wave comes back after ONE stream and the other one goes on (each call walks the
            // stream headers from the block's first again, over the streams the other wave took)
            const int first = nw > 1 ? (b & 1) : 0;
            g_emu_zstd_take = (nw > 1 && (b & 2)) ? 1 : (1 << 30);
            bool active[2] = {true, nw > 1};
            for (int it = 0; it < 1024 && (active[0] || active[1]); ++it) {
                const int k = (first + it) % nw;
                if (!active[k]) continue;
                ws[k]->phase_a(k);
                const uint32_t* ctl = ws[k]->ctl();
                if ((int)ctl[2 + 2 * k] != ZSTD_BLOCK_FINE || g_emu_zstd_take > 1024 || (int)ctl[0] > ws[k]->ns) active[k] = false;
            }
            g_emu_zstd_take = 1 << 30;
            for (int k = 0; k < nw; k++) ws[k]->phase_b(k);
        }
    }
    for (int b = 0; b < plan.total_blocks; b++) if (fallwords[(size_t)b + 1] == ZFALL_PENDING) return -1;   // a block nobody decoded
    return 0;
}

// interleaved pixels -> planes through the emulated kernel, tile by tile (LDS of exactly the launch size)
int emu_deinterleave(const uint8_t* src, int nch, int ts, int64_t npixels, uint8_t* dst, int64_t plane_stride)
{
    if (nch < 1 || (ts != 1 && ts != 2 && ts != 4 && ts != 8) || (plane_stride & 15) || plane_stride < npixels * ts || nch * ts * 16 > 16384) return -12;
    const int tile = deinterleave_tile_pixels(nch, ts), lds_bytes = deinterleave_lds_bytes(nch, ts);
    DeinterleaveArgs a{src, dst, plane_stride, npixels, nch, ts, tile, lds_bytes};
    for (int64_t t = 0; t * tile < npixels; t++) {
        std::vector<uint8_t> lds((size_t)lds_bytes + EMU_LDS_SLACK, 0xCD);
        deinterleave_wave(a, lds.data(), t);
    }
    return 0;
}

// single-stream entry points for the LZ4 wave codec
int emu_lz4_encode(const uint8_t* src, int n, uint8_t* dst, int cap, int accel, int* need)
{
    int nd = 0, r;
    if (g_emu_enc_rt) {     // the register-table form: the LDS holds the plane and its margin (exactly the launch size)
        std::vector<uint8_t> lds((size_t)encode_lds_bytes_rt(n) + EMU_LDS_SLACK, 0xCD);
        memcpy(lds.data(), src, (size_t)n);
        r = lz4_encode_rt_body(lds.data(), n, dst, cap, accel, nd);
    } else {
        std::vector<uint8_t> lds((size_t)round16(n) + 64 + LZ4_HASH_BYTES, 0xCD);
        memcpy(lds.data(), src, (size_t)n);
        r = lz4_encode_wave(lds.data(), 0, round16(n), n, dst, cap, accel, &nd);
    }
    if (need) *need = nd;
    return r;
}

// one stream through the zstd ENCODER of csrc/zstd_encode.h (the LZ4 match finder with a sequence sink + the frame around it)
int emu_zstd_encode(const uint8_t* src, int n, uint8_t* dst, int cap)
{
    if (n > ZSTD_ENC_MAX_INPUT) return -1;
    std::vector<uint8_t> lds((size_t)encode_lds_bytes(n, CODEC_ZSTD) + EMU_LDS_SLACK, 0xCD);
    memcpy(lds.data(), src, (size_t)n);
    std::vector<uint32_t> seq((size_t)2 * (n / 4 + 64), 0xA5A5A5A5u);
    static ZstdEncTables tabs;
    zstd_build_enc_tables(&tabs);
    SeqSink sink{seq.data(), 0, 0};
    int need = 0;
    const int prefix = zstd_frame_prefix(n);
    if (prefix + 16 >= cap) return 0;
    const int nseq = lz4_encode_body<true>(lds.data(), lds.data() + round16(n), n, dst + prefix, cap - prefix, 1, need, nullptr, 0, &sink);
    if (nseq <= 0) return 0;
    return zstd_finish_frame(lds.data(), round16(n), n, dst, cap, sink, &tabs);
}

// one zstd frame through csrc/zstd_decode.h (the work area the kernel keeps in LDS is on the heap here), the ways the kernel
// uses it: the frame as a copy the decoder reads in place (tail = 0), and the frame "in global memory" (tail = 1) with a stage
// like the kernel's and with one so small that most sections are read where they lie.  All must agree; -999 says they did not.
void emu_d2_stats(long* out) { for (int i = 0; i < 8; i++) { out[i] = g_emu_d2[i]; g_emu_d2[i] = 0; } }
void emu_zstd_exec_stats(long* out) { out[0] = g_emu_zx_batches; out[1] = g_emu_zx_rounds; out[2] = g_emu_zx_par; out[3] = g_emu_zx_serial; out[4] = g_emu_zx_longlit;
    g_emu_zx_batches = g_emu_zx_rounds = g_emu_zx_par = g_emu_zx_serial = g_emu_zx_longlit = 0; }
int emu_zstd_decode(const uint8_t* src, int csize, uint8_t* dst, int cap)
{
    std::vector<uint8_t> in(src, src + csize);            // exact-size copy: a read past the end is an ASAN finding
    int rc3[3] = {0, 0, 0};
    std::vector<uint8_t> out3[3];
    for (int mode = 0; mode < 3; ++mode) {
        const int stage_cap = mode == 1 ? (int)ZSTD_KERNEL_STAGE : 96;
        std::vector<uint8_t> stage((size_t)stage_cap + 16);
        std::vector<ZstdWork> w(1);
        w[0].stage = stage.data() + ((16 - ((uintptr_t)stage.data() & 15)) & 15);
        w[0].stage_cap = stage_cap;
        w[0].tail = mode ? 1 : 0;
        out3[mode].assign((size_t)cap, 0);
        // (the executor's dword fetches stay inside the output buffer: exactly, so that ASAN sees a read past it)
        w[0].mem_lo = (const uint8_t*)(((uintptr_t)out3[mode].data() + 3) & ~(uintptr_t)3);
        w[0].mem_hi = (const uint8_t*)(((uintptr_t)out3[mode].data() + (size_t)cap) & ~(uintptr_t)3);
        if (mode == 2) w[0].mem_lo = w[0].mem_hi = nullptr;      // and once with nothing readable: every copy the serial way
        rc3[mode] = zstd_decode_frame(in.data(), csize, out3[mode].data(), cap, &w[0]);
    }
    if (getenv("CIMG_EMU_ZSTD_DEBUG") && rc3[0] > 0) {
        for (int m = 0; m < 2; ++m) {
            size_t k = 0;
            while (k < (size_t)rc3[0] && out3[m][k] == out3[2][k]) ++k;
            if (k < (size_t)rc3[0]) fprintf(stderr, "[emu zstd] mode %d differs from mode 2 at byte %zu of %d\n", m, k, rc3[0]);
        }
    }
    // the frame WALKED into a plan (zstd_walk_kernel.h: ops, records, coded literals -- exact-size areas here) and the plan replayed:
    // the same answer; a plan that does not fit its areas is the one thing that may differ (the kernel then takes the decoder proper)
    {
        const int stage_cap = (int)ZSTD_KERNEL_STAGE;
        std::vector<uint8_t> stage((size_t)stage_cap + 16);
        std::vector<ZstdWork> w(1);
        w[0].stage = stage.data() + ((16 - ((uintptr_t)stage.data() & 15)) & 15);
        w[0].stage_cap = stage_cap;
        w[0].tail = 1;
        std::vector<ZstdOp> ops(16);
        std::vector<uint64_t> recs((size_t)cap / 3 + 8);
        std::vector<uint8_t> lits((size_t)((cap + 15) & ~15) + 16);
        w[0].ops = ops.data(); w[0].op_cap = (int)ops.size(); w[0].recs = recs.data(); w[0].rec_cap = (int)recs.size();
        w[0].lits = lits.data(); w[0].lit_cap = (int)lits.size() & ~15; w[0].stream = 5;
        w[0].mem_lo = stage.data(); w[0].mem_hi = stage.data() + 8;          // ("this is the kernel": the typed Huffman loop where it applies)
        const int rw = zstd_decode_frame(in.data(), csize, nullptr, cap, &w[0]);
        if (rw != ZSTD_WALK_OVERFLOW) {
            // (a walker may accept what only the output shows to be wrong: the replay must refuse it then)
            if (rw >= 0) {
                std::vector<uint8_t> out((size_t)cap, 0);
                const uint8_t* lo = (const uint8_t*)(((uintptr_t)out.data() + 3) & ~(uintptr_t)3);
                const uint8_t* hi = (const uint8_t*)(((uintptr_t)out.data() + (size_t)cap) & ~(uintptr_t)3);
                int next = 0;
                const int rr = zstd_replay_frame<uint8_t*>(ops.data(), 0, w[0].op_n, 5, recs.data(), w[0].rec_n, out.data(), cap, lo, hi, &next);
                if ((rr >= 0) != (rc3[0] >= 0)) return -997;
                if (rr >= 0 && (rr != rc3[0] || next != w[0].op_n || (rr > 0 && memcmp(out.data(), out3[0].data(), (size_t)rr)))) return -996;
            } else if (rc3[0] >= 0) return -995;
        }
    }
    for (int mode = 1; mode < 3; ++mode)
        if ((rc3[0] >= 0) != (rc3[mode] >= 0) || (rc3[0] >= 0 && (rc3[0] != rc3[mode] || (rc3[0] > 0 && memcmp(out3[0].data(), out3[mode].data(), (size_t)rc3[0]))))) return -999;
    if (rc3[0] > 0) memcpy(dst, out3[0].data(), (size_t)rc3[0]);
    return rc3[0];
}

int emu_lz4_decode(const uint8_t* src, int csize, uint8_t* dst, int n)
{
    const int rs = region_stride(n);
    std::vector<uint8_t> lds((size_t)rs + 32 + EMU_LDS_SLACK, 0xCD);
    const int park = rs - round16(csize);
    memcpy(lds.data() + park, src, (size_t)csize);
    const int rc = lz4_decode_wave(lds.data(), 0, n, park, csize, rs + 32);
    // (the second form of the decoder -- tokens first, bytes 63 at a time -- on the same stream: it must agree, also on damaged ones)
    std::vector<uint8_t> lds2((size_t)rs + 32 + EMU_LDS_SLACK, 0xCD);
    memcpy(lds2.data() + park, src, (size_t)csize);
    const int rc2 = lz4_decode_wave2(lds2.data(), 0, n, park, csize, rs + 32);
    if ((rc >= 0) != (rc2 >= 0) || (rc >= 0 && memcmp(lds.data(), lds2.data(), (size_t)n))) return -999;
    memcpy(dst, lds.data(), (size_t)n);
    return rc;
}

// single-stream entry points for the BloscLZ wave codec
int emu_blosclz_encode(const uint8_t* src, int n, uint8_t* dst, int cap, int clevel, int* need)
{
    const int lds_bytes = blz_encode_lds_bytes(n);
    std::vector<uint8_t> lds((size_t)lds_bytes + EMU_LDS_SLACK, 0xCD);
    memcpy(lds.data(), src, (size_t)n);
    int nd = 0;
    const int r = blosclz_encode_body(lds.data(), lds.data() + round16(n) + 16, n, dst, cap, clevel, nd);
    if (need) *need = nd;
    return r;
}

int emu_blosclz_decode(const uint8_t* src, int csize, uint8_t* dst, int n)
{
    const int rs = blz_region_stride(n);
    std::vector<uint8_t> lds((size_t)rs + 32 + EMU_LDS_SLACK, 0xCD);
    const int park = rs - round16(csize);
    memcpy(lds.data() + park, src, (size_t)csize);
    const int rc = blosclz_decode_wave(lds.data(), 0, n, park, csize, rs + 32);
    memcpy(dst, lds.data(), (size_t)n);
    return rc;
}

}  // extern "C"
