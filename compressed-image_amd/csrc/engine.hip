// engine.hip -- libcimg_hip.so: kernel entry points, the batch engine and the C ABI of
// include/cimg_hip.h.  gfx950 only.  There is no CPU path in this library: every codec call ends in
// one of the four kernels below or in an error code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "plan.h"
#include "assemble_kernel.h"
#include "deinterleave_kernel.h"
#include "zstd_kernel.h"
#include "zstd_walk_kernel.h"
#include "zstd_seq_kernel.h"
#include "zstd_lit_kernel.h"
#include "../../include/cimg_hip.h"

using namespace cimg;

#ifndef CIMG_ENC_GANG_MAX
#define CIMG_ENC_GANG_MAX 8            // waves per encode workgroup at most (encode_gang)
#endif
enum : int { LDS_GRANULE = 1280 };      // bytes; LDS is handed out per workgroup in multiples of this (measured, tests/ubench/launch.hip)

// ====================================================================================================
//  kernels
// ====================================================================================================
// Persistent chains.  A workgroup is a GANG of independent waves: each owns lds_bytes of the workgroup's LDS and runs its own
// loop over the work queue; they never synchronise.  Why gangs: a CU hands out its 160 KiB of LDS per WORKGROUP in 1280-byte
// granules (measured, tests/ubench/launch.hip: single-wave workgroups of 32000 B -> 5 per CU, of 32256 .. 32768 B -> 4), so
// five one-wave workgroups of 32 KiB (plane + byU16 table) do not fit a CU, but ONE five-wave workgroup of exactly 160 KiB does.
template <int CODEC> CIMG_DEV void encode_gang(uint8_t* lds)
{
    const auto ap = kernel_args<EncodeArgs>();                   // read through the kernel-argument segment (wave.h: kernarg_ptr)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gang = (int)(blockDim.x >> 6);
    EncodeStream<CODEC> es(ap, lds + (size_t)wave * (size_t)fresh(ap)->lds_bytes, (int)blockIdx.x * gang + wave);
    es.run();
}

extern "C" __global__ __launch_bounds__(CIMG_ENC_GANG_MAX * 64) void cimg_encode_streams(EncodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    (void)a;
    encode_gang<CODEC_LZ4>(lds);
}

// LZ4 / LZ4HC streams with the hash table in registers (encode_rt_kernel.h): LDS is the plane alone, eight chains a CU
extern "C" __global__ __launch_bounds__(CIMG_ENC_GANG_MAX * 64) void cimg_encode_streams_rt(EncodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    (void)a;
    encode_gang<CODEC_LZ4_RT>(lds);
}

extern "C" __global__ __launch_bounds__(CIMG_ENC_GANG_MAX * 64) void cimg_encode_streams_blosclz(EncodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    (void)a;
    encode_gang<CODEC_BLOSCLZ>(lds);
}

// codec::zstd: the LZ4 match finder with a sequence sink + one zstd frame per stream (zstd_encode.h; format-valid, not byte-pinned)
extern "C" __global__ __launch_bounds__(CIMG_ENC_GANG_MAX * 64) void cimg_encode_streams_zstd(EncodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    (void)a;
    encode_gang<CODEC_ZSTD>(lds);
}

// interleaved pixels -> planes, one wave per 16 KiB tile (deinterleave_kernel.h)
extern "C" __global__ __launch_bounds__(64) void cimg_deinterleave(DeinterleaveArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    deinterleave_wave(a, lds, (int64_t)blockIdx.x);
}

extern "C" __global__ __launch_bounds__(64) void cimg_layout_chunks(AssembleArgs a)
{
    LayoutChunk lc(a, (int)blockIdx.x);
    lc.run();
}

extern "C" __global__ __launch_bounds__(256) void cimg_emit_blocks(AssembleArgs a)
{
    EmitBlock eb(a, (int)blockIdx.x);
    eb.run(__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
}

// persistent: one wave per workgroup owns one LDS plane and walks blocks blockIdx.x, + gridDim.x, ... (decode_lean_kernel.h).
// Two waves per SIMD (the stored plane of a block travels through its chain in 64 VGPRs): at most 256 registers.
extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void cimg_decode_lean(DecodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    DecodeLeanWave w(a, lds);
    w.run((int)blockIdx.x, (int)gridDim.x);
}

// blocks of zstd-coded chunks: the slow path that keeps chunks written with enums::codec::zstd readable (zstd_kernel.h); launched
// only behind a batch in which cimg_decode_blocks met such a chunk.  One wave per block for chunks of one stream per block, two
// waves per block -- each with tables of its own, taking streams from a counter -- for split ones.
extern "C" __global__ __launch_bounds__(64) void cimg_decode_zstd(DecodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    DecodeZstdBlock blk(a, lds, a.blk_first + (int)blockIdx.x, 1);
    blk.init();
    blk.phase_a(0);
    blk.phase_b(0);
}
// the zstd read path as two launches (zstd_walk_kernel.h): what the frames say goes to a plan in global memory (entropy decoding,
// eight waves a CU), then the plans are replayed into the blocks (four waves a CU, the output in LDS)
extern "C" __global__ __launch_bounds__(64) void cimg_zstd_walk(DecodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    ZstdWalkBlock blk(a, lds, a.blk_first + (int)blockIdx.x);
    blk.run();
}
// the Huffman-coded literals of the blocks' jobs, one lane per stream (zstd_lit_kernel.h)
extern "C" __global__ __launch_bounds__(64) void cimg_zstd_lit(DecodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    ZstdLitLanes w(a, lds, (int)blockIdx.x);
    w.run();
}
// the sequences of the blocks' jobs, one LANE per block (zstd_seq_kernel.h)
extern "C" __global__ __launch_bounds__(64) void cimg_zstd_seq(DecodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    ZstdSeqLanes w(a, lds, (int)blockIdx.x);
    w.run();
}
extern "C" __global__ __launch_bounds__(64) void cimg_zstd_replay(DecodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    ZstdReplayBlock blk(a, lds, a.blk_first + (int)blockIdx.x);
    blk.run();
}
extern "C" __global__ __launch_bounds__(128) void cimg_decode_zstd_split(DecodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    DecodeZstdBlock blk(a, lds, a.blk_first + (int)blockIdx.x, 2);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave == 0) blk.init();
    __syncthreads();
    blk.phase_a(wave);
    __syncthreads();
    blk.phase_b(wave);
}

extern "C" __global__ __launch_bounds__(256) void cimg_decode_blocks(DecodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    // (blk_step == 0: workgroup k takes the LAST block of chunk k -- the leftover blocks of a batch, whatever its chunk sizes)
    const int b = a.blk_step ? a.blk_first + (int)blockIdx.x * a.blk_step
                             : __builtin_amdgcn_readfirstlane(a.descs[blockIdx.x].blk0 + a.descs[blockIdx.x].nblocks - 1);
    DecodeBlock blk(a, lds, b);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#ifdef CIMG_PROFILE
    // diagnostic builds: the 16 uint64 per workgroup carry the LZ4 decoder's cycle laps instead of phase stamps
    blk.phase_a(wave);
    __syncthreads();
    blk.phase_b(wave);
#else
    if (wave == 0) debug_stamp(a.dbg, b, 0);
    blk.phase_a(wave);
    if (wave == 0) debug_stamp(a.dbg, b, 1);                     // stream 0 staged
    if (wave == 1) debug_stamp(a.dbg, b, 2);                     // stream 1 staged
    __syncthreads();
    blk.phase_b(wave);
    if (wave == 0) debug_stamp(a.dbg, b, 3);
#endif
}

// ====================================================================================================
//  engine
// ====================================================================================================
namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
};
struct EventPair {
    hipEvent_t a, b;
};

}  // namespace

struct cimg_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    // host-buffer calls: pixels / chunks of the NEXT group travel over PCIe (copy streams) while the kernels of the
    // current group run (stream) and the results of the PREVIOUS one travel back
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipStream_t s_side = nullptr;       // the small one of two encode launches of a batch runs here, beside the large one (compress_launch)
    hipEvent_t ev_side_pre = nullptr, ev_side_done = nullptr;
    bool side_used = false;             // something was launched on s_side since the last synchronize (an error path must not leave it running)
    bool no_side = getenv("CIMG_NO_SIDE_STREAM") != nullptr;   // diagnostic: the two encode launches one behind the other again
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_done = nullptr;
    int64_t host_register_bytes = getenv("CIMG_HOST_REGISTER_MIB") ? atoll(getenv("CIMG_HOST_REGISTER_MIB")) << 20 : 32ll << 20;   // pageable host spans from this size on are page-locked for the call (HostPin); 0: never
    int64_t host_registrations = 0;
    int64_t host_group_bytes = getenv("CIMG_HOST_GROUP_MIB") ? atoll(getenv("CIMG_HOST_GROUP_MIB")) << 20 : 16ll << 20;   // measured: 8 / 16 / 32 / 64 MiB -> 45.9 / 46.6 / 44.3 / 39.0 GB/s
    DevBuf descs_enc, descs_dec, recs, layout, scratch, stage_raw, stage_comp, stage_il, dbg, queue;
    // chunk descriptors last uploaded for encode / decode: a batch with the same geometry as the previous one
    // (the steady state of an image pipeline) skips the upload
    std::vector<uint8_t> shadow_enc, shadow_dec;
    bool spin_sync = getenv("CIMG_SYNC_SPIN") != nullptr;
    hipEvent_t sync_ev = nullptr;
    std::vector<int64_t> fetch_off;     // device staging offsets / sizes of the chunks of the last host _begin
    std::vector<int32_t> fetch_len;
    // one batch at a time per engine: the calls share the stream, the staging buffers and the result area.
    // (recursive: the host-buffer calls run the device calls inside)
    std::recursive_mutex mu;
    // `qheads`: the encode launches' work-queue heads -- per kind of launch (split / unsplit) two sets of ENC_NQ heads; a launch pops
    // one set and zeroes the other for the next launch of its kind (encode_kernel.h, "the work queue").
    // `sync` holds (behind 16 unused words): per chunk the count of finished streams, then per chunk the generation at which the
    // chunk was last laid out inside a launch.  Both are zeroed when (re)allocated, when the generation wraps and after any
    // failed batch (sync_dirty).
    DevBuf sync, next_item, qheads, zstd_seq, zstd_tables;   // (zstd encoder: per-wave sequence records, FSE tables of the predefined distributions)
    uint32_t qpar[2] = {0, 0}, fold_gen = 0;   // qpar: which of its two sets of queue heads the next split / unsplit launch pops
    bool sync_dirty = true;
    size_t sync_chunks = 0;             // chunks the current layout of `sync` was made for
    bool no_fold = getenv("CIMG_NO_ASSEMBLE_IN_LAUNCH") != nullptr;   // diagnostic: cimg_layout_chunks / cimg_emit_blocks behind every encode launch
    int64_t fold_batches = 0;
    // decode: the lean kernel (decode_lean_kernel.h) runs in front of the general one while it pays off
    DevBuf done;                        // uint32 per block: generation stamp of the lean kernel
    int lean_wgs_cu = 0, lean_wgs_lds = -1;   // resident lean decode waves per CU for that much LDS (occupancy query, cached)
    int lean_wgs_limit = getenv("CIMG_LEAN_WGS_PER_CU") ? atoi(getenv("CIMG_LEAN_WGS_PER_CU")) : 0;   // diagnostic: fewer persistent waves
    uint32_t done_gen = 0;
    // DecodeArgs::tune.  Two lean waves share a SIMD, the issue arbiter serves the older first, and in the first round -- every wave
    // of the CU in the same phase of the same code -- the younger one pays for it (tools/diag_stamps_hw.py: first block 22 us in the
    // older slot, 27 in the younger; later blocks 21 / 22).  Default: every wave of a CU starts (2 x simd + slot) x 16 x 64 cycles
    // late, and the younger wave runs its later blocks at raised priority: 66 -> 64 us on configs[1] (0: off).
    int lean_tune = getenv("CIMG_LEAN_TUNE") ? atoi(getenv("CIMG_LEAN_TUNE")) : (0x10000 | 0x100 | (3 << 9) | 16);
    int lean_hold = getenv("CIMG_NO_LEAN") ? (1 << 30) : 0;   // batches for which the lean launch is skipped
    int64_t zstd_batches = 0;           // decode batches that needed cimg_decode_zstd
    int64_t zstd_blocks_refused = 0;    // blocks whose plan did not fit its slot (decoded by cimg_decode_zstd behind the two launches)
    int zstd_fused = getenv("CIMG_ZSTD_FUSED") ? atoi(getenv("CIMG_ZSTD_FUSED")) : 0;          // 1: cimg_decode_zstd only (no walk / replay launches)
    int zstd_plan_cap = getenv("CIMG_ZSTD_PLAN_CAP") ? atoi(getenv("CIMG_ZSTD_PLAN_CAP")) : 0;   // diagnostic: bytes of records / literals a plan may take (0: the block area)
    int64_t zstd_plan_bytes = getenv("CIMG_ZSTD_PLAN_MIB") ? atoll(getenv("CIMG_ZSTD_PLAN_MIB")) << 20 : 1024ll << 20;   // device memory for the plans of one group of launches (a plan is ~4.25 x its block: 139 KB for 32 KiB; larger batches go in groups; round 4 took up to 2 GiB and kept it: ADVICE r4.  Measured on configs[4]'s share, 1 GiB of libzstd-22 chunks: 2 GiB = 2 groups 17.9 ms, 512 MiB = 9 groups 19.6 ms)
    bool zstd_plan_fail = getenv("CIMG_ZSTD_PLAN_FAIL") != nullptr;   // test hook: the plans' device memory cannot be had (the fused kernel reads the batch)
    int zstd_walk_stage = getenv("CIMG_ZSTD_WALK_STAGE") ? atoi(getenv("CIMG_ZSTD_WALK_STAGE")) : 2048;   // bytes of LDS through which a walker reads a frame's sections (measured on 128 MiB of level-22 float32: 8192 = 8 waves a CU 2.51 ms, 4096 = 10 waves 2.45, 2048 = 11 waves 2.28 -- a section that does not fit is read where it lies)
    int zstd_lanes = getenv("CIMG_ZSTD_LANES") ? atoi(getenv("CIMG_ZSTD_LANES")) : 8;   // blocks a wave of cimg_zstd_seq decodes side by side (0: the walkers decode sequences themselves)
    DevBuf zplan;
    int64_t lean_batches = 0, lean_blocks_skipped = 0, lean_blocks_total = 0;
    uint32_t lean_last_skipped = 1;     // blocks the previous lean batch left over BEYOND the leftover blocks its geometry announced (1: unknown yet -> general kernel enqueued up front)
    int num_cus = 256;
    int enc_waves_cu = 1;               // encode waves per CU the registers allow (occupancy query, cached with enc_wgs_lds)
    int lds_per_cu = 163840, lds_per_wg = 65536;      // device properties
    int enc_gang = getenv("CIMG_ENC_GANG") ? atoi(getenv("CIMG_ENC_GANG")) : 0;
    int enc_rt = getenv("CIMG_ENC_RT") ? atoi(getenv("CIMG_ENC_RT")) : 0;   // LZ4 / LZ4HC: the encoder with its hash table in registers (encode_rt_kernel.h; experimental: measured slower, LABNOTES.md); 0: the LDS-table form
    int enc_said_lds = -1, enc_said_gang = -1;
    int enc_wgs_lds[2] = {-1, -1};
    int enc_wgs_codec = -1;
    bool stamps = false;
    bool trace = getenv("CIMG_TRACE") != nullptr;
    // environment knobs are read ONCE, when the engine is created (diagnostics only; none changes results)
    bool verbose = getenv("CIMG_VERBOSE") != nullptr;
    int enc_wgs_limit = getenv("CIMG_ENC_WGS_PER_CU") ? atoi(getenv("CIMG_ENC_WGS_PER_CU")) : 0;
    int enc_whole_rounds = getenv("CIMG_ENC_WHOLE_ROUNDS") ? atoi(getenv("CIMG_ENC_WHOLE_ROUNDS")) : 1 << 20;   // diagnostic: of the full rounds of a small batch, how many go out as whole blocks
    int enc_block_items = getenv("CIMG_ENC_BLOCK_ITEMS") ? atoi(getenv("CIMG_ENC_BLOCK_ITEMS")) : -1;   // -1: by batch size
    int enc_hybrid = getenv("CIMG_ENC_HYBRID") ? atoi(getenv("CIMG_ENC_HYBRID")) : 1;                  // 0: a small batch goes plane by plane throughout
    int lean_lds_pad = getenv("CIMG_LEAN_LDS_PAD") ? atoi(getenv("CIMG_LEAN_LDS_PAD")) : 0;   // diagnostic: fewer resident lean decode workgroups
    int dbg_count[2] = {0, 0};          // workgroups stamped by the last encode / decode launch
    PinBuf h_descs, h_descs_dec, h_out, h_dec;      // compress and decompress batches may be in flight together: nothing pinned is shared
    int max_dyn_lds[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // largest dynamic LDS already enabled for encode (lz4) / decode / lean decode / encode (blosclz) / encode (zstd) / decode (zstd), one and two waves per block
    bool timing = false;              // events around the kernels of the current batch call
    int timing_period = 0;            // 0 = off, n = every n-th batch call is timed
    int64_t batch_no[2] = {0, 0};     // compress / decompress batch calls since timing was switched on
    void begin_batch(int which) { timing = timing_period > 0 && (batch_no[which]++ % timing_period) == 0; }
    std::vector<EventPair> pending[CIMG_K_COUNT];
    std::vector<EventPair> pending_extra;   // late general decode launches: time counts towards CIMG_K_DECODE, launches do not
    std::vector<EventPair> free_events;
    std::vector<float> samples[CIMG_K_COUNT];   // every timed launch since the last reset (medians: cimg_engine_kernel_samples)
    double total_ms[CIMG_K_COUNT] = {};
    int64_t launches[CIMG_K_COUNT] = {};
    std::string err;
    // a decode batch between decompress_launch() and decompress_finish()
    struct DecodeFlight {
        bool lean = false, general_now = true, timed = false;
        bool launched = false;            // decompress_launch got as far as enqueueing kernels: decompress_finish has something to wait for
        int32_t nchunks = 0, total_blocks = 0, lds_bytes = 0, max_blocksize = 0;
        int32_t lean_grid = 0;            // waves of the lean launch = words of its left-over counts behind the status words
        int32_t known_left = 0;           // leftover blocks (one per chunk) that the geometry says the lean kernel leaves
        size_t st_bytes = 0;
        DecodeArgs da{};
    } dflight;
    bool dflight_open = false;            // between cimg_decompress_batch_device_begin and _fetch
    int32_t cflight_chunks = -1;          // chunks of the compress batch between _device_begin and _device_fetch (-1: none)
    bool claunched = false;               // compress_launch got past the planner and the allocations: h_out holds (or will hold) this batch's sizes

    int fail(int code, const char* fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
    int hip(hipError_t e, const char* what)
    {
        if (e == hipSuccess) return 0;
        return fail(ERR_FAILURE, "%s: %s", what, hipGetErrorString(e));
    }
    int reserve(DevBuf& b, size_t bytes)
    {
        if (bytes <= b.cap) return 0;
        if (b.p) { hipError_t e = hipFree(b.p); b.p = nullptr; b.cap = 0; if (e != hipSuccess) return hip(e, "hipFree"); }
        const size_t want = std::max(bytes + bytes / 4, (size_t)4096);
        int rc = hip(hipMalloc(&b.p, want), "hipMalloc");
        if (rc) return rc;
        b.cap = want;
        return 0;
    }
    // kernels write their small results (chunk sizes, status words) straight into pinned host memory:
    // no D2H copy node between the last kernel and the synchronize
    template <class T> int device_alias(PinBuf& b, T** out)
    {
        void* d = nullptr;
        int rc = hip(hipHostGetDevicePointer(&d, b.p, 0), "hipHostGetDevicePointer");
        *out = (T*)d;
        return rc;
    }
    // descriptors -> device, unless the device copy already holds exactly these bytes
    int upload_descs(DevBuf& dev, std::vector<uint8_t>& shadow, PinBuf& staging, const void* src, size_t bytes)
    {
        if (shadow.size() == bytes && dev.p && memcmp(shadow.data(), src, bytes) == 0) return 0;
        int rc;
        shadow.clear();
        if ((rc = reserve(dev, bytes))) return rc;
        if ((rc = reserve(staging, bytes))) return rc;
        memcpy(staging.p, src, bytes);
        // every batch of one kind ends with a stream synchronize before the next of its kind begins, so its staging
        // buffer is free again
        if ((rc = hip(hipMemcpyAsync(dev.p, staging.p, bytes, hipMemcpyHostToDevice, stream), "descs H2D"))) return rc;
        shadow.assign((const uint8_t*)src, (const uint8_t*)src + bytes);
        return 0;
    }
    int reserve(PinBuf& b, size_t bytes)
    {
        if (bytes <= b.cap) return 0;
        if (b.p) { (void)hipHostFree(b.p); b.p = nullptr; b.cap = 0; }
        const size_t want = std::max(bytes + bytes / 4, (size_t)4096);
        int rc = hip(hipHostMalloc(&b.p, want, hipHostMallocDefault), "hipHostMalloc");
        if (rc) return rc;
        b.cap = want;
        return 0;
    }
    // a launch on the side stream is ordered behind everything `stream` holds now
    int side_follows_stream()
    {
        int rc;
        if ((rc = hip(hipEventRecord(ev_side_pre, stream), "event record"))) return rc;
        if ((rc = hip(hipStreamWaitEvent(s_side, ev_side_pre, 0), "stream wait"))) return rc;
        side_used = true;
        return 0;
    }
    EventPair get_events()
    {
        if (!free_events.empty()) { EventPair e = free_events.back(); free_events.pop_back(); return e; }
        EventPair e{};
        (void)hipEventCreate(&e.a);
        (void)hipEventCreate(&e.b);
        return e;
    }
    void drain_timing()
    {
        for (int k = 0; k < CIMG_K_COUNT; k++) {
            for (EventPair& ev : pending[k]) {
                float ms = 0.f;
                if (hipEventSynchronize(ev.b) == hipSuccess && hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
                    total_ms[k] += ms;
                    launches[k] += 1;
                    if (samples[k].size() < 65536) samples[k].push_back(ms);
                }
                free_events.push_back(ev);
            }
            pending[k].clear();
        }
        for (EventPair& ev : pending_extra) {
            float ms = 0.f;
            if (hipEventSynchronize(ev.b) == hipSuccess && hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) total_ms[CIMG_K_DECODE] += ms;
            free_events.push_back(ev);
        }
        pending_extra.clear();
    }
    template <class Args>
    int launch(int kid, void (*kernel)(Args), const Args& args, int grid, int block, int lds, hipStream_t on = nullptr)
    {
        if (grid <= 0) return 0;
        if (!on) on = stream;
        EventPair ev{};
        if (timing) { ev = get_events(); (void)hipEventRecord(ev.a, on); }
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3((unsigned)block), (size_t)lds, on, args);
        int rc = hip(hipGetLastError(), cimg_kernel_name(kid));
        if (timing) { (void)hipEventRecord(ev.b, on); pending[kid].push_back(ev); }
        if (!rc && trace) {             // CIMG_TRACE=1: find the launch that does not come back
            fprintf(stderr, "[cimg] launched %s grid %d block %d lds %d ... ", cimg_kernel_name(kid), grid, block, lds);
            fflush(stderr);
            rc = hip(hipStreamSynchronize(on), cimg_kernel_name(kid));
            fprintf(stderr, "done (%d)\n", rc);
        }
        return rc;
    }
    template <class Args>
    int allow_lds(void (*kernel)(Args), int which, int bytes)
    {
        if (bytes <= max_dyn_lds[which]) return 0;
        int rc = hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes),
                     "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        if (rc) return rc;
        max_dyn_lds[which] = bytes;
        return 0;
    }
};

extern "C" {

const char* cimg_kernel_name(int k)
{
    switch (k) {
    case CIMG_K_ENCODE: return "cimg_encode_streams";
    case CIMG_K_LAYOUT: return "cimg_layout_chunks";
    case CIMG_K_EMIT: return "cimg_emit_blocks";
    case CIMG_K_DECODE: return "cimg_decode_blocks";
    case CIMG_K_DEINTERLEAVE: return "cimg_deinterleave";
    case CIMG_K_DECODE_ZSTD: return "cimg_decode_zstd";
    case CIMG_K_ENCODE_ZSTD: return "cimg_encode_streams_zstd";
    case CIMG_K_ZSTD_WALK: return "cimg_zstd_walk";
    case CIMG_K_ZSTD_REPLAY: return "cimg_zstd_replay";
    case CIMG_K_ZSTD_FUSED: return "cimg_decode_zstd_fused";
    case CIMG_K_ZSTD_SEQ: return "cimg_zstd_seq";
    case CIMG_K_ZSTD_LIT: return "cimg_zstd_lit";
    default: return "?";
    }
}

void cimg_cparams_init(cimg_cparams* p, int32_t typesize)
{
    memset(p, 0, sizeof(*p));
    p->typesize = typesize;
    p->clevel = 9;
    p->blocksize = 32768;
    p->compcode = CODEC_LZ4;
    p->splitmode = SPLIT_AUTO;
    p->filters[5] = FILTER_SHUFFLE;
}

int cimg_engine_create(int device, cimg_engine** out)
{
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        g_create_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return ERR_FAILURE;
    }
    if (device < 0) { e = hipGetDevice(&device); if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return ERR_FAILURE; } }
    if (device >= count) { g_create_error = "device index out of range"; return ERR_INVALID_PARAM; }
    e = hipSetDevice(device);
    if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return ERR_FAILURE; }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return ERR_FAILURE; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("libcimg_hip.so is built for gfx950 only; device is ") + prop.gcnArchName;
        return ERR_FAILURE;
    }
    cimg_engine* eng = new cimg_engine();
    eng->device = device;
    eng->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (prop.maxSharedMemoryPerMultiProcessor > 0) eng->lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    {   // what ONE workgroup may allocate (gfx950: the whole 160 KiB)
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, device) == hipSuccess && v > 0) eng->lds_per_wg = v;
        else if (prop.sharedMemPerBlock > 0) eng->lds_per_wg = (int)prop.sharedMemPerBlock;
        if (eng->verbose) fprintf(stderr, "[cimg] LDS: %d bytes per CU, %d per workgroup\n", eng->lds_per_cu, eng->lds_per_wg);
    }
    e = hipStreamCreateWithFlags(&eng->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&eng->s_h2d, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&eng->s_d2h, hipStreamNonBlocking);
    if (e == hipSuccess) {
        int least = 0, greatest = 0;                // (the small launch should win a tie for the dispatcher)
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        e = hipStreamCreateWithPriority(&eng->s_side, hipStreamNonBlocking, greatest);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipStreamCreateWithFlags(&eng->s_side, hipStreamNonBlocking); }   // (no priorities here: an ordinary stream does)
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&eng->ev_side_pre, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&eng->ev_side_done, hipEventDisableTiming);
    for (int k = 0; k < 2 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&eng->ev_h2d[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&eng->ev_done, hipEventDisableTiming);
    if (e != hipSuccess) { g_create_error = hipGetErrorString(e); delete eng; return ERR_FAILURE; }
    *out = eng;
    return 0;
}

void cimg_engine_destroy(cimg_engine* e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    e->drain_timing();
    for (EventPair& ev : e->free_events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (DevBuf* b : {&e->descs_enc, &e->descs_dec, &e->recs, &e->layout, &e->scratch, &e->stage_raw, &e->stage_comp, &e->stage_il, &e->dbg, &e->queue, &e->done, &e->sync, &e->next_item, &e->qheads, &e->zstd_seq, &e->zstd_tables, &e->zplan})
        if (b->p) (void)hipFree(b->p);
    for (PinBuf* b : {&e->h_descs, &e->h_descs_dec, &e->h_out, &e->h_dec})
        if (b->p) (void)hipHostFree(b->p);
    (void)hipStreamDestroy(e->stream);
    if (e->s_h2d) { (void)hipStreamSynchronize(e->s_h2d); (void)hipStreamDestroy(e->s_h2d); }
    if (e->s_d2h) { (void)hipStreamSynchronize(e->s_d2h); (void)hipStreamDestroy(e->s_d2h); }
    if (e->s_side) { (void)hipStreamSynchronize(e->s_side); (void)hipStreamDestroy(e->s_side); }
    if (e->ev_side_pre) (void)hipEventDestroy(e->ev_side_pre);
    if (e->ev_side_done) (void)hipEventDestroy(e->ev_side_done);
    for (int k = 0; k < 2; k++) if (e->ev_h2d[k]) (void)hipEventDestroy(e->ev_h2d[k]);
    if (e->ev_done) (void)hipEventDestroy(e->ev_done);
    delete e;
}

const char* cimg_last_error(const cimg_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }
void* cimg_engine_stream(cimg_engine* e) { return (void*)e->stream; }
static int engine_synchronize_main(cimg_engine* e);
int cimg_engine_synchronize(cimg_engine* e)
{
    // (every side launch is joined into the main stream by an event when all goes well; after an error in between it may not be)
    int rc_side = 0;
    if (e->side_used) { rc_side = e->hip(hipStreamSynchronize(e->s_side), "hipStreamSynchronize(side)"); e->side_used = false; }
    const int rc = engine_synchronize_main(e);
    if (rc || rc_side) e->sync_dirty = true;        // a failed wait: the encode launches' counters are not where the host thinks
    return rc ? rc : rc_side;
}
static int engine_synchronize_main(cimg_engine* e)
{
    if (!e->spin_sync) return e->hip(hipStreamSynchronize(e->stream), "hipStreamSynchronize");
    // a batch is a few hundred microseconds of kernels: poll an event instead of sleeping on the stream
    if (!e->sync_ev) {
        int rc = e->hip(hipEventCreateWithFlags(&e->sync_ev, hipEventDisableTiming), "hipEventCreate");
        if (rc) return rc;
    }
    int rc = e->hip(hipEventRecord(e->sync_ev, e->stream), "hipEventRecord");
    if (rc) return rc;
    hipError_t q;
    while ((q = hipEventQuery(e->sync_ev)) == hipErrorNotReady) {}
    return e->hip(q, "hipEventQuery");
}

void cimg_engine_lock(cimg_engine* e) { e->mu.lock(); }
void cimg_engine_unlock(cimg_engine* e) { e->mu.unlock(); }

void* cimg_device_malloc(cimg_engine* e, size_t bytes)
{
    void* p = nullptr;
    (void)hipSetDevice(e->device);
    if (e->hip(hipMalloc(&p, bytes ? bytes : 16), "hipMalloc")) return nullptr;
    return p;
}
void cimg_device_free(cimg_engine* e, void* p)
{
    if (!p) return;
    if (e) (void)hipSetDevice(e->device);
    (void)hipFree(p);
}
void* cimg_host_malloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void cimg_host_free(void* p) { if (p) (void)hipHostFree(p); }
int cimg_memcpy_h2d(cimg_engine* e, void* d, const void* h, size_t n)
{
    int rc = e->hip(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, e->stream), "hipMemcpyAsync(H2D)");
    return rc ? rc : cimg_engine_synchronize(e);
}
int cimg_memcpy_d2h(cimg_engine* e, void* h, const void* d, size_t n)
{
    int rc = e->hip(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, e->stream), "hipMemcpyAsync(D2H)");
    return rc ? rc : cimg_engine_synchronize(e);
}

void cimg_engine_enable_timing(cimg_engine* e, int on) { e->timing_period = on > 0 ? on : 0; e->timing = false; e->batch_no[0] = e->batch_no[1] = 0; }
void cimg_engine_reset_timing(cimg_engine* e)
{
    (void)hipStreamSynchronize(e->stream);
    e->drain_timing();
    for (int k = 0; k < CIMG_K_COUNT; k++) { e->total_ms[k] = 0; e->launches[k] = 0; e->samples[k].clear(); }
}
int cimg_engine_kernel_time(cimg_engine* e, int kernel, double* total_ms, int64_t* launches)
{
    if (kernel < 0 || kernel >= CIMG_K_COUNT) return ERR_INVALID_PARAM;
    int rc = cimg_engine_synchronize(e);
    if (rc) return rc;
    e->drain_timing();
    if (total_ms) *total_ms = e->total_ms[kernel];
    if (launches) *launches = e->launches[kernel];
    return 0;
}

int cimg_engine_kernel_samples(cimg_engine* e, int kernel, float* ms, int max_samples)
{
    if (kernel < 0 || kernel >= CIMG_K_COUNT || max_samples < 0) return ERR_INVALID_PARAM;
    int rc = cimg_engine_synchronize(e);
    if (rc) return rc;
    e->drain_timing();
    const int n = std::min((int)e->samples[kernel].size(), max_samples);
    if (ms && n) memcpy(ms, e->samples[kernel].data(), sizeof(float) * (size_t)n);
    return n;
}
void cimg_engine_decode_stats(cimg_engine* e, int64_t* lean_batches, int64_t* blocks_left_to_general, int64_t* blocks_total, int64_t* zstd_batches)
{
    if (lean_batches) *lean_batches = e->lean_batches;
    if (blocks_left_to_general) *blocks_left_to_general = e->lean_blocks_skipped;
    if (blocks_total) *blocks_total = e->lean_blocks_total;
    if (zstd_batches) *zstd_batches = e->zstd_batches;
}

void cimg_engine_zstd_stats(cimg_engine* e, int64_t* zstd_batches, int64_t* blocks_refused)
{
    if (zstd_batches) *zstd_batches = e->zstd_batches;
    if (blocks_refused) *blocks_refused = e->zstd_blocks_refused;
}

// diagnostics: per-workgroup {shader clock, 100 MHz clock, HW_ID, XCC_ID} x {start, end} of the most recent
// encode (which = 0) or decode (which = 1) launch; 16 uint64 per workgroup (four stamps).  Returns the workgroup count.
void cimg_engine_debug_stamps(cimg_engine* e, int on) { e->stamps = on != 0; }
int cimg_engine_read_stamps(cimg_engine* e, int which, uint64_t* out, int max_workgroups)
{
    if (which < 0 || which > 1 || !e->dbg.p) return 0;
    const int n = e->dbg_count[which] < max_workgroups ? e->dbg_count[which] : max_workgroups;
    if (cimg_memcpy_d2h(e, out, e->dbg.p, (size_t)n * 128)) return ERR_FAILURE;
    return n;
}

static HostCParams to_host(const cimg_cparams* p)
{
    HostCParams h;
    h.typesize = p->typesize; h.clevel = p->clevel; h.blocksize = p->blocksize;
    h.compcode = p->compcode; h.splitmode = p->splitmode;
    memcpy(h.filters, p->filters, 6);
    memcpy(h.filters_meta, p->filters_meta, 6);
    return h;
}

// the kernels of one compress batch, enqueued on the engine's stream; compress_finish() waits and fetches the sizes
// inputs_behind_stream: the pixels are produced by work already enqueued on e->stream (a copy the stream waits for, the
// deinterleave kernel): a launch on another stream has to wait for that too.  The device-resident entry points pass false --
// their caller's pixels are there when the call is made.
static int compress_launch(cimg_engine* e, const cimg_cparams* p, int32_t nchunks,
                           const void* d_raw, const int64_t* raw_off, const int32_t* nbytes,
                           void* d_comp, const int64_t* comp_off, const int32_t* destsize, bool inputs_behind_stream = true)
{
    (void)hipSetDevice(e->device);
    e->claunched = false;
    e->begin_batch(0);
    EncodePlan plan;
    int rc = plan_encode_batch(to_host(p), nchunks, raw_off, nbytes, comp_off, destsize, &plan);
    if (rc < 0) return e->fail(rc, "compress batch rejected by the planner (code %d): codec %d / filter pipeline / block size %d not available on the GPU path",
                               rc, p->compcode, p->blocksize);
    if ((rc = e->reserve(e->recs, sizeof(StreamRec) * (size_t)plan.total_blocks * plan.cp.streams_per_block))) return rc;
    if ((rc = e->reserve(e->layout, sizeof(ChunkLayout) * (size_t)nchunks))) return rc;
    if ((rc = e->reserve(e->h_out, sizeof(ChunkLayout) * ((size_t)nchunks + 1)))) return rc;
    if ((rc = e->reserve(e->scratch, (size_t)plan.total_blocks * plan.cp.slot_bytes + 64))) return rc;
    ChunkLayout* lay_host = nullptr;
    if ((rc = e->device_alias(e->h_out, &lay_host))) return rc;
    // Chunks whose streams all belong to one encode launch are assembled INSIDE that launch (encode_kernel.h: ChunkDesc::assemble,
    // set by the planner); the two assembly kernels run behind the launches only when a chunk is left for them (memcpyed up front,
    // or full blocks split into planes plus an unsplit leftover block).
    const bool fold = !e->no_fold;
    bool leftovers = false;                         // some chunk is NOT assembled in a launch
    if (!fold) for (ChunkDesc& d : plan.descs) d.assemble = 0;
    for (const ChunkDesc& d : plan.descs) if (!d.assemble) leftovers = true;
    const size_t desc_bytes = sizeof(ChunkDesc) * (size_t)nchunks;
    if ((rc = e->upload_descs(e->descs_enc, e->shadow_enc, e->h_descs, plan.descs.data(), desc_bytes))) return rc;
    const size_t sync_words = 16 + 2 * (size_t)nchunks;
    if (sync_words * 4 > e->sync.cap || (size_t)nchunks > e->sync_chunks) e->sync_dirty = true;
    if ((rc = e->reserve(e->sync, sync_words * 4))) return rc;
    if (++e->fold_gen == 0) { e->fold_gen = 1; e->sync_dirty = true; }
    // the encode launches' work-queue heads (encode_kernel.h, "the work queue"): per kind of launch (split / unsplit) two sets of
    // ENC_NQ heads; a launch pops one set and zeroes the other for the next launch of its kind
    const size_t qset_bytes = (size_t)ENC_NQ * ENC_QSTRIDE * 4;
    if (4 * qset_bytes > e->qheads.cap) e->sync_dirty = true;
    if ((rc = e->reserve(e->qheads, 4 * qset_bytes))) return rc;
    if (e->sync_dirty) {
        if ((rc = e->hip(hipMemsetAsync(e->qheads.p, 0, e->qheads.cap, e->stream), "queue heads memset"))) return rc;
        if ((rc = e->hip(hipMemsetAsync(e->sync.p, 0, e->sync.cap, e->stream), "sync memset"))) return rc;
        e->qpar[0] = e->qpar[1] = 0;
        e->sync_chunks = (e->sync.cap / 4 - 16) / 2;
        e->sync_dirty = false;
    }
    ((ChunkLayout*)e->h_out.p)[nchunks].cbytes = 0;   // turns negative when a wave gave up waiting for a chunk (compress_finish)
    e->claunched = true;                          // from here on kernels may be in flight and h_out is this batch's
    e->sync_dirty = true;                         // until the launches below are all enqueued: an error in between leaves the counters unknown

    // Two launches in a batch -- the byte planes of the full blocks, and the unsplit leftover block that every chunk of an image
    // has whose row size does not divide 4 MiB (c-blosc2 never splits a chunk's last, shorter block) -- do not wait for each other:
    // the small one goes FIRST, on a stream of its own, and the large one's persistent chains start beside it (its work queue
    // absorbs the few CUs that come free a little later).  One behind the other the small launch cost a whole item's latency at
    // the end of the batch (configs[1]'s pixels in chunks of 4 MiB + 4 KiB: 471 us a batch, of which 60 for 31 leftover blocks).
    // The launches share nothing but read-only inputs: queue heads, item lists and scratch slots are per launch / per block.
    const bool side = plan.lds_split && plan.lds_unsplit && !e->no_side && plan.cp.compcode != CODEC_ZSTD;   // (zstd: both launches would grow one sequence buffer)
    // The small launch is ordered behind everything the engine's stream holds at this point -- descriptors uploaded or counters
    // cleared for this batch, and ANY producer of the pixels enqueued there before this call (cimg_deinterleave_device, a caller's
    // own kernel on cimg_engine_stream(), a decode batch begun and not fetched): the stream's contract is "in order", and a launch
    // that leaves it must not break that.  (Round 3 skipped the wait when this batch itself had put nothing on the stream: an
    // image deinterleaved on the device and compressed right behind it had its leftover blocks encoded from planes that were not
    // written yet.)  With an idle stream -- the steady state of a synchronous caller -- the event is complete when it is recorded.
    if (side && (rc = e->side_follows_stream())) return rc;
    size_t next_item_used = 0;                        // (the launches of a batch get regions of their own in next_item)
    if ((rc = e->reserve(e->next_item, sizeof(int32_t) * ((size_t)plan.total_blocks * (size_t)(plan.cp.streams_per_block + 1) + 64)))) return rc;
    for (int pass = 0; pass < 2; pass++) {
        const int split = side ? pass : 1 - pass;     // side by side: the small (unsplit) launch first
        int lds_bytes = split ? plan.lds_split : plan.lds_unsplit;
        if (!lds_bytes) continue;                     // no blocks of that kind in the batch
        const bool rt = e->enc_rt && (plan.cp.compcode == CODEC_LZ4 || plan.cp.compcode == CODEC_LZ4HC);
        if (rt) lds_bytes = encode_lds_bytes_rt(lds_bytes - LZ4_HASH_BYTES);   // the plane and its margin: the table lives in registers
        hipStream_t const on = (side && !split) ? e->s_side : e->stream;
        // Split launch: whole blocks as work items -- each block read from HBM ONCE instead of once per byte plane -- when
        // the geometry allows it AND the batch is large.  A block item is `typesize` times coarser than a plane item, and
        // on a small batch the coarser granularity costs more than the second read saves (HBM is a few per cent utilised;
        // measured on 4 x 4096^2 float16 = 3.2 blocks per resident wave: 537 us against 512 us; from 8 rounds on the tail
        // is noise).  CIMG_ENC_BLOCK_ITEMS=1 / 0 forces the choice.
        bool block_items_ok = split != 0 && !rt;    // (the register-table kernel has no registers left for planes that wait)
        if (block_items_ok)
            for (const ChunkDesc& d : plan.descs)
                if (!d.memcpyed && d.split && !encode_block_items_ok(plan.cp.typesize, plan.cp.filter, d.blocksize)) { block_items_ok = false; break; }
        // how many of the blocks go out whole: all of them on a large batch; on a small one the rounds every chain takes anyway,
        // while the END of the launch stays plane by plane, most significant planes first (encode_item_place): a coarser item would
        // cost a whole extra item on the slowest chain, and the light planes (noise: 12 us against 77 for a coded one) are what idle
        // chains fill the tail with.  When the last round is short (3.2 rounds on configs[1]) the full round in front of it goes plane
        // by plane too: 4096 blocks on 1280 chains -> 2560 whole blocks, 1536 plane by plane: 369 -> 360 us (3 / 2 / 1 / 0 whole
        // rounds: 369 / 360 / 366 / 380 -- a plane item reads its whole block, and costs a pop).
        int whole_blocks = 0;
        if (block_items_ok) {
            const int chains = std::max(1, e->lds_per_cu / std::max(lds_bytes, 1)) * e->num_cus;
            const double rounds = (double)plan.total_blocks / chains;
            if (e->enc_block_items == 1 || (e->enc_block_items < 0 && rounds >= 8.0)) whole_blocks = plan.total_blocks;
            else if (e->enc_block_items < 0 && e->enc_hybrid) whole_blocks = std::min(plan.total_blocks, std::min(std::max(0, (int)(rounds - 0.5)), e->enc_whole_rounds) * chains);   // the full rounds, less the one in front of a short last round (CIMG_ENC_WHOLE_ROUNDS: at most so many)
        }
        const int items = encode_items(plan.total_blocks, plan.cp.streams_per_block, split != 0, whole_blocks);
        uint64_t* dbg = nullptr;
        if (e->stamps && split) {       // diagnostics: 16 uint64 per item (start / end stamps; -DCIMG_PROFILE builds: cycle accounting)
            if ((rc = e->reserve(e->dbg, (size_t)items * 128))) return rc;
            if ((rc = e->hip(hipMemsetAsync(e->dbg.p, 0, (size_t)items * 128, e->stream), "dbg memset"))) return rc;
            dbg = (uint64_t*)e->dbg.p;
            e->dbg_count[0] = items;
        }
        uint32_t* const sync = (uint32_t*)e->sync.p;
        uint32_t* const qsets = (uint32_t*)e->qheads.p + (size_t)(split ? 0 : 2) * ENC_NQ * ENC_QSTRIDE;
        uint32_t* const head = qsets + (size_t)(e->qpar[split] & 1) * ENC_NQ * ENC_QSTRIDE;
        uint32_t* const head_next = qsets + (size_t)((e->qpar[split] + 1) & 1) * ENC_NQ * ENC_QSTRIDE;
        const size_t nslots = e->sync_chunks;
        int32_t* const next_item = (int32_t*)e->next_item.p + next_item_used;
        next_item_used += (size_t)items + 16;
        EncodeArgs ea{(const ChunkDesc*)e->descs_enc.p, nchunks, plan.cp, (const uint8_t*)d_raw, (uint8_t*)e->scratch.p,
                      (StreamRec*)e->recs.p, lds_bytes, plan.total_blocks, split, dbg, head, plan.uniform_nblocks, whole_blocks,
                      nullptr, 0, nullptr,
                      head_next, 0, fold ? 1 : 0, (uint8_t*)d_comp, (ChunkLayout*)e->layout.p, lay_host,
                      sync + 16, sync + 16 + nslots, next_item, e->fold_gen};
        const bool blz = plan.cp.compcode == CODEC_BLOSCLZ, zst = plan.cp.compcode == CODEC_ZSTD;
        void (*const enc_kernel)(EncodeArgs) = blz ? cimg_encode_streams_blosclz : zst ? cimg_encode_streams_zstd : rt ? cimg_encode_streams_rt : cimg_encode_streams;
        const int lds_slot = blz ? 3 : zst ? 4 : rt ? 11 : 0;
        // persistent chains, as many as are resident at once and never more than there are items; ganged into workgroups so
        // that the 1280-byte LDS granules of a CU come out even (encode_gang above)
        if (e->enc_wgs_lds[split] != lds_bytes || e->enc_wgs_codec != plan.cp.compcode) {
            int waves_cu = 0;
            e->enc_wgs_lds[0] = e->enc_wgs_lds[1] = -1;
            e->enc_wgs_codec = plan.cp.compcode;
            if ((rc = e->hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&waves_cu, enc_kernel, 64, 0), "occupancy query"))) return rc;
            e->enc_waves_cu = waves_cu > 0 ? waves_cu : 1;           // what the registers allow
            e->enc_wgs_lds[split] = lds_bytes;
        }
        const auto wgs_for = [&](int gang) {                         // resident workgroups per CU for a gang size
            const int granules = (gang * lds_bytes + LDS_GRANULE - 1) / LDS_GRANULE;
            return std::min(e->lds_per_cu / LDS_GRANULE / std::max(granules, 1), std::max(e->enc_waves_cu / gang, 0));
        };
        int gang = 1;
        if (e->enc_gang > 0) {                                       // diagnostic: forced, as far as one workgroup's LDS goes
            gang = std::min(e->enc_gang, (int)CIMG_ENC_GANG_MAX);
            while (gang > 1 && (gang * lds_bytes > e->lds_per_wg || wgs_for(gang) < 1)) --gang;
        }
        else if (items > wgs_for(1) * e->num_cus)                    // a small batch spreads single waves over the CUs
            for (int g = 2; g <= CIMG_ENC_GANG_MAX; ++g)
                if (g * lds_bytes <= e->lds_per_wg && g * wgs_for(g) > gang * wgs_for(gang)) gang = g;
        if (gang > 1 && e->allow_lds(enc_kernel, lds_slot, gang * lds_bytes)) gang = 1;   // the runtime refused that much LDS for one workgroup
        if ((rc = e->allow_lds(enc_kernel, lds_slot, gang * lds_bytes))) return rc;
        int per_cu_use = std::max(wgs_for(gang), 1);
        if (e->enc_wgs_limit > 0) per_cu_use = std::max(1, std::min(per_cu_use, e->enc_wgs_limit));   // diagnostic: fewer resident workgroups
        const int grid = std::min((items + gang - 1) / gang, per_cu_use * e->num_cus);
        if (e->verbose && (e->enc_said_lds != lds_bytes || e->enc_said_gang != gang)) {
            e->enc_said_lds = lds_bytes; e->enc_said_gang = gang;
            fprintf(stderr, "[cimg] encode launch: %d bytes LDS per chain, gangs of %d -> %d workgroup(s) per CU = %d chains per CU x %d CUs (grid %d)\n",
                    lds_bytes, gang, per_cu_use, gang * per_cu_use, e->num_cus, grid);
        }
        if (zst) {
            // every wave of the launch owns room for the sequences of the largest stream (a sequence covers at least four bytes)
            const int stride = 2 * (plan.cp.max_blocksize / 4 + 64);
            if ((rc = e->reserve(e->zstd_seq, sizeof(uint32_t) * (size_t)stride * (size_t)grid * (size_t)gang))) return rc;
            if (!e->zstd_tables.p) {
                ZstdEncTables t;
                zstd_build_enc_tables(&t);
                if ((rc = e->reserve(e->zstd_tables, sizeof(t)))) return rc;
                if ((rc = e->hip(hipMemcpyAsync(e->zstd_tables.p, &t, sizeof(t), hipMemcpyHostToDevice, e->stream), "zstd tables H2D"))) return rc;
                if ((rc = e->hip(hipStreamSynchronize(e->stream), "zstd tables H2D"))) return rc;     // (t lives on this stack frame)
            }
            ea.zstd_seq = (uint32_t*)e->zstd_seq.p; ea.zstd_seq_stride = stride; ea.zstd_tables = (const ZstdEncTables*)e->zstd_tables.p;
        }
        ea.nwaves = grid * gang;
        if ((rc = e->launch(zst ? CIMG_K_ENCODE_ZSTD : CIMG_K_ENCODE, enc_kernel, ea, grid, 64 * gang, gang * lds_bytes, on))) return rc;
        if (side && !split) {
            // everything that follows on the main stream behind the large launch -- the assembly kernels, the host's wait -- also
            // waits for the small one
            if ((rc = e->hip(hipEventRecord(e->ev_side_done, e->s_side), "event record"))) return rc;
        }
        e->qpar[split] ^= 1;                          // the next launch of this kind pops the heads this one zeroes
    }
    if (side && (rc = e->hip(hipStreamWaitEvent(e->stream, e->ev_side_done, 0), "stream wait"))) return rc;
    if (leftovers) {
        AssembleArgs aa{(const ChunkDesc*)e->descs_enc.p, nchunks, plan.cp, (const uint8_t*)d_raw, (const uint8_t*)e->scratch.p,
                        (const StreamRec*)e->recs.p, (uint8_t*)d_comp, (ChunkLayout*)e->layout.p, plan.uniform_nblocks, lay_host, 1};
        if ((rc = e->launch(CIMG_K_LAYOUT, cimg_layout_chunks, aa, nchunks, 64, 0))) return rc;
        if ((rc = e->launch(CIMG_K_EMIT, cimg_emit_blocks, aa, plan.total_blocks, 256, 0))) return rc;
    }
    e->sync_dirty = false;                        // everything enqueued: the counters end where the bases say
    return 0;
}

static int compress_finish(cimg_engine* e, int32_t nchunks, int32_t* cbytes)
{
    int rc;
    if ((rc = cimg_engine_synchronize(e))) return rc;
    if (!e->claunched) return 0;                  // the batch was rejected before anything ran: there are no sizes to read
    const ChunkLayout* lay = (const ChunkLayout*)e->h_out.p;
    if (lay[nchunks].cbytes < 0) {                // a wave of the encode launch waited in vain for a chunk to be laid out
        e->sync_dirty = true;
        return e->fail(ERR_FAILURE, "encode launch: a chunk was never published for assembly (in-launch hand-over timed out)");
    }
    for (int i = 0; i < nchunks; i++) cbytes[i] = lay[i].cbytes;
    return 0;
}

int cimg_compress_batch_device(cimg_engine* e, const cimg_cparams* p, int32_t nchunks,
                               const void* d_raw, const int64_t* raw_off, const int32_t* nbytes,
                               void* d_comp, const int64_t* comp_off, const int32_t* destsize, int32_t* cbytes)
{
    std::lock_guard<std::recursive_mutex> lock_(e->mu);
    if (nchunks <= 0) return 0;
    if (!p || !raw_off || !nbytes || !comp_off || !destsize || !cbytes) return e->fail(ERR_INVALID_PARAM, "null argument");
    e->cflight_chunks = -1;
    // (pixels a decode batch of this engine is still writing -- begun, not fetched -- are "behind the stream")
    const int rc = compress_launch(e, p, nchunks, d_raw, raw_off, nbytes, d_comp, comp_off, destsize, e->dflight_open);
    return rc ? rc : compress_finish(e, nchunks, cbytes);
}

int cimg_compress_batch_device_begin(cimg_engine* e, const cimg_cparams* p, int32_t nchunks,
                                     const void* d_raw, const int64_t* raw_off, const int32_t* nbytes,
                                     void* d_comp, const int64_t* comp_off, const int32_t* destsize)
{
    std::lock_guard<std::recursive_mutex> lock_(e->mu);
    e->cflight_chunks = -1;
    if (nchunks <= 0) { e->cflight_chunks = 0; return 0; }
    if (!p || !raw_off || !nbytes || !comp_off || !destsize) return e->fail(ERR_INVALID_PARAM, "null argument");
    const int rc = compress_launch(e, p, nchunks, d_raw, raw_off, nbytes, d_comp, comp_off, destsize, e->dflight_open);
    if (!rc) e->cflight_chunks = nchunks;
    return rc;
}

int cimg_compress_batch_device_fetch(cimg_engine* e, int32_t nchunks, int32_t* cbytes)
{
    std::lock_guard<std::recursive_mutex> lock_(e->mu);
    if (e->cflight_chunks < 0 || e->cflight_chunks != nchunks)
        return e->fail(ERR_INVALID_PARAM, "no compress batch of %d chunks is in flight (cimg_compress_batch_device_begin comes first)", nchunks);
    e->cflight_chunks = -1;
    if (nchunks == 0) return 0;
    if (!cbytes) return e->fail(ERR_INVALID_PARAM, "null argument");
    return compress_finish(e, nchunks, cbytes);
}

// the kernels of one decode batch, enqueued on the engine's stream; decompress_finish() waits, launches the general
// kernel late if the lean one left blocks behind, and collects the status words
static int decompress_launch(cimg_engine* e, int32_t nchunks, const void* d_comp, const int64_t* comp_off,
                             const int32_t* nbytes, const int32_t* blocksize, void* d_raw, const int64_t* raw_off,
                             const int32_t* comp_size = nullptr, bool inputs_behind_stream = true)
{
    (void)hipSetDevice(e->device);
    e->dflight.launched = false;                  // a rejected batch must not be finished on the previous batch's state
    e->dflight.lean_grid = 0;
    e->dflight.nchunks = 0;
    e->begin_batch(1);
    DecodePlan plan;
    int rc = plan_decode_batch(nchunks, comp_off, nbytes, blocksize, raw_off, &plan, comp_size);
    if (rc < 0) return e->fail(rc, "decompress batch rejected by the planner (code %d)", rc);
    if (plan.lds_lean > 0) plan.lds_lean += e->lean_lds_pad;
    const size_t desc_bytes = sizeof(ChunkDesc) * (size_t)nchunks;
    const size_t st_bytes = sizeof(int32_t) * (size_t)nchunks;
    if ((rc = e->upload_descs(e->descs_dec, e->shadow_dec, e->h_descs_dec, plan.descs.data(), desc_bytes))) return rc;
    // (behind the status words: one word per wave of the lean launch -- at most a wave per block --, the blocks it left over)
    if ((rc = e->reserve(e->h_dec, st_bytes + 32 + sizeof(uint32_t) * (size_t)plan.total_blocks))) return rc;
    int32_t* st_dev = nullptr;                    // the status words live in pinned host memory; only failing blocks write
    if ((rc = e->device_alias(e->h_dec, &st_dev))) return rc;
    memset(e->h_dec.p, 0, st_bytes);
    uint64_t* dbg = nullptr;
    if (e->stamps) {
        if ((rc = e->reserve(e->dbg, (size_t)plan.total_blocks * 128))) return rc;
        if ((rc = e->hip(hipMemsetAsync(e->dbg.p, 0, (size_t)plan.total_blocks * 128, e->stream), "dbg memset"))) return rc;
        dbg = (uint64_t*)e->dbg.p;
        e->dbg_count[1] = plan.total_blocks;
    }
    // Lean launch first (blocks with at most one LZ4-coded plane, 2x the residency), the general kernel behind it
    // for whatever it left.  A batch the lean kernel mostly skips (e.g. every plane LZ4-coded) switches it off for
    // the next 16 batches; the skipped count comes back through the pinned status area.
    const bool stamping = e->stamps;
    const bool lean = e->lean_hold == 0 && plan.lds_lean > 0 && plan.lds_lean < plan.lds_bytes;
    if (e->lean_hold > 0 && e->lean_hold < (1 << 30)) e->lean_hold--;
    uint32_t* done = nullptr;
    uint32_t* skipped_dev = nullptr;
    volatile uint32_t* skipped_host = nullptr;
    if (lean) {
        const size_t done_bytes = sizeof(uint32_t) * (size_t)plan.total_blocks;
        if (done_bytes > e->done.cap) {
            if ((rc = e->reserve(e->done, done_bytes))) return rc;
            if ((rc = e->hip(hipMemsetAsync(e->done.p, 0, e->done.cap, e->stream), "done memset"))) return rc;
            e->done_gen = 0;
        }
        if (++e->done_gen == 0) {                  // wrapped: stale stamps could match again
            if ((rc = e->hip(hipMemsetAsync(e->done.p, 0, e->done.cap, e->stream), "done memset"))) return rc;
            e->done_gen = 1;
        }
        done = (uint32_t*)e->done.p;
        skipped_host = (volatile uint32_t*)((uint8_t*)e->h_dec.p + ((st_bytes + 15) & ~(size_t)15));
        skipped_dev = (uint32_t*)((uint8_t*)st_dev + ((st_bytes + 15) & ~(size_t)15));
    }
    // What the lean kernel leaves by GEOMETRY is known up front: the leftover (last, shorter, never split) block of every chunk
    // -- one per chunk for every image whose row size does not divide 4 MiB.  The general kernel is launched over exactly them
    // (one workgroup per chunk instead of one per block), and the lean kernel does not count them among the blocks it reports
    // as left over.
    int known_left = 0;
    if (lean) for (const ChunkDesc& d : plan.descs) if (d.leftover && !d.memcpyed) known_left++;
    EventPair ev{};
    const bool timed = e->timing;
    if (timed) { ev = e->get_events(); (void)hipEventRecord(ev.a, e->stream); e->timing = false; }   // lean + general = ONE timed decode
    // The leftover blocks' launch may go to the side stream (below): it reads descs_dec and done[] like every decode launch, so it
    // is ordered behind everything the main stream holds HERE -- uploads and memsets of this batch, producers of the chunks --
    // and runs beside the lean launch that follows.  (Round 3 launched it unordered: a batch whose geometry differs from the one
    // before could read the previous batch's descriptors.)
    bool side_ordered = false;
    if (lean && known_left && e->lean_last_skipped == 0 && !inputs_behind_stream && !e->no_side && !timed) {
        if ((rc = e->side_follows_stream())) return rc;
        side_ordered = true;
    }
    if (lean) {
        // persistent waves, as many as are resident at once (registers: two per SIMD; LDS: 1280-byte granules per workgroup),
        // never more than there are blocks; wave w walks blocks w, w + G, ...
        if (!(rc = e->allow_lds(cimg_decode_lean, 2, plan.lds_lean))) {
            if (e->lean_wgs_lds != plan.lds_lean) {
                int per_cu = 0;
                if ((rc = e->hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cimg_decode_lean, 64, (size_t)plan.lds_lean), "occupancy query"))) return rc;
                const int granules = (plan.lds_lean + LDS_GRANULE - 1) / LDS_GRANULE;
                e->lean_wgs_cu = std::max(1, std::min(per_cu, e->lds_per_cu / LDS_GRANULE / std::max(granules, 1)));
                e->lean_wgs_lds = plan.lds_lean;
                if (e->verbose) fprintf(stderr, "[cimg] lean decode launch: %d bytes LDS -> %d persistent waves per CU (occupancy query %d)\n", plan.lds_lean, e->lean_wgs_cu, per_cu);
            }
            int per_cu_use = e->lean_wgs_cu;
            if (e->lean_wgs_limit > 0) per_cu_use = std::max(1, std::min(per_cu_use, e->lean_wgs_limit));
            // Up to three rounds the launch is persistent (a wave walks blocks w, w + G, ... with the next blocks' loads in flight);
            // a larger batch gets one workgroup per block, which the hardware deals out as slots fall free: with a fixed stride
            // the launch waits for the waves that drew the slow lots (1 GiB, 16 blocks per wave: 495 against 446 us), and a queue
            // popped per block serialises on its one address (8192 device-scope atomics inside 60 us: 152 against 68 us).
            const int resident = per_cu_use * e->num_cus;
            const int grid = plan.total_blocks > 3 * resident ? plan.total_blocks : std::min(plan.total_blocks, resident);
            memset((void*)skipped_host, 0, sizeof(uint32_t) * (size_t)grid);        // a wave only writes its word if it left blocks over
            e->dflight.lean_grid = grid;
            DecodeArgs la{(const ChunkDesc*)e->descs_dec.p, nchunks, (const uint8_t*)d_comp, (uint8_t*)d_raw, st_dev, plan.lds_lean, dbg,
                          plan.uniform_nblocks, done, e->done_gen, skipped_dev, plan.total_blocks, known_left ? 1 : 0, 1, e->lean_tune};   // (blk_first here: leftover blocks are not counted)
            rc = e->launch(CIMG_K_DECODE, cimg_decode_lean, la, grid, 64, plan.lds_lean);
        }
    }
    // The general kernel goes right behind the lean one -- unless the previous lean batch left it nothing to do: then
    // it is only launched (and waited for) if the skipped count that comes back says a block is still undecoded.
    const bool general_now = !lean || e->lean_last_skipped != 0;
    // (diagnostic stamps go to the lean launch when there is one: both would write the same slots)
    DecodeArgs da{(const ChunkDesc*)e->descs_dec.p, nchunks, (const uint8_t*)d_comp, (uint8_t*)d_raw, st_dev, plan.lds_bytes, (lean && stamping) ? nullptr : dbg,
                  plan.uniform_nblocks, done, e->done_gen, nullptr, plan.total_blocks, 0, 1, 0};
    if (!rc && general_now) {
        if (!(rc = e->allow_lds(cimg_decode_blocks, 1, plan.lds_bytes)))
            rc = e->launch(CIMG_K_DECODE, cimg_decode_blocks, da, plan.total_blocks, 256, plan.lds_bytes);
    } else if (!rc && known_left) {
        // (beside the lean launch, on the side stream, when the chunks are there already: its few workgroups find room as the lean
        // launch's waves retire -- behind it on the same stream they would run alone on an idle, clocked-down device)
        const bool beside = !inputs_behind_stream && !e->no_side && !timed;
        DecodeArgs dk = da;
        dk.blk_first = 0; dk.blk_step = 0;                  // workgroup k: the last block of chunk k (a chunk without a leftover block: done already)
        // (the side stream follows the main one up to the point right in front of the lean launch -- ev_side_pre, recorded there:
        // this batch's descriptor upload and its `done` / stamp memsets lie before it, and so does whatever produced the chunks;
        // what the side stream gains is running BESIDE the lean launch)
        if (beside && !side_ordered) rc = e->side_follows_stream();              // (no lean launch in front after all: plain order)
        if (!rc && !(rc = e->allow_lds(cimg_decode_blocks, 1, plan.lds_bytes)))
            rc = e->launch(CIMG_K_DECODE, cimg_decode_blocks, dk, nchunks, 256, plan.lds_bytes, beside ? e->s_side : e->stream);
        if (!rc && beside) {
            if (!(rc = e->hip(hipEventRecord(e->ev_side_done, e->s_side), "event record")))
                rc = e->hip(hipStreamWaitEvent(e->stream, e->ev_side_done, 0), "stream wait");
        }
    }
    if (timed) { e->timing = true; (void)hipEventRecord(ev.b, e->stream); e->pending[CIMG_K_DECODE].push_back(ev); }
    e->dflight.lean = lean; e->dflight.general_now = general_now; e->dflight.timed = timed; e->dflight.known_left = known_left;
    e->dflight.nchunks = nchunks; e->dflight.total_blocks = plan.total_blocks; e->dflight.lds_bytes = plan.lds_bytes;
    e->dflight.st_bytes = st_bytes; e->dflight.da = da;
    e->dflight.max_blocksize = 0;
    for (const ChunkDesc& d : plan.descs) e->dflight.max_blocksize = std::max(e->dflight.max_blocksize, (int32_t)d.blocksize);
    e->dflight.launched = true;
    return rc;
}

static int decompress_finish(cimg_engine* e, int32_t* status)
{
    const cimg_engine::DecodeFlight& f = e->dflight;
    if (!f.launched) return cimg_engine_synchronize(e);          // nothing of this batch was enqueued (planner / allocation failure)
    const bool lean = f.lean, general_now = f.general_now, timed = f.timed;
    const int32_t nchunks = f.nchunks;
    const DecodeArgs da = f.da;
    struct { int32_t total_blocks, lds_bytes; } plan{f.total_blocks, f.lds_bytes};
    volatile uint32_t* skipped_host = (volatile uint32_t*)((uint8_t*)e->h_dec.p + ((f.st_bytes + 15) & ~(size_t)15));
    int rc;
    if ((rc = cimg_engine_synchronize(e))) return rc;
    if (lean) {
        uint32_t skipped = 0;
        for (int w = 0; w < f.lean_grid; ++w) skipped += skipped_host[w];
        // (leftover blocks the geometry announced are not in this count: the general kernel has decoded them already, launched
        // over every block or over exactly them -- what counts here is what the lean kernel left BEYOND those)
        e->lean_batches++; e->lean_blocks_skipped += skipped + (uint32_t)f.known_left; e->lean_blocks_total += plan.total_blocks;
        e->lean_last_skipped = skipped;
        if ((int64_t)skipped * 4 > plan.total_blocks) e->lean_hold = 16;
        if (e->verbose) fprintf(stderr, "[cimg] decode: lean kernel left %u of %d blocks to the general kernel%s\n", skipped, plan.total_blocks,
                                            (!general_now && skipped) ? " (launched late)" : "");
        if (!general_now && skipped) {
            // (timed batches: this launch gets its own event pair, added to the decode total without counting a second launch)
            if ((rc = e->allow_lds(cimg_decode_blocks, 1, plan.lds_bytes))) return rc;
            EventPair ev2{};
            if (timed) { ev2 = e->get_events(); (void)hipEventRecord(ev2.a, e->stream); e->timing = false; }
            rc = e->launch(CIMG_K_DECODE, cimg_decode_blocks, da, plan.total_blocks, 256, plan.lds_bytes);
            if (timed) { e->timing = true; (void)hipEventRecord(ev2.b, e->stream); e->pending_extra.push_back(ev2); }
            if (rc) return rc;
            if ((rc = cimg_engine_synchronize(e))) return rc;
        }
    }
    int32_t* st = (int32_t*)e->h_dec.p;
    // zstd chunks (codec format 4) came back with the internal word STATUS_ZSTD_PENDING: they have a decoder of their own -- one
    // wave per block, only ever launched here.  Exactly those words are cleared (a chunk of this engine's codecs with a filter
    // pipeline nobody reads keeps its ERR_CODEC_SUPPORT), cimg_decode_zstd goes over the batch, and the words are read once more.
    std::vector<int> unread_chunks[2];                      // [1]: split chunks
    for (int i = 0; i < nchunks; i++) if (st[i] == STATUS_ZSTD_PENDING || st[i] == STATUS_ZSTD_PENDING_SPLIT) { unread_chunks[st[i] == STATUS_ZSTD_PENDING_SPLIT].push_back(i); st[i] = 0; }
    bool launched = false;
    const bool any_zstd = !unread_chunks[0].empty() || !unread_chunks[1].empty();
    // (one event pair around whatever the read path launches: CIMG_K_DECODE_ZSTD is the time of the whole path)
    EventPair zev{};
    const bool ztimed = any_zstd && e->timing;
    if (ztimed) { zev = e->get_events(); (void)hipEventRecord(zev.a, e->stream); }
    // The fused kernels (one launch, output + stage + tables in LDS together): everything when CIMG_ZSTD_FUSED=1, and the blocks the
    // walk refused.  (Blocks so large that two sets of tables do not fit a workgroup's LDS beside them: the one-wave launch reads
    // every kind.)  only_refused: DecodeArgs::zplan is set and the kernels read the blocks whose plan says ZPLAN_FALLBACK.
    auto fused = [&](const DecodeArgs& base, int blk_first, int nblocks) -> int {
        const bool two_waves = !unread_chunks[1].empty() && zstd_kernel_lds_bytes(f.max_blocksize, 2) <= e->lds_per_wg;
        for (int kind = 0; kind < 2; kind++) {
            const bool has = kind ? (two_waves && !unread_chunks[1].empty()) : (!unread_chunks[0].empty() || (!two_waves && !unread_chunks[1].empty()));
            if (!has) continue;
            DecodeArgs za = base;
            const int waves = kind ? 2 : 1;
            za.lds_bytes = zstd_kernel_lds_bytes(f.max_blocksize, waves);
            za.blk_first = blk_first;
            za.tune = two_waves ? 1 : 0;                        // 1: each launch reads the chunks of its own kind only
            if (za.lds_bytes > e->lds_per_wg) {
                // blocks too large for one workgroup's LDS (behind a walk that refused blocks nobody knows whose they are: the call fails)
                if (base.zplan) return e->fail(ERR_CODEC_SUPPORT, "zstd blocks of %d bytes: a plan did not fit and the blocks are too large for cimg_decode_zstd", (int)f.max_blocksize);
                for (int k2 = 0; k2 < 2; k2++) for (int i : unread_chunks[k2]) if (kind == k2 || !two_waves) st[i] = ERR_CODEC_SUPPORT;
            } else if (kind) {
                if ((rc = e->allow_lds(cimg_decode_zstd_split, 6, za.lds_bytes))) return rc;
                if ((rc = e->launch(CIMG_K_ZSTD_FUSED, cimg_decode_zstd_split, za, nblocks, 128, za.lds_bytes))) return rc;
                launched = true;
            } else {
                if ((rc = e->allow_lds(cimg_decode_zstd, 5, za.lds_bytes))) return rc;
                if ((rc = e->launch(CIMG_K_ZSTD_FUSED, cimg_decode_zstd, za, nblocks, 64, za.lds_bytes))) return rc;
                launched = true;
            }
        }
        return 0;
    };
    auto read_path = [&]() -> int {
        if (!any_zstd) return 0;
        DecodeArgs zb = da;
        zb.dbg = nullptr; zb.done = nullptr; zb.skipped = nullptr; zb.zplan = nullptr;
        const int replay_lds = zstd_replay_lds_bytes(f.max_blocksize);
        if (e->zstd_fused || replay_lds > e->lds_per_wg) {
            if ((rc = fused(zb, 0, plan.total_blocks))) return rc;
        } else {
            const int area = zstd_kernel_area(f.max_blocksize);
            const int cap = e->zstd_plan_cap > 0 ? e->zstd_plan_cap : area;
            const int lanes = std::max(0, std::min(e->zstd_lanes, std::min(64, (e->lds_per_wg - 64 - (int)ZSTD_SEQ_CODES_BYTES) / (int)ZSTD_SEQ_LANE_BYTES)));
            const int64_t stride = zstd_plan_stride(cap, lanes > 0);
            int group = (int)std::max<int64_t>(1, std::min<int64_t>(plan.total_blocks, e->zstd_plan_bytes / stride));
            // the plans are a convenience, not a requirement: when the device cannot spare the memory the groups shrink, and below
            // 64 blocks a group the fused kernel -- which needs no plan -- reads everything
            while ((rc = e->zstd_plan_fail ? -4 /* BLOSC2_ERROR_MEMORY_ALLOC */ : e->reserve(e->zplan, (size_t)group * (size_t)stride))) {
                (void)hipGetLastError();
                if (group <= 64) break;
                group /= 2;
            }
            if (rc) {
                if (e->verbose) fprintf(stderr, "[cimg] zstd: no device memory for plans (%lld bytes a block): cimg_decode_zstd reads the batch\n", (long long)stride);
                if ((rc = fused(zb, 0, plan.total_blocks))) return rc;
                return 0;
            }
            volatile uint32_t* refused = skipped_host;           // (the lean launch's words have been read: the first one counts refused plans now)
            zb.skipped = (uint32_t*)((uint8_t*)da.status + ((f.st_bytes + 15) & ~(size_t)15));
            zb.zplan = (uint8_t*)e->zplan.p; zb.zplan_stride = stride; zb.zcap = cap; zb.zarea = area; zb.zlanes = lanes;
            if (lanes > 0 && (rc = e->allow_lds(cimg_zstd_seq, 9, zstd_seq_lds_bytes(lanes)))) return rc;
            if (lanes > 0 && (rc = e->allow_lds(cimg_zstd_lit, 10, zstd_lit_lds_bytes()))) return rc;
            if ((rc = e->allow_lds(cimg_zstd_walk, 7, zstd_walk_lds_bytes(e->zstd_walk_stage)))) return rc;
            if ((rc = e->allow_lds(cimg_zstd_replay, 8, replay_lds))) return rc;
            for (int g0 = 0; g0 < plan.total_blocks; g0 += group) {
                const int nb = std::min(group, plan.total_blocks - g0);
                *refused = 0;
                // one word per block behind the counter: ZFALL_PENDING from the replay for a block it left to cimg_decode_zstd,
                // ZFALL_DONE from that kernel (the lean launch's words, which lived here, have been read)
                volatile uint32_t* const fall = refused + 1;
                for (int b = g0; b < g0 + nb; b++) fall[b] = 0;
                DecodeArgs wa = zb;
                wa.blk_first = g0; wa.lds_bytes = zstd_walk_lds_bytes(e->zstd_walk_stage);
                if ((rc = e->launch(CIMG_K_ZSTD_WALK, cimg_zstd_walk, wa, nb, 64, wa.lds_bytes))) return rc;
                if (lanes > 0) {
                    DecodeArgs sa = zb;
                    sa.blk_first = g0; sa.zblocks = nb; sa.lds_bytes = zstd_seq_lds_bytes(lanes);
                    DecodeArgs la = sa;
                    la.lds_bytes = zstd_lit_lds_bytes();
                    // (literals and sequences of a block are independent of each other: the two launches side by side -- the side
                    // stream follows the main one up to the walk and is joined in front of the replay)
                    const bool beside = !e->no_side;
                    if (beside && (rc = e->side_follows_stream())) return rc;
                    if ((rc = e->launch(CIMG_K_ZSTD_LIT, cimg_zstd_lit, la, (nb + ZSTD_LIT_BLOCKS - 1) / ZSTD_LIT_BLOCKS, 64, la.lds_bytes, beside ? e->s_side : e->stream))) return rc;
                    if ((rc = e->launch(CIMG_K_ZSTD_SEQ, cimg_zstd_seq, sa, (nb + lanes - 1) / lanes, 64, sa.lds_bytes))) return rc;
                    if (beside) {
                        if ((rc = e->hip(hipEventRecord(e->ev_side_done, e->s_side), "event record"))) return rc;
                        if ((rc = e->hip(hipStreamWaitEvent(e->stream, e->ev_side_done, 0), "stream wait"))) return rc;
                    }
                }
                DecodeArgs ra = zb;
                const bool fused_fits = zstd_kernel_lds_bytes(f.max_blocksize, 1) <= e->lds_per_wg;
                ra.blk_first = g0; ra.lds_bytes = replay_lds;
                ra.tune = fused_fits ? 0 : 2;                  // 2: a block the walk refuses has nobody to go to -- the replay fails its chunk
                if ((rc = e->launch(CIMG_K_ZSTD_REPLAY, cimg_zstd_replay, ra, nb, 64, ra.lds_bytes))) return rc;
                launched = true;
                if ((rc = cimg_engine_synchronize(e))) return rc;
                // what the replay left over is read from the blocks' own words; the counter is statistics (and a second witness)
                int pending = 0;
                for (int b = g0; b < g0 + nb; b++) pending += fall[b] == ZFALL_PENDING;
                if (pending || *refused) {
                    e->zstd_blocks_refused += std::max<int64_t>(pending, *refused);
                    if (e->verbose) fprintf(stderr, "[cimg] zstd: %d of %d plans did not fit their slots (counter: %u; cimg_decode_zstd reads those blocks)\n", pending, nb, (unsigned)*refused);
                    if (fused_fits) {
                        if ((rc = fused(zb, g0, nb))) return rc;
                        if ((rc = cimg_engine_synchronize(e))) return rc;
                    }
                    // a block that is still pending was decoded by nobody: its chunk fails, loudly
                    for (int b = g0; b < g0 + nb; b++) {
                        if (fall[b] != ZFALL_PENDING) continue;
                        const ChunkDesc* const descs = (const ChunkDesc*)e->shadow_dec.data();   // (this batch's descriptors as uploaded)
                        for (int i = 0; i < nchunks && e->shadow_dec.size() >= sizeof(ChunkDesc) * (size_t)nchunks; i++) {
                            if (b >= descs[i].blk0 && b < descs[i].blk0 + descs[i].nblocks) { if (st[i] >= 0) st[i] = ERR_FAILURE; break; }
                        }
                        if (e->shadow_dec.size() < sizeof(ChunkDesc) * (size_t)nchunks) return e->fail(ERR_FAILURE, "zstd: block %d was left to cimg_decode_zstd and never decoded", b);
                    }
                }
            }
        }
        return 0;
    };
    rc = read_path();
    if (rc) { if (ztimed) e->free_events.push_back(zev); return rc; }
    if (ztimed) { (void)hipEventRecord(zev.b, e->stream); e->pending[CIMG_K_DECODE_ZSTD].push_back(zev); }
    if (launched) {
        if ((rc = cimg_engine_synchronize(e))) return rc;
        e->zstd_batches++;
    }
    int first = 0;
    for (int i = 0; i < nchunks; i++) {
        if (status) status[i] = st[i];
        if (!first && st[i] < 0) first = st[i];
    }
    if (first) return e->fail(first, "chunk decode failed with blosc2 error %d", first);
    return 0;
}

int cimg_decompress_batch_device(cimg_engine* e, int32_t nchunks, const void* d_comp, const int64_t* comp_off,
                                 const int32_t* nbytes, const int32_t* blocksize, void* d_raw, const int64_t* raw_off,
                                 int32_t* status)
{
    return cimg_decompress_batch_device_sized(e, nchunks, d_comp, comp_off, nullptr, nbytes, blocksize, d_raw, raw_off, status);
}

int cimg_decompress_batch_device_sized(cimg_engine* e, int32_t nchunks, const void* d_comp, const int64_t* comp_off, const int32_t* comp_size,
                                       const int32_t* nbytes, const int32_t* blocksize, void* d_raw, const int64_t* raw_off,
                                       int32_t* status)
{
    std::lock_guard<std::recursive_mutex> lock_(e->mu);
    if (nchunks <= 0) return 0;
    if (!comp_off || !nbytes || !blocksize || !raw_off) return e->fail(ERR_INVALID_PARAM, "null argument");
    e->dflight_open = false;
    // (chunks a compress batch of this engine is still writing -- begun, not fetched -- are "behind the stream")
    const int rc = decompress_launch(e, nchunks, d_comp, comp_off, nbytes, blocksize, d_raw, raw_off, comp_size, e->cflight_chunks >= 0);
    if (rc) { if (e->dflight.launched) (void)cimg_engine_synchronize(e); return rc; }
    return decompress_finish(e, status);
}

int cimg_decompress_batch_device_begin(cimg_engine* e, int32_t nchunks, const void* d_comp, const int64_t* comp_off,
                                       const int32_t* nbytes, const int32_t* blocksize, void* d_raw, const int64_t* raw_off)
{
    return cimg_decompress_batch_device_begin_sized(e, nchunks, d_comp, comp_off, nullptr, nbytes, blocksize, d_raw, raw_off);
}

int cimg_decompress_batch_device_begin_sized(cimg_engine* e, int32_t nchunks, const void* d_comp, const int64_t* comp_off, const int32_t* comp_size,
                                             const int32_t* nbytes, const int32_t* blocksize, void* d_raw, const int64_t* raw_off)
{
    std::lock_guard<std::recursive_mutex> lock_(e->mu);
    e->dflight_open = false;
    e->dflight.nchunks = 0;
    if (nchunks <= 0) { e->dflight_open = true; return 0; }
    if (!comp_off || !nbytes || !blocksize || !raw_off) return e->fail(ERR_INVALID_PARAM, "null argument");
    const int rc = decompress_launch(e, nchunks, d_comp, comp_off, nbytes, blocksize, d_raw, raw_off, comp_size, e->cflight_chunks >= 0);
    if (!rc) e->dflight_open = true;
    else if (e->dflight.launched) (void)cimg_engine_synchronize(e);
    return rc;
}

int cimg_decompress_batch_device_fetch(cimg_engine* e, int32_t* status)
{
    std::lock_guard<std::recursive_mutex> lock_(e->mu);
    if (!e->dflight_open) return e->fail(ERR_INVALID_PARAM, "no decompress batch is in flight (cimg_decompress_batch_device_begin comes first)");
    e->dflight_open = false;
    if (e->dflight.nchunks <= 0) return 0;
    return decompress_finish(e, status);
}

// ---- host-resident batches: stage through device buffers owned by the engine -----------------------
// The batch is cut into groups of about 16 MiB of pixels.  While the kernels of group g run on the engine's stream, the
// pixels (or chunks) of group g + 1 travel host -> device on a second stream and the results of group g - 1 device -> host
// on a third: PCIe Gen5 moves both directions at once, and the kernels hide behind the copies.  (Launch first, copy
// second: from pageable memory a hipMemcpyAsync keeps the calling thread busy staging, so the kernels must already be
// queued when it starts.)
namespace {

struct Groups {
    std::vector<int> first;          // first chunk of every group, plus nchunks at the end
    void cut(int nchunks, const int32_t* bytes, int64_t target)
    {
        first.assign(1, 0);
        int64_t acc = 0;
        for (int i = 0; i < nchunks; i++) {
            acc += bytes[i];
            if (acc >= target && i + 1 < nchunks) { first.push_back(i + 1); acc = 0; }
        }
        first.push_back(nchunks);
    }
    int count() const { return (int)first.size() - 1; }
};

// A caller's PAGEABLE span page-locked for the duration of one batch call (hipHostRegister ... hipHostUnregister): from pageable
// memory hipMemcpyAsync stages through the runtime's own pinned bounce buffers on the calling thread (31 GB/s combined on
// configs[1], against 46 from page-locked memory: INTEGRATION.md section 4); a registered span is DMA'd from where it lies.
// Registering costs time per page, so only spans of at least `host_register_bytes` are registered (CIMG_HOST_REGISTER_MIB; 0 = never);
// memory that is page-locked already (hipHostMalloc, torch pinned tensors, a span the caller registered) is left alone, and so is
// everything when the registration fails (the copies then run as before).  Nothing stays registered behind the call: the
// engine cannot know when the caller frees the buffer.
struct HostPin {
    enum : int { MAX_RUNS = 32 };
    void* base[MAX_RUNS];
    int nruns = 0;
    cimg_engine* eng;
    // The chunks of a batch may lie in several separate allocations (an image handed over as one array per channel): they are
    // grouped into runs of memory (gaps of up to 64 KiB bridged) and every run of at least 4 MiB is registered by itself.
    HostPin(cimg_engine* e, const void* host, const int64_t* off, const int32_t* len, int n, bool writable) : eng(e)
    {
        (void)writable;
        if (!host || n <= 0 || e->host_register_bytes <= 0) return;
        int64_t total = 0;
        std::vector<std::pair<int64_t, int64_t>> seg;
        seg.reserve((size_t)n);
        for (int i = 0; i < n; i++) if (len[i] > 0) { seg.emplace_back(off[i], off[i] + (int64_t)len[i]); total += len[i]; }
        if (total < e->host_register_bytes || seg.empty()) return;
        std::sort(seg.begin(), seg.end());
        std::vector<std::pair<int64_t, int64_t>> runs;
        for (const auto& sg : seg) {
            if (!runs.empty() && sg.first <= runs.back().second + 65536) runs.back().second = std::max(runs.back().second, sg.second);
            else runs.push_back(sg);
        }
        for (const auto& r : runs) {
            if (nruns >= MAX_RUNS) break;
            if (r.second - r.first < (4ll << 20)) continue;
            // The run is rounded OUTWARD to whole pages: a copy whose source is page-locked in the middle and pageable at its ends is
            // refused by the runtime ("invalid argument" -- tried in round 5 after ADVICE r4 asked for inward rounding).  What the
            // advice was after is kept where it can be: "page-locked already" needs BOTH ends of the run to say so, and a run that is
            // locked at one end only -- partly inside somebody else's registration -- is left pageable rather than registered across it.
            const uintptr_t a = ((uintptr_t)host + (uintptr_t)r.first) & ~(uintptr_t)4095;
            const uintptr_t b = (((uintptr_t)host + (uintptr_t)r.second) + 4095) & ~(uintptr_t)4095;
            hipPointerAttribute_t at{}, at2{};
            const bool first_locked = hipPointerGetAttributes(&at, (const void*)a) == hipSuccess && at.type != hipMemoryTypeUnregistered;
            (void)hipGetLastError();
            const bool last_locked = hipPointerGetAttributes(&at2, (const void*)(b - 4096)) == hipSuccess && at2.type != hipMemoryTypeUnregistered;
            (void)hipGetLastError();
            if (first_locked || last_locked) continue;                  // page-locked already, or partly somebody else's registration
            if (hipHostRegister((void*)a, (size_t)(b - a), hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); continue; }
            base[nruns++] = (void*)a;
            e->host_registrations++;
        }
    }
    ~HostPin()
    {
        if (!nruns) return;
        // (every regular return has waited for its copies; an error return in the middle of a pipeline may not have)
        (void)hipStreamSynchronize(eng->s_h2d); (void)hipStreamSynchronize(eng->s_d2h); (void)hipStreamSynchronize(eng->stream);
        for (int i = 0; i < nruns; i++) (void)hipHostUnregister(base[i]);
    }
    HostPin(const HostPin&) = delete;
    HostPin& operator=(const HostPin&) = delete;
};

// host -> device copy of chunks [a, b): one transfer when they are contiguous on both sides
int copy_in(cimg_engine* e, hipStream_t st, uint8_t* dev, const int64_t* dev_off, const uint8_t* host, const int64_t* host_off,
            const int32_t* len, int a, int b, const char* what)
{
    bool contiguous = true;
    for (int i = a + 1; i < b; i++)
        if (host_off[i] != host_off[i - 1] + len[i - 1] || dev_off[i] != dev_off[i - 1] + len[i - 1]) { contiguous = false; break; }
    if (contiguous) {
        int64_t total = 0;
        for (int i = a; i < b; i++) total += len[i];
        return total ? e->hip(hipMemcpyAsync(dev + dev_off[a], host + host_off[a], (size_t)total, hipMemcpyHostToDevice, st), what) : 0;
    }
    for (int i = a; i < b; i++)
        if (len[i] > 0)
            if (int rc = e->hip(hipMemcpyAsync(dev + dev_off[i], host + host_off[i], (size_t)len[i], hipMemcpyHostToDevice, st), what)) return rc;
    return 0;
}

int copy_out(cimg_engine* e, hipStream_t st, uint8_t* host, const int64_t* host_off, const uint8_t* dev, const int64_t* dev_off,
             const int32_t* len, int a, int b, const char* what)
{
    bool contiguous = true;
    for (int i = a + 1; i < b; i++)
        if (host_off[i] != host_off[i - 1] + len[i - 1] || dev_off[i] != dev_off[i - 1] + len[i - 1]) { contiguous = false; break; }
    if (contiguous) {
        int64_t total = 0;
        for (int i = a; i < b; i++) total += len[i];
        return total ? e->hip(hipMemcpyAsync(host + host_off[a], dev + dev_off[a], (size_t)total, hipMemcpyDeviceToHost, st), what) : 0;
    }
    for (int i = a; i < b; i++)
        if (len[i] > 0)
            if (int rc = e->hip(hipMemcpyAsync(host + host_off[i], dev + dev_off[i], (size_t)len[i], hipMemcpyDeviceToHost, st), what)) return rc;
    return 0;
}

// chunks [a, b) of a finished group -> a block of host memory the caller hands out now that their sizes are known
int packed_out(cimg_engine* e, hipStream_t st, cimg_alloc_fn alloc, void* user, void** chunk_ptr, const uint8_t* dev, const int64_t* dev_off,
               const int32_t* cbytes, int a, int b)
{
    size_t total = 0;
    for (int i = a; i < b; i++) {
        if (cbytes[i] < 0) return e->fail(cbytes[i], "chunk %d failed to compress (code %d)", i, cbytes[i]);
        total += ((size_t)cbytes[i] + 63) & ~(size_t)63;
    }
    if (!total) { for (int i = a; i < b; i++) chunk_ptr[i] = nullptr; return 0; }
    uint8_t* block = (uint8_t*)alloc(user, total);
    if (!block) return e->fail(ERR_FAILURE, "the caller's allocator returned no memory for %zu bytes of chunks", total);
    std::vector<int64_t> hoff((size_t)b, 0);             // (indexed by chunk, like every offset table of copy_out)
    size_t at = 0;
    for (int i = a; i < b; i++) {
        hoff[(size_t)i] = (int64_t)at;
        chunk_ptr[i] = cbytes[i] > 0 ? block + at : nullptr;
        at += ((size_t)cbytes[i] + 63) & ~(size_t)63;
    }
    return copy_out(e, st, block, hoff.data(), dev, dev_off, cbytes, a, b, "chunk D2H");
}

// compress from host pixels; chunks stay in the device staging area, and go to h_comp as well when it is given
// (alloc != nullptr: the chunks of every group go, packed back to back at 64-byte boundaries, into a block the caller hands out when
// the group's sizes are known -- cimg_compress_batch_host_packed -- and chunk_ptr[i] says where chunk i went)
int compress_host_pipeline(cimg_engine* e, const cimg_cparams* p, int32_t nchunks, const void* h_raw, const int64_t* raw_off,
                           const int32_t* nbytes, const int32_t* destsize, int32_t* cbytes, void* h_comp, const int64_t* comp_off,
                           cimg_alloc_fn alloc = nullptr, void* alloc_user = nullptr, void** chunk_ptr = nullptr)
{
    e->fetch_off.clear();
    if (!p || !h_raw || !raw_off || !nbytes || !destsize || !cbytes) return e->fail(ERR_INVALID_PARAM, "null argument");
    (void)hipSetDevice(e->device);
    // pack pixels back to back (16-byte aligned) and give every chunk its full destsize on the device
    std::vector<int64_t> d_raw_off((size_t)nchunks), d_comp_off((size_t)nchunks);
    int64_t raw_total = 0, comp_total = 0;
    bool contiguous = true;
    for (int i = 0; i < nchunks; i++) {
        if (nbytes[i] < 0 || destsize[i] < 0) return e->fail(ERR_INVALID_PARAM, "negative size");
        if (i > 0 && raw_off[i] != raw_off[i - 1] + nbytes[i - 1]) contiguous = false;
    }
    for (int i = 0; i < nchunks; i++) {
        d_raw_off[(size_t)i] = raw_total;
        raw_total += contiguous ? (int64_t)nbytes[i] : (((int64_t)nbytes[i] + 15) & ~15ll);
        d_comp_off[(size_t)i] = comp_total;
        comp_total += ((int64_t)destsize[i] + 63) & ~63ll;
    }
    int rc;
    if ((rc = e->reserve(e->stage_raw, (size_t)raw_total + 64))) return rc;
    if ((rc = e->reserve(e->stage_comp, (size_t)comp_total + 64))) return rc;
    Groups G;
    G.cut(nchunks, nbytes, e->host_group_bytes);
    const uint8_t* hr = (const uint8_t*)h_raw;
    uint8_t* sr = (uint8_t*)e->stage_raw.p;
    uint8_t* sc = (uint8_t*)e->stage_comp.p;
    const int ng = G.count();
    HostPin pin_raw(e, h_raw, raw_off, nbytes, nchunks, false);       // (released when this function returns: every copy has been waited for)
    if (ng == 1) {
        // a small batch (the blosc2_compress_ctx shim: one chunk) has nothing to overlap: one stream, no events
        if ((rc = copy_in(e, e->stream, sr, d_raw_off.data(), hr, raw_off, nbytes, 0, nchunks, "pixels H2D"))) return rc;
        rc = compress_launch(e, p, nchunks, sr, d_raw_off.data(), nbytes, sc, d_comp_off.data(), destsize);
        const int frc = compress_finish(e, nchunks, cbytes);
        if (rc || frc) return rc ? rc : frc;
        if (h_comp) {
            if ((rc = copy_out(e, e->stream, (uint8_t*)h_comp, comp_off, sc, d_comp_off.data(), cbytes, 0, nchunks, "chunk D2H"))) return rc;
            if ((rc = cimg_engine_synchronize(e))) return rc;
        } else if (alloc) {
            if ((rc = packed_out(e, e->stream, alloc, alloc_user, chunk_ptr, sc, d_comp_off.data(), cbytes, 0, nchunks))) return rc;
            if ((rc = cimg_engine_synchronize(e))) return rc;
        }
        e->fetch_off = std::move(d_comp_off);
        e->fetch_len.assign(cbytes, cbytes + nchunks);
        return 0;
    }
    if ((rc = copy_in(e, e->s_h2d, sr, d_raw_off.data(), hr, raw_off, nbytes, G.first[0], G.first[1], "pixels H2D"))) return rc;
    if ((rc = e->hip(hipEventRecord(e->ev_h2d[0], e->s_h2d), "event record"))) return rc;
    for (int g = 0; g < ng; g++) {
        const int a = G.first[(size_t)g], b = G.first[(size_t)g + 1];
        if ((rc = e->hip(hipStreamWaitEvent(e->stream, e->ev_h2d[g & 1], 0), "stream wait"))) return rc;
        rc = compress_launch(e, p, b - a, sr, d_raw_off.data() + a, nbytes + a, sc, d_comp_off.data() + a, destsize + a);
        if (!rc && g + 1 < ng) {
            rc = copy_in(e, e->s_h2d, sr, d_raw_off.data(), hr, raw_off, nbytes, b, G.first[(size_t)g + 2], "pixels H2D");
            if (!rc) rc = e->hip(hipEventRecord(e->ev_h2d[(g + 1) & 1], e->s_h2d), "event record");
        }
        const int frc = compress_finish(e, b - a, cbytes + a);          // always: nothing may stay in flight on an error path
        if (rc || frc) { (void)hipStreamSynchronize(e->s_h2d); (void)hipStreamSynchronize(e->s_d2h); return rc ? rc : frc; }
        if (h_comp)
            if ((rc = copy_out(e, e->s_d2h, (uint8_t*)h_comp, comp_off, sc, d_comp_off.data(), cbytes, a, b, "chunk D2H"))) { (void)hipStreamSynchronize(e->s_d2h); return rc; }
        if (!h_comp && alloc)
            if ((rc = packed_out(e, e->s_d2h, alloc, alloc_user, chunk_ptr, sc, d_comp_off.data(), cbytes, a, b))) { (void)hipStreamSynchronize(e->s_d2h); return rc; }
    }
    if (h_comp || alloc) { if ((rc = e->hip(hipStreamSynchronize(e->s_d2h), "chunk D2H"))) return rc; }
    e->fetch_off = std::move(d_comp_off);
    e->fetch_len.assign(cbytes, cbytes + nchunks);
    return 0;
}

}  // namespace

int cimg_compress_batch_host_begin(cimg_engine* e, const cimg_cparams* p, int32_t nchunks, const void* h_raw,
                                   const int64_t* raw_off, const int32_t* nbytes, const int32_t* destsize, int32_t* cbytes)
{
    std::lock_guard<std::recursive_mutex> lock_(e->mu);
    e->fetch_off.clear();
    if (nchunks <= 0) return 0;
    return compress_host_pipeline(e, p, nchunks, h_raw, raw_off, nbytes, destsize, cbytes, nullptr, nullptr);
}

// ---- interleaved pixels -> planes --------------------------------------------------------------------------------------
static int deinterleave_launch(cimg_engine* e, const void* d_interleaved, int32_t nch, int32_t ts, int64_t npixels, void* d_planar, int64_t plane_stride)
{
    if (!d_interleaved || !d_planar) return e->fail(ERR_INVALID_PARAM, "null argument");
    if (nch < 1 || nch > 4096 || (ts != 1 && ts != 2 && ts != 4 && ts != 8)) return e->fail(ERR_INVALID_PARAM, "deinterleave: %d channels of %d-byte elements are not supported", nch, ts);
    if (npixels < 0 || (plane_stride & 15) || plane_stride < npixels * ts) return e->fail(ERR_INVALID_PARAM, "deinterleave: the plane stride must be a multiple of 16 and hold a plane");
    if ((((uintptr_t)d_interleaved) | ((uintptr_t)d_planar)) & 15) return e->fail(ERR_INVALID_PARAM, "deinterleave: buffers must be 16-byte aligned");
    if (nch * ts * 16 > 16384) return e->fail(ERR_INVALID_PARAM, "deinterleave: %d channels of %d bytes do not fit a tile", nch, ts);
    if (npixels == 0) return 0;
    const int tile = deinterleave_tile_pixels(nch, ts);
    const int64_t tiles = (npixels + tile - 1) / tile;
    if (tiles > 0x7fffffff) return e->fail(ERR_INVALID_PARAM, "deinterleave: too many pixels for one call");
    const int lds = deinterleave_lds_bytes(nch, ts);
    DeinterleaveArgs a{(const uint8_t*)d_interleaved, (uint8_t*)d_planar, plane_stride, npixels, nch, ts, tile, lds};
    return e->launch(CIMG_K_DEINTERLEAVE, cimg_deinterleave, a, (int)tiles, 64, lds);
}

int cimg_deinterleave_device(cimg_engine* e, const void* d_interleaved, int32_t nchannels, int32_t typesize, int64_t npixels,
                             void* d_planar, int64_t plane_stride)
{
    std::lock_guard<std::recursive_mutex> lock_(e->mu);
    (void)hipSetDevice(e->device);
    e->begin_batch(0);
    return deinterleave_launch(e, d_interleaved, nchannels, typesize, npixels, d_planar, plane_stride);
}

int cimg_compress_batch_host_interleaved_begin(cimg_engine* e, const cimg_cparams* p, int32_t nchannels, int64_t npixels,
                                               const void* h_interleaved, int32_t nchunks, const int64_t* raw_off,
                                               const int32_t* nbytes, const int32_t* destsize, int32_t* cbytes)
{
    std::lock_guard<std::recursive_mutex> lock_(e->mu);
    e->fetch_off.clear();
    if (nchunks <= 0) return 0;
    if (!p || !h_interleaved || !raw_off || !nbytes || !destsize || !cbytes) return e->fail(ERR_INVALID_PARAM, "null argument");
    (void)hipSetDevice(e->device);
    const int ts = p->typesize;
    if (nchannels < 1 || npixels <= 0 || (ts != 1 && ts != 2 && ts != 4 && ts != 8)) return e->fail(ERR_INVALID_PARAM, "interleaved batch: %d channels of %d-byte elements are not supported", nchannels, ts);
    const int64_t plane_stride = (npixels * ts + 15) & ~15ll;
    const int64_t planar_bytes = plane_stride * nchannels;
    std::vector<int64_t> d_comp_off((size_t)nchunks);
    int64_t comp_total = 0;
    for (int i = 0; i < nchunks; i++) {
        if (nbytes[i] < 0 || destsize[i] < 0 || raw_off[i] < 0 || raw_off[i] + nbytes[i] > planar_bytes) return e->fail(ERR_INVALID_PARAM, "interleaved batch: chunk %d lies outside the planes", i);
        d_comp_off[(size_t)i] = comp_total;
        comp_total += ((int64_t)destsize[i] + 63) & ~63ll;
    }
    int rc;
    const size_t il_bytes = (size_t)npixels * (size_t)nchannels * (size_t)ts;
    if ((rc = e->reserve(e->stage_il, il_bytes + 64))) return rc;
    if ((rc = e->reserve(e->stage_raw, (size_t)planar_bytes + 64))) return rc;
    if ((rc = e->reserve(e->stage_comp, (size_t)comp_total + 64))) return rc;
    // upload once, split on the device, compress from there: everything on the engine's stream, in order
    if ((rc = e->hip(hipMemcpyAsync(e->stage_il.p, h_interleaved, il_bytes, hipMemcpyHostToDevice, e->stream), "interleaved pixels H2D"))) return rc;
    e->begin_batch(0);
    if ((rc = deinterleave_launch(e, e->stage_il.p, nchannels, ts, npixels, e->stage_raw.p, plane_stride))) { (void)hipStreamSynchronize(e->stream); return rc; }
    rc = compress_launch(e, p, nchunks, e->stage_raw.p, raw_off, nbytes, e->stage_comp.p, d_comp_off.data(), destsize);
    const int frc = compress_finish(e, nchunks, cbytes);                // always: nothing may stay in flight on an error path
    if (rc || frc) return rc ? rc : frc;
    e->fetch_off = std::move(d_comp_off);
    e->fetch_len.assign(cbytes, cbytes + nchunks);
    return 0;
}

int cimg_compress_batch_host_fetch(cimg_engine* e, int32_t nchunks, void* h_comp, const int64_t* comp_off)
{
    std::lock_guard<std::recursive_mutex> lock_(e->mu);
    if (nchunks <= 0) return 0;
    if (!h_comp || !comp_off) return e->fail(ERR_INVALID_PARAM, "null argument");
    if ((size_t)nchunks != e->fetch_off.size()) return e->fail(ERR_INVALID_PARAM, "no compressed batch of %d chunks is waiting to be fetched", nchunks);
    (void)hipSetDevice(e->device);
    int rc = copy_out(e, e->s_d2h, (uint8_t*)h_comp, comp_off, (const uint8_t*)e->stage_comp.p, e->fetch_off.data(), e->fetch_len.data(), 0, nchunks, "chunk D2H");
    e->fetch_off.clear();
    const int src = e->hip(hipStreamSynchronize(e->s_d2h), "chunk D2H");
    return rc ? rc : src;
}

int cimg_compress_batch_host_packed(cimg_engine* e, const cimg_cparams* p, int32_t nchunks, const void* h_raw, const int64_t* raw_off,
                                    const int32_t* nbytes, const int32_t* destsize, int32_t* cbytes, cimg_alloc_fn alloc, void* user, void** chunk_ptr)
{
    std::lock_guard<std::recursive_mutex> lock_(e->mu);
    if (nchunks <= 0) return 0;
    if (!alloc || !chunk_ptr) return e->fail(ERR_INVALID_PARAM, "null argument");
    const int rc = compress_host_pipeline(e, p, nchunks, h_raw, raw_off, nbytes, destsize, cbytes, nullptr, nullptr, alloc, user, chunk_ptr);
    e->fetch_off.clear();                                  // delivered: nothing is left to fetch
    return rc;
}

int cimg_compress_batch_host(cimg_engine* e, const cimg_cparams* p, int32_t nchunks, const void* h_raw,
                             const int64_t* raw_off, const int32_t* nbytes, void* h_comp, const int64_t* comp_off,
                             const int32_t* destsize, int32_t* cbytes)
{
    std::lock_guard<std::recursive_mutex> lock_(e->mu);
    if (nchunks <= 0) return 0;
    if (!h_comp || !comp_off) return e->fail(ERR_INVALID_PARAM, "null argument");
    const int rc = compress_host_pipeline(e, p, nchunks, h_raw, raw_off, nbytes, destsize, cbytes, h_comp, comp_off);
    e->fetch_off.clear();                                  // delivered: nothing is left to fetch
    return rc;
}

int cimg_decompress_batch_host(cimg_engine* e, int32_t nchunks, const void* h_comp, const int64_t* comp_off,
                               void* h_raw, const int64_t* raw_off, const int32_t* raw_capacity, int32_t* status)
{
    return cimg_decompress_batch_host_sized(e, nchunks, h_comp, comp_off, nullptr, h_raw, raw_off, raw_capacity, status);
}

int cimg_decompress_batch_host_sized(cimg_engine* e, int32_t nchunks, const void* h_comp, const int64_t* comp_off, const int32_t* comp_size,
                                     void* h_raw, const int64_t* raw_off, const int32_t* raw_capacity, int32_t* status)
{
    std::lock_guard<std::recursive_mutex> lock_(e->mu);
    if (nchunks <= 0) return 0;
    if (!h_comp || !h_raw || !comp_off || !raw_off || !raw_capacity) return e->fail(ERR_INVALID_PARAM, "null argument");
    (void)hipSetDevice(e->device);
    e->fetch_off.clear();                                  // the staging area is reused: a pending _fetch is void
    const uint8_t* hc = (const uint8_t*)h_comp;
    std::vector<int64_t> d_comp_off((size_t)nchunks), d_raw_off((size_t)nchunks);
    std::vector<int32_t> nb((size_t)nchunks), bs((size_t)nchunks), cb((size_t)nchunks);
    int64_t comp_total = 0, raw_total = 0;
    bool dense = true;                                     // pixels back to back on the host: keep them so on the device
    for (int i = 0; i < nchunks; i++) {
        const uint8_t* c = hc + comp_off[i];
        int32_t n, cbv, b;
        if (comp_size && comp_size[i] < HEADER_LEN) return e->fail(ERR_READ_BUFFER, "chunk %d: %d bytes cannot hold a header", i, comp_size[i]);
        memcpy(&n, c + OFF_NBYTES, 4); memcpy(&b, c + OFF_BLOCKSIZE, 4); memcpy(&cbv, c + OFF_CBYTES, 4);
        if (c[0] > 5) return e->fail(ERR_VERSION_SUPPORT, "chunk %d: format version %d", i, c[0]);
        if (n < 0 || cbv < HEADER_LEN || b <= 0 || (n > 0 && b > n) || c[OFF_TYPESIZE] == 0) return e->fail(ERR_INVALID_HEADER, "chunk %d: invalid header", i);
        if (comp_size && cbv > comp_size[i]) return e->fail(ERR_READ_BUFFER, "chunk %d: header says %d compressed bytes, the buffer holds %d", i, cbv, comp_size[i]);
        if (n > raw_capacity[i]) return e->fail(ERR_WRITE_BUFFER, "chunk %d: needs %d bytes, buffer has %d", i, n, raw_capacity[i]);
        nb[(size_t)i] = n; bs[(size_t)i] = b; cb[(size_t)i] = cbv;
        if (i > 0 && raw_off[i] != raw_off[i - 1] + nb[(size_t)i - 1]) dense = false;
    }
    for (int i = 0; i < nchunks; i++) {
        d_comp_off[(size_t)i] = comp_total; comp_total += ((int64_t)cb[(size_t)i] + 63) & ~63ll;
        d_raw_off[(size_t)i] = raw_total; raw_total += dense ? (int64_t)nb[(size_t)i] : (((int64_t)nb[(size_t)i] + 15) & ~15ll);
    }
    int rc;
    if ((rc = e->reserve(e->stage_comp, (size_t)comp_total + 64))) return rc;
    if ((rc = e->reserve(e->stage_raw, (size_t)raw_total + 64))) return rc;
    Groups G;
    G.cut(nchunks, nb.data(), e->host_group_bytes);
    const int ng = G.count();
    uint8_t* sr = (uint8_t*)e->stage_raw.p;
    uint8_t* sc = (uint8_t*)e->stage_comp.p;
    uint8_t* hr = (uint8_t*)h_raw;
    HostPin pin_out(e, h_raw, raw_off, nb.data(), nchunks, true);     // the pixels' destination (every return path below has waited for its copies)
    std::vector<int32_t> st((size_t)nchunks, 0);
    std::vector<int32_t> deliver((size_t)nchunks, 0);
    if (ng == 1) {
        if ((rc = copy_in(e, e->stream, sc, d_comp_off.data(), hc, comp_off, cb.data(), 0, nchunks, "chunk H2D"))) return rc;
        rc = decompress_launch(e, nchunks, sc, d_comp_off.data(), nb.data(), bs.data(), sr, d_raw_off.data());
        if (!e->dflight.launched) { (void)cimg_engine_synchronize(e); return rc; }     // rejected before anything was enqueued
        int drc = rc;
        { const int frc = decompress_finish(e, st.data()); if (!drc) drc = frc; }
        if (status) memcpy(status, st.data(), sizeof(int32_t) * (size_t)nchunks);
        if (rc || drc == ERR_FAILURE) return drc;
        const std::string chunk_error1 = e->err;
        for (int i = 0; i < nchunks; i++) deliver[(size_t)i] = (nb[(size_t)i] > 0 && st[(size_t)i] == 0) ? nb[(size_t)i] : 0;
        if ((rc = copy_out(e, e->stream, hr, raw_off, sr, d_raw_off.data(), deliver.data(), 0, nchunks, "pixels D2H"))) return rc;
        if ((rc = cimg_engine_synchronize(e))) return rc;
        if (drc) e->err = chunk_error1;
        return drc;
    }
    if ((rc = copy_in(e, e->s_h2d, sc, d_comp_off.data(), hc, comp_off, cb.data(), G.first[0], G.first[1], "chunk H2D"))) return rc;
    if ((rc = e->hip(hipEventRecord(e->ev_h2d[0], e->s_h2d), "event record"))) return rc;
    int first_bad = 0;
    std::string chunk_error;
    for (int g = 0; g < ng; g++) {
        const int a = G.first[(size_t)g], b = G.first[(size_t)g + 1];
        if ((rc = e->hip(hipStreamWaitEvent(e->stream, e->ev_h2d[g & 1], 0), "stream wait"))) return rc;
        rc = decompress_launch(e, b - a, sc, d_comp_off.data() + a, nb.data() + a, bs.data() + a, sr, d_raw_off.data() + a);
        int crc = 0;
        if (g + 1 < ng) {
            crc = copy_in(e, e->s_h2d, sc, d_comp_off.data(), hc, comp_off, cb.data(), b, G.first[(size_t)g + 2], "chunk H2D");
            if (!crc) crc = e->hip(hipEventRecord(e->ev_h2d[(g + 1) & 1], e->s_h2d), "event record");
        }
        int drc = rc;
        if (e->dflight.launched) { const int frc = decompress_finish(e, st.data() + a); if (!drc) drc = frc; }
        else (void)cimg_engine_synchronize(e);
        if (rc || drc == ERR_FAILURE || crc) { (void)hipStreamSynchronize(e->s_h2d); (void)hipStreamSynchronize(e->s_d2h); return crc ? crc : drc; }
        if (drc && !first_bad) { first_bad = drc; chunk_error = e->err; }
        // chunks that decoded cleanly are still delivered when a neighbour in the batch is damaged
        for (int i = a; i < b; i++) deliver[(size_t)i] = (nb[(size_t)i] > 0 && st[(size_t)i] == 0) ? nb[(size_t)i] : 0;
        if ((rc = copy_out(e, e->s_d2h, hr, raw_off, sr, d_raw_off.data(), deliver.data(), a, b, "pixels D2H"))) { (void)hipStreamSynchronize(e->s_d2h); return rc; }
    }
    if (status) memcpy(status, st.data(), sizeof(int32_t) * (size_t)nchunks);
    if ((rc = e->hip(hipStreamSynchronize(e->s_d2h), "pixels D2H"))) return rc;
    if (first_bad) e->err = chunk_error;
    return first_bad;
}

}  // extern "C"
