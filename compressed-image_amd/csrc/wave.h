// wave.h -- the 64-lane wavefront vocabulary the codec kernels are written in.
//
// The kernels in this directory are written once, in an explicit "wave-uniform control flow +
// per-lane data" style:
//   * plain C++ variables are wave-uniform (they live in SGPRs on gfx950),
//   * LV<T> variables hold one value per lane, touched only inside FOR_LANES(l) { ... x[l] ... },
//   * cross-lane traffic goes through ballot()/readlane()/wave_*().
// Compiled by hipcc for gfx950 this collapses to ordinary SIMT code (LV<T> is a single register,
// FOR_LANES runs once with l = lane id).  Compiled with -DCIMG_EMULATE by g++ the same source runs
// on the host with LV<T> = T[64] and FOR_LANES = a 64-iteration loop.  The emulated build exists only
// so that tests/ can exercise the kernel *logic* in a container without a GPU (tests/emu); it is
// never linked into libcimg_hip.so and the product has no CPU path.
//
// Rules that keep both builds equivalent:
//   R1  A FOR_LANES body never reads an LDS/global location that another lane writes in the same
//       body (on the GPU all lanes read before any lane writes; the emulator runs lanes in turn).
//   R2  Bodies that store to possibly-colliding addresses use FOR_LANES_W: the emulator then visits
//       lanes in a test-selected order (ascending / descending / shuffled) so that code depending on
//       which colliding lane "wins" is caught.
//   R3  Everything outside FOR_LANES is wave-uniform.
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef CIMG_EMULATE
// ------------------------------------------------------------------------------------------------
//  host emulation
// ------------------------------------------------------------------------------------------------
#define CIMG_DEV inline
#define CIMG_HD inline
#define CIMG_UNROLL
#define CIMG_DEV_NOINLINE inline
#define CIMG_DEV_OUTLINE inline
// the workgroup's LDS: in the emulator it is whatever buffer the harness passed in
#define CIMG_LDS_BASE(passed) (passed)
typedef uint8_t* cimg_global_u8p;
#define CIMG_AS_GLOBAL(p) (p)
// kernel arguments: a plain pointer in the emulator (see the device build)
template <class T> using kernarg_ptr = const T*;
template <class T> inline kernarg_ptr<T> fresh(kernarg_ptr<T> p) { return p; }
#define CIMG_OWN_KERNARGS(T, passed) (passed)
typedef volatile uint16_t* cimg_lds_vu16p;
typedef const uint32_t* cimg_lds_cu32p;
typedef const uint8_t* cimg_lds_cu8p;
typedef const uint16_t* cimg_lds_cu16p;
typedef uint8_t* cimg_lds_u8p;
#define CIMG_AS_LDS_U8(p) ((uint8_t*)(p))
#define CIMG_AS_LDS_CU16(p) ((const uint16_t*)(p))
#define CIMG_AS_LDS_CU32(p) ((const uint32_t*)(p))
#define CIMG_AS_LDS_CU8(p) ((const uint8_t*)(p))
#define CIMG_AS_LDS_VU16(p) (reinterpret_cast<volatile uint16_t*>(p))

namespace cimg {

template <class T> struct LV {
    T v[64];
    T& operator[](int l) { return v[l]; }
    const T& operator[](int l) const { return v[l]; }
};

template <class T> inline void needed_here(const LV<T>&) {}
extern int g_emu_write_order;                 // 0 ascending, 1 descending, 2 shuffled
inline int emu_lane(int i)
{
    if (g_emu_write_order == 1) return 63 - i;
    if (g_emu_write_order == 2) return (i * 37 + 11) & 63;     // 37 is odd -> a permutation of 0..63
    return i;
}
#define FOR_LANES(l) for (int l = 0; l < 64; ++l)
#define FOR_LANES_W(l) for (int l##_i = 0, l = ::cimg::emu_lane(0); l##_i < 64; ++l##_i, l = ::cimg::emu_lane(l##_i & 63))

inline uint64_t ballot(const LV<bool>& p)
{
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) if (p.v[l]) m |= 1ull << l;
    return m;
}
template <class T> inline T readlane(const LV<T>& x, int lane) { return x.v[lane & 63]; }
inline int uni(int x) { return x; }
inline uint32_t uni(uint32_t x) { return x; }
inline int ctz64(uint64_t m) { return m ? __builtin_ctzll(m) : 64; }
inline int popc64(uint64_t m) { return __builtin_popcountll(m); }
inline uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (sh & 3))); }
// exclusive prefix sum across the wave
inline void wave_exscan(const LV<int>& x, LV<int>& out, int& total)
{
    int run = 0;
    for (int l = 0; l < 64; ++l) { out.v[l] = run; run += x.v[l]; }
    total = run;
}
inline float fast_rcp(float x) { return 1.0f / x; }
inline int wave_max(const LV<int>& x) { int m = x.v[0]; for (int l = 1; l < 64; ++l) m = x.v[l] > m ? x.v[l] : m; return m; }
inline void debug_stamp(uint64_t*, int, int) {}
inline int wave_slot() { return 0; }
inline int wave_simd() { return 0; }
inline void wave_priority(int) {}
inline void wave_sleep64(int) {}
template <class T> inline void writelane(LV<T>& x, int lane, T value) { x.v[lane & 63] = value; }
// lane `lane` of x becomes the wave-uniform `value` (v_writelane_b32)
template <class T> inline void setlane(LV<T>& x, int lane, T value) { x.v[lane & 63] = value; }
template <class T> inline void lane_gather(const LV<T>& x, const LV<int>& idx, LV<T>& out)
{
    T tmp[64];
    for (int l = 0; l < 64; ++l) tmp[l] = x.v[idx.v[l] & 63];
    for (int l = 0; l < 64; ++l) out.v[l] = tmp[l];
}
// out[idx[l]] = x[l]; lanes nobody sends to get 0 (ds_permute).  Colliding senders: any one of them wins.
template <class T> inline void lane_scatter(const LV<T>& x, const LV<int>& idx, LV<T>& out)
{
    T tmp[64];
    for (int l = 0; l < 64; ++l) tmp[l] = T(0);
    for (int i = 0; i < 64; ++i) { const int l = emu_lane(i); tmp[idx.v[l] & 63] = x.v[l]; }
    for (int l = 0; l < 64; ++l) out.v[l] = tmp[l];
}
// number of set bits of mask below lane l
inline int lane_rank(uint64_t mask, int l) { return __builtin_popcountll(mask & ((1ull << l) - 1)); }
inline uint32_t queue_pop(uint32_t* head) { return (*head)++; }
// cross-workgroup hand-over inside a launch (encode_kernel.h: chunks assembled by the waves that encoded them): plain memory here
inline void fence_release() {}
inline void fence_acquire() {}
inline uint32_t atomic_add_agent(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
inline uint32_t atomic_load_agent(const uint32_t* p) { return *p; }
inline uint32_t atomic_add_workgroup(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
inline void atomic_store_agent(uint32_t* p, uint32_t v) { *p = v; }
inline uint32_t atomic_exchange_agent(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = v; return o; }
inline void wave_nap() {}
inline void stores_performed() {}
inline void atomic_count(uint32_t* p) { ++*p; }
// value held by lane l-1 (lane 0 keeps its own)
template <class T> inline void lane_prev(const LV<T>& x, LV<T>& out)
{
    T keep = x.v[0];
    for (int l = 0; l < 64; ++l) { const T cur = x.v[l]; out.v[l] = keep; keep = cur; }
    out.v[0] = x.v[0];
}

}  // namespace cimg

#else
// ------------------------------------------------------------------------------------------------
//  gfx950 device build
// ------------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#define CIMG_DEV __device__ __forceinline__
#define CIMG_UNROLL _Pragma("unroll")
#define CIMG_HD __host__ __device__ __forceinline__
#define CIMG_DEV_NOINLINE __device__ __forceinline__   /* out-of-line was measured 10 % slower (call + flat pointers) */
#define CIMG_DEV_OUTLINE __device__ __attribute__((noinline))   /* really out of line: code that runs once per work item, no LDS access */
// the workgroup's LDS is always reached through the ONE pointer the kernel derives from its
// `extern __shared__` array: a second extern symbol would physically alias it while the compiler treats
// two globals as distinct objects and may reorder accesses between them
#define CIMG_LDS_BASE(passed) (passed)
// a pointer parameter of an out-of-line device function is generic (flat); this names it global again
typedef __attribute__((address_space(1))) uint8_t* cimg_global_u8p;
#define CIMG_AS_GLOBAL(p) ((cimg_global_u8p)(p))
// Kernel arguments of a persistent kernel are read THROUGH a pointer into the kernel-argument segment (s_load at the
// point of use) instead of being taken by value: by value the compiler loads every field once at entry and keeps it in
// SGPRs across the whole codec loop, where scalar registers are the scarce resource (the LZ4 kernel spilled 38 of them).
// fresh() hides the pointer from the optimizer for a moment so that loads after it are not merged with loads before it.
template <class T> using kernarg_ptr = const T __attribute__((address_space(4)))*;
template <class T> __device__ __forceinline__ kernarg_ptr<T> fresh(kernarg_ptr<T> p) { asm volatile("" : "+s"(p)); return p; }
template <class T> __device__ __forceinline__ kernarg_ptr<T> kernel_args() { return (kernarg_ptr<T>)__builtin_amdgcn_kernarg_segment_ptr(); }
// inside an out-of-line device function the pointer arrives in vector registers (the calling convention): both halves through
// v_readfirstlane make it a scalar pointer into the kernel-argument segment again.  (__builtin_amdgcn_kernarg_segment_ptr() is
// NOT usable there: in a callee it came back null on gfx950 / ROCm 7.2 and the first s_load faulted.)
template <class T> __device__ __forceinline__ kernarg_ptr<T> kernarg_scalar_again(kernarg_ptr<T> p)
{
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (kernarg_ptr<T>)(((uint64_t)hi << 32) | lo);
}
#define CIMG_OWN_KERNARGS(T, passed) (::kernarg_scalar_again<T>(passed))
// volatile accesses are skipped by the compiler's address-space inference and would become FLAT ops
// (slow, and not ordered with ds_* ops): LDS pointers that must be volatile carry the address space explicitly
typedef volatile __attribute__((address_space(3))) uint16_t* cimg_lds_vu16p;
#define CIMG_AS_LDS_VU16(p) ((cimg_lds_vu16p)(p))
// read-only LDS data behind a pointer the compiler can no longer trace to the __shared__ array (it went through a struct in memory,
// or a select between an LDS and a global address): as a generic pointer every access is a FLAT load -- several hundred cycles and
// a vmcnt wait each -- so the hot loops that KNOW where their tables lie say so
typedef const __attribute__((address_space(3))) uint32_t* cimg_lds_cu32p;
typedef const __attribute__((address_space(3))) uint8_t* cimg_lds_cu8p;
typedef const __attribute__((address_space(3))) uint16_t* cimg_lds_cu16p;
typedef __attribute__((address_space(3))) uint8_t* cimg_lds_u8p;
#define CIMG_AS_LDS_U8(p) ((cimg_lds_u8p)(p))
#define CIMG_AS_LDS_CU16(p) ((cimg_lds_cu16p)(p))
#define CIMG_AS_LDS_CU32(p) ((cimg_lds_cu32p)(p))
#define CIMG_AS_LDS_CU8(p) ((cimg_lds_cu8p)(p))

namespace cimg {

template <class T> struct LV {
    T v;
    CIMG_DEV T& operator[](int) { return v; }
    CIMG_DEV const T& operator[](int) const { return v; }
};

#define FOR_LANES(l) for (int l = (int)__lane_id(), l##_once = 1; l##_once; l##_once = 0)
#define FOR_LANES_W(l) FOR_LANES(l)

// "this value is needed HERE": keeps the compiler from sinking the load that produces x into a branch behind this point, where it
// would be a dependent LDS round trip of its own instead of sharing one with the loads around it (encode_kernel.h: run_path)
template <class T> CIMG_DEV void needed_here(const LV<T>& x) { asm volatile("" :: "v"(x.v)); }

CIMG_DEV uint64_t ballot(const LV<bool>& p) { return __ballot(p.v); }
template <class T> CIMG_DEV T readlane(const LV<T>& x, int lane)
{
    static_assert(sizeof(T) <= 4, "readlane moves one dword");
    return (T)__builtin_amdgcn_readlane((int)x.v, lane);
}
CIMG_DEV int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
CIMG_DEV uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
CIMG_DEV int ctz64(uint64_t m) { return m ? (int)__builtin_ctzll(m) : 64; }
CIMG_DEV int popc64(uint64_t m) { return (int)__builtin_popcountll(m); }
CIMG_DEV uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
// exclusive prefix sum across the wave: the DPP scan ladder (row_shr 1/2/4/8, row_bcast 15/31) -- six VALU
// adds and no trip through the LDS crossbar (a __shfl_up ladder costs six ds_bpermute round trips)
CIMG_DEV void wave_exscan(const LV<int>& x, LV<int>& out, int& total)
{
    int v = x.v;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    total = __builtin_amdgcn_readlane(v, 63);
    out.v = v - x.v;
}
// v_rcp_f32: one instruction, 1 ulp (callers fix the quotient up)
CIMG_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
CIMG_DEV int wave_max(const LV<int>& x)
{
    int v = x.v;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int n = __shfl_xor(v, d); v = n > v ? n : v; }
    return __builtin_amdgcn_readfirstlane(v);
}
template <class T> CIMG_DEV void writelane(LV<T>& x, int lane, T value)
{
    static_assert(sizeof(T) == 4, "writelane moves one dword");
    x.v = ((int)__lane_id() == lane) ? value : x.v;
}
// lane `lane` of x becomes the wave-uniform `value`: ONE v_writelane_b32 (the select form above is a compare + a conditional move
// and needs the lane id in a register); the builtin has no clang spelling in ROCm 7.2, the intrinsic is reached by its name
__device__ int cimg_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
template <class T> CIMG_DEV void setlane(LV<T>& x, int lane, T value)
{
    static_assert(sizeof(T) == 4, "setlane moves one dword");
    x.v = (T)cimg_writelane_i32((int)value, lane, (int)x.v);
}
// out[l] = x[idx[l]] (ds_bpermute: LDS crossbar, no memory access)
template <class T> CIMG_DEV void lane_gather(const LV<T>& x, const LV<int>& idx, LV<T>& out)
{
    static_assert(sizeof(T) == 4, "lane_gather moves one dword");
    out.v = (T)__builtin_amdgcn_ds_bpermute(idx.v << 2, (int)x.v);
}
// out[idx[l]] = x[l]; lanes nobody sends to get 0 (ds_permute: LDS crossbar, no memory access).  Every lane sends, so
// callers give idle lanes a destination nobody reads.
template <class T> CIMG_DEV void lane_scatter(const LV<T>& x, const LV<int>& idx, LV<T>& out)
{
    static_assert(sizeof(T) == 4, "lane_scatter moves one dword");
    out.v = (T)__builtin_amdgcn_ds_permute(idx.v << 2, (int)x.v);
}
// number of set bits of mask below this lane (v_mbcnt_lo / v_mbcnt_hi)
CIMG_DEV int lane_rank(uint64_t mask, int)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// one returning device-scope atomic on the queue head (MI355X_MICROARCH.md: 'dequeue', ~0.3-1.1 us)
CIMG_DEV uint32_t queue_pop(uint32_t* head) { return __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Cross-workgroup hand-over INSIDE a launch (encode_kernel.h: a chunk is laid out and copied into place by the waves of the launch
// that encoded it).  Producer: plain stores, fence_release() by the whole wave, then a relaxed agent-scope atomic by one lane;
// consumer: relaxed agent-scope atomic by one lane, then fence_acquire() by the whole wave, then plain loads.  On gfx950 the two
// fences are what writes the producer's L2 back / drops the consumer's stale lines when the two sit on different XCDs.
#ifdef CIMG_EXP_WG_FENCE     /* timing experiment only: no L2 write-back / invalidate -> results may be stale */
CIMG_DEV void fence_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
CIMG_DEV void fence_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
#else
CIMG_DEV void fence_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
CIMG_DEV void fence_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
#endif
CIMG_DEV uint32_t atomic_add_agent(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CIMG_DEV uint32_t atomic_load_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a counter in LDS that the waves of one workgroup share (zstd_kernel.h: which stream of the block next)
CIMG_DEV uint32_t atomic_add_workgroup(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
CIMG_DEV void atomic_store_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CIMG_DEV uint32_t atomic_exchange_agent(uint32_t* p, uint32_t v) { return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every store this wave issued so far is acknowledged (write-through stores: at the memory side)
CIMG_DEV void stores_performed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#ifndef CIMG_NAP_REPEAT
#define CIMG_NAP_REPEAT 1
#endif
CIMG_DEV void wave_nap() { _Pragma("unroll") for (int i = 0; i < CIMG_NAP_REPEAT; ++i) __builtin_amdgcn_s_sleep(127); }
// a statistics counter that may live in page-locked host memory (system scope)
CIMG_DEV void atomic_count(uint32_t* p) { (void)__hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// Clock reads for diagnostics.  s_memtime / s_memrealtime are SMEM ops: they count on lgkmcnt together
// with LDS reads but return OUT OF ORDER with them, so a counted wait after one no longer says which LDS
// read is back (a plain __builtin_readcyclecounter() inside the LZ4 loops produced wrong LDS data and
// wedged the kernel).  The wait therefore lives inside the same asm statement (cdna_hip_programming.md
// section 7, 'In-kernel stamps').
CIMG_DEV unsigned long long cimg_cycles()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
CIMG_DEV unsigned long long cimg_realtime()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
// the wave's slot on its SIMD (HW_ID.wave_id): waves that share a SIMD have different slots; the lower slot is the older wave,
// and the issue arbiter serves the oldest wave first
CIMG_DEV int wave_slot() { return (int)__builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (3 << 11)); }
CIMG_DEV int wave_simd() { return (int)__builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (4 << 6) | (1 << 11)); }
// s_setprio takes an immediate
CIMG_DEV void wave_priority(int p)
{
    if (p <= 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}
// n x 64 cycles of s_sleep (n wave-uniform, at most a few hundred)
CIMG_DEV void wave_sleep64(int n)
{
    for (int i = 0; i < n && i < 1024; i += 8) __builtin_amdgcn_s_sleep(8);
}
// diagnostic builds only (dbg != nullptr): slot[16*w + 4*which], which = 0..3 = {shader clock, 100 MHz wall clock | hw id}
CIMG_DEV void debug_stamp(uint64_t* dbg, int w, int which)
{
    if (dbg == nullptr) return;
    const uint64_t t = cimg_cycles();
    const uint64_t r = cimg_realtime();
    const uint32_t hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11));
    if (__lane_id() == 0) {
        dbg[16 * (size_t)w + 4 * which + 0] = t;
        dbg[16 * (size_t)w + 4 * which + 1] = r;
        dbg[16 * (size_t)w + 4 * which + 2] = hw;
        dbg[16 * (size_t)w + 4 * which + 3] = xcc;
    }
}
// value held by lane l-1 (lane 0 keeps its own)
template <class T> CIMG_DEV void lane_prev(const LV<T>& x, LV<T>& out)
{
    static_assert(sizeof(T) == 4, "lane_prev moves one dword");
    // DPP wave_shr:1 -- one VALU op, no LDS-pipe round trip (a ds_bpermute would cost ~100 cycles here)
#ifdef CIMG_LANE_PREV_BPERMUTE
    out.v = (T)__shfl_up((int)x.v, 1);
#else
    out.v = (T)__builtin_amdgcn_update_dpp((int)x.v, (int)x.v, 0x138, 0xF, 0xF, false);
#endif
}

}  // namespace cimg
#endif

// -DCIMG_PROFILE (diagnostic builds only): per-item cycle accounting written to the kernel's dbg buffer (16 uint64 per item)
#if defined(CIMG_PROFILE) && !defined(CIMG_EMULATE)
#define CIMG_PROF_DECL unsigned long long prof_t_ = cimg_cycles(); unsigned long long prof_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int prof_cnt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#ifndef CIMG_PROFILE_MASK
#define CIMG_PROFILE_MASK 0xFF
#endif
#define CIMG_PROF_LAP(i) do { if ((CIMG_PROFILE_MASK >> (i)) & 1) { const unsigned long long n_ = cimg_cycles(); prof_acc_[i] += n_ - prof_t_; prof_t_ = n_; } } while (0)
#define CIMG_PROF_COUNT(i) (++prof_cnt_[i])
#define CIMG_PROF_STORE(dbg, item) do { if (dbg && __lane_id() == 0) { for (int k_ = 0; k_ < 8; k_++) dbg[16 * (size_t)(item) + k_] = prof_acc_[k_]; for (int k_ = 0; k_ < 8; k_++) dbg[16 * (size_t)(item) + 8 + k_] = (unsigned long long)prof_cnt_[k_]; } } while (0)
#else
#define CIMG_PROF_DECL
#define CIMG_PROF_LAP(i) ((void)0)
#define CIMG_PROF_COUNT(i) ((void)0)
#define CIMG_PROF_STORE(dbg, item) ((void)0)
#endif

namespace cimg {

// ---- byte helpers shared by both builds ----------------------------------------------------------
CIMG_DEV uint32_t ld32u(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
CIMG_DEV int32_t ld32s(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; }
CIMG_DEV void st32(uint8_t* p, int32_t v) { memcpy(p, &v, 4); }

struct u128 { uint32_t x, y, z, w; };
CIMG_DEV u128 ld128u(const uint8_t* p) { u128 v; memcpy(&v, p, 16); return v; }      // any alignment
CIMG_DEV void st128u(uint8_t* p, const u128& v) { memcpy(p, &v, 16); }               // any alignment
CIMG_DEV u128 ld128a(const uint8_t* p) { return *reinterpret_cast<const u128*>(__builtin_assume_aligned(p, 16)); }
CIMG_DEV void st128a(uint8_t* p, const u128& v) { *reinterpret_cast<u128*>(__builtin_assume_aligned(p, 16)) = v; }

// transpose of an 8 x 8 bit matrix held as 8 bytes (byte i = row i, least significant byte first): bit k of byte i
// moves to bit i of byte k.  An involution; the three masked-swap rounds of the bitshuffle filter.
CIMG_DEV uint64_t bit_transpose8(uint64_t x)
{
    uint64_t t;
    t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;  x ^= t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x ^= t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x ^= t ^ (t << 28);
    return x;
}

// unaligned 32-bit read from LDS built from two aligned dword reads (always legal on the LDS path)
CIMG_DEV uint32_t lds_ld32u(const uint8_t* lds, int off)
{
    const int a = off & ~3;
    const uint32_t lo = *reinterpret_cast<const uint32_t*>(lds + a);
    const uint32_t hi = *reinterpret_cast<const uint32_t*>(lds + a + 4);
    return alignbyte(hi, lo, (uint32_t)off & 3u);
}

// unaligned 32-bit store to LDS: ONE ds_write_b32 on the GPU (gfx950 runs LDS in unaligned access mode)
CIMG_DEV void lds_st32u(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

CIMG_HD int imin(int a, int b) { return a < b ? a : b; }
CIMG_HD int imax(int a, int b) { return a > b ? a : b; }

}  // namespace cimg
