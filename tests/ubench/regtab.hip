// regtab.hip -- what does a slot of a REGISTER-resident hash table cost?  Diagnostic only (round 5, encode_rt_kernel.h).
// One wave; a table of 64 VGPRs (two tuples of 32); slots addressed by a wave-uniform 13-bit hash from a scalar LCG.
//   0: the compiler's form: tuple[r] (s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off) + v_readlane + v_writelane + tuple[r] = ...
//   1: the read half of it only
//   2: v_readlane + v_writelane on ONE fixed register (no indexing): the floor
//   3: the scalar LCG + address arithmetic alone (loop overhead)
//   4: s_set_gpr_idx_on / s_set_gpr_idx_off pairs around a v_mov of a fixed register (what the mode switch itself costs)
//   5: LDS: ds_read_u16 + ds_write_b16 of the same slot by all lanes (the LDS-table form's slot access), dependent
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t v32 __attribute__((ext_vector_type(32)));
__device__ int wl_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

__device__ __forceinline__ unsigned long long stamp()
{
    unsigned long long c;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c) :: "memory");
    return c;
}
__device__ __forceinline__ uint32_t rt_xchg(v32& ta, v32& tb, uint32_t h, uint32_t pos)
{
    const uint32_t r = (h >> 7) & 31, lane = (h >> 1) & 63, sh = (h & 1) << 4;
    uint32_t old;
    if (h & 0x1000) {
        const uint32_t w = tb[r];
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)w, (int)lane);
        old = (d >> sh) & 0xFFFFu;
        tb[r] = (uint32_t)wl_i32((int)((d & ~(0xFFFFu << sh)) | (pos << sh)), (int)lane, (int)w);
    } else {
        const uint32_t w = ta[r];
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)w, (int)lane);
        old = (d >> sh) & 0xFFFFu;
        ta[r] = (uint32_t)wl_i32((int)((d & ~(0xFFFFu << sh)) | (pos << sh)), (int)lane, (int)w);
    }
    return old;
}
__device__ __forceinline__ uint32_t rt_get(const v32& ta, const v32& tb, uint32_t h)
{
    const uint32_t r = (h >> 7) & 31, lane = (h >> 1) & 63, sh = (h & 1) << 4;
    const uint32_t w = (h & 0x1000) ? tb[r] : ta[r];
    return ((uint32_t)__builtin_amdgcn_readlane((int)w, (int)lane) >> sh) & 0xFFFFu;
}

extern "C" __global__ void __launch_bounds__(64) k(uint64_t* res, int iters, uint32_t seed)
{
    __shared__ uint16_t tab[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) tab[i] = 0;
    __syncthreads();
    v32 ta = 0, tb = 0;
    uint32_t x = __builtin_amdgcn_readfirstlane(seed), acc = 0;
    uint64_t* out = res + (size_t)blockIdx.x * 16;
    unsigned long long c0;
    c0 = stamp();
#pragma unroll 1
    for (int i = 0; i < iters; i++) { x = x * 1664525u + 1013904223u; acc += rt_xchg(ta, tb, x >> 19, (uint32_t)i & 0xFFFF); }
    out[0] = stamp() - c0;
    c0 = stamp();
#pragma unroll 1
    for (int i = 0; i < iters; i++) { x = x * 1664525u + 1013904223u; acc += rt_get(ta, tb, x >> 19); }
    out[1] = stamp() - c0;
    uint32_t w = threadIdx.x;
    c0 = stamp();
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t lane = (x >> 20) & 63;
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)w, (int)lane);
        acc += d;
        w = (uint32_t)wl_i32((int)(d + i), (int)lane, (int)w);
    }
    out[2] = stamp() - c0;
    c0 = stamp();
#pragma unroll 1
    for (int i = 0; i < iters; i++) { x = x * 1664525u + 1013904223u; acc += (x >> 19) & 31; asm volatile("" : "+s"(acc)); }
    out[3] = stamp() - c0;
    c0 = stamp();
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        asm volatile("s_set_gpr_idx_on %1, 1\n\tv_mov_b32 %0, %0\n\ts_set_gpr_idx_off" : "+v"(w) : "s"(0));
    }
    out[4] = stamp() - c0;
    c0 = stamp();
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t h = (x >> 19) & 8191;
        const uint32_t o = ((volatile uint16_t*)tab)[h];
        ((volatile uint16_t*)tab)[h] = (uint16_t)i;
        x += __builtin_amdgcn_readfirstlane(o);      // dependent, as the encoder's next step is
    }
    out[5] = stamp() - c0;
    if (threadIdx.x == 0) out[15] = acc + w + ta[3] + tb[5] + x;
}

int main()
{
    const int iters = 4096;
    uint64_t* d;
    hipMalloc(&d, 16 * 8 * 2048);
    for (int grid : {1, 256 * 8}) {
        std::vector<uint64_t> h(16 * (size_t)grid);
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d, iters, 12345u);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d, iters, 12345u);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        const char* names[6] = {"xchg (indexed RMW)", "get (indexed read)", "readlane+writelane fixed", "loop + LCG alone", "idx_on/v_mov/idx_off", "LDS u16 read+write, dependent"};
        printf("grid %d (%s):\n", grid, grid == 1 ? "one wave alone" : "8 waves a CU");
        for (int v = 0; v < 6; v++) {
            double sum = 0;
            for (int b = 0; b < grid; b++) sum += (double)h[(size_t)b * 16 + v];
            printf("  %-32s %7.1f cycles per iteration\n", names[v], sum / grid / iters);
        }
    }
    return 0;
}
