// latency.hip -- single-wave cost model for gfx950: what one wave alone on a SIMD pays for dependent
// LDS reads, SALU/VALU chains, taken branches, readlane/ballot.  Diagnostic only (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ unsigned long long cyc()
{
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

extern "C" __global__ void __launch_bounds__(64) k(uint64_t* res, int iters, int seed)
{
    __shared__ uint32_t lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = (uint32_t)((i * 37 + 11) & 4095);
    __syncthreads();
    unsigned long long t0, t1;
    int r = 0;

    // 1. dependent LDS reads (pointer chase), per-lane addresses
    uint32_t p = (uint32_t)(lane + seed) & 4095;
    t0 = cyc();
    for (int i = 0; i < iters; i++) p = lds[p];
    t1 = cyc();
    if (lane == 0) res[r] = t1 - t0;
    r++;
    uint32_t sink = p;

    // 2. dependent LDS read -> readfirstlane -> uniform address (the scalar chain walk pattern)
    int u = seed & 4095;
    t0 = cyc();
    for (int i = 0; i < iters; i++) u = __builtin_amdgcn_readfirstlane((int)lds[u]);
    t1 = cyc();
    if (lane == 0) res[r] = t1 - t0;
    r++;
    sink += u;

    // 3. dependent VALU chain
    uint32_t x = lane + seed;
    t0 = cyc();
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        asm volatile("v_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\t"
                     "v_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1" : "+v"(x));
    }
    t1 = cyc();
    if (lane == 0) res[r] = t1 - t0;    // 8 per iter
    r++;
    sink += x;

    // 4. dependent SALU chain
    int s = seed;
    t0 = cyc();
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        asm volatile("s_add_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 3\n\t"
                     "s_add_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 3" : "+s"(s) :: "scc");
    }
    t1 = cyc();
    if (lane == 0) res[r] = t1 - t0;    // 8 per iter
    r++;
    sink += s;

    // 5. independent VALU (8 different registers)
    uint32_t a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3;
    t0 = cyc();
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        asm volatile("v_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %1, %1, 3, 1\n\tv_mad_u32_u24 %2, %2, 3, 1\n\tv_mad_u32_u24 %3, %3, 3, 1\n\t"
                     "v_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %1, %1, 3, 1\n\tv_mad_u32_u24 %2, %2, 3, 1\n\tv_mad_u32_u24 %3, %3, 3, 1"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    }
    t1 = cyc();
    if (lane == 0) res[r] = t1 - t0;
    r++;
    sink += a0 + a1 + a2 + a3;

    // 6. VALU compare -> SALU consumer (ballot pattern): v_cmp, s_ff1, v_readlane with that index
    uint32_t y = lane * 2654435761u + seed;
    int acc = 0;
    t0 = cyc();
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        const uint64_t m = __ballot((y >> (i & 15)) & 1);
        const int f = m ? __builtin_ctzll(m) : 0;
        acc += __builtin_amdgcn_readlane((int)y, f);
        y += acc;
    }
    t1 = cyc();
    if (lane == 0) res[r] = t1 - t0;
    r++;
    sink += acc + y;

    // 7. data-dependent uniform branches written in C (whatever the compiler makes of them)
    int b = seed;
    t0 = cyc();
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        if (b & 1) b = b * 3 + 1; else b = b >> 1;
        b = __builtin_amdgcn_readfirstlane(b);
        if (b & 2) b += 5; else b ^= 9;
        b = __builtin_amdgcn_readfirstlane(b);
        if (b & 4) b -= 7; else b |= 16;
        b = __builtin_amdgcn_readfirstlane(b);
        if (b & 8) b += 11; else b ^= 3;
        b = __builtin_amdgcn_readfirstlane(b);
    }
    t1 = cyc();
    if (lane == 0) res[r] = t1 - t0;
    r++;
    sink += b;

    // 8. LDS write then read-back of the same address (collision check pattern)
    uint32_t w = lane;
    t0 = cyc();
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        volatile uint32_t* q = lds;
        q[(w & 4095)] = w;
        w = q[(w & 4095)] + 1;
    }
    t1 = cyc();
    if (lane == 0) res[r] = t1 - t0;
    r++;
    sink += w;

    // 9. ds_bpermute chain
    int g = lane;
    t0 = cyc();
#pragma unroll 1
    for (int i = 0; i < iters; i++) g = __builtin_amdgcn_ds_bpermute(((g + 1) & 63) << 2, g);
    t1 = cyc();
    if (lane == 0) res[r] = t1 - t0;
    r++;
    sink += g;

    // 10. empty timer pair
    t0 = cyc();
    t1 = cyc();
    if (lane == 0) res[r] = t1 - t0;
    r++;
    if (sink == 0x12345678) res[31] = sink;
}

int main(int argc, char** argv)
{
    const int iters = 1000;
    uint64_t* d;
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("start\n");
    hipError_t e0 = hipMalloc(&d, 32 * 8 * 8);
    printf("malloc %d\n", (int)e0);
    const char* names[] = {"dep LDS read (per-lane addr)", "dep LDS read + readfirstlane", "dep VALU x8", "dep SALU x8", "indep VALU x8 (4 regs)",
                           "ballot+ctz+readlane chain", "4x uniform if/else", "LDS write + read-back", "ds_bpermute chain", "empty timer pair"};
    const double per[] = {1, 1, 8, 8, 8, 1, 4, 1, 1, 1};
    for (int grid : {1, 256, 1024, 2048}) {
        hipMemset(d, 0, 32 * 8 * 8);
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d, iters, 5);
        printf("launched %d\n", grid);
        hipError_t e1 = hipDeviceSynchronize();
        printf("sync %d\n", (int)e1);
        std::vector<uint64_t> h(32);
        hipMemcpy(h.data(), d, 32 * 8, hipMemcpyDeviceToHost);
        printf("grid %d (waves of 64, %d per CU if spread evenly)\n", grid, (grid + 255) / 256);
        for (int i = 0; i < 10; i++) printf("  %-34s %8.1f cycles per op\n", names[i], (double)h[i] / (i == 9 ? 1 : iters) / per[i]);
    }
    return 0;
}
