// clock.hip -- what does s_memtime count, and how fast does one wave issue?  Diagnostic only.
// For several grid sizes: a dependent VALU chain, an independent VALU stream, a SALU chain and an LDS pointer chase,
// each timed with s_memtime (shader clock?) and s_memrealtime (100 MHz).  Prints cycles per instruction by both clocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

__device__ __forceinline__ void stamps(unsigned long long& c, unsigned long long& r)
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r) :: "memory");
}

extern "C" __global__ void __launch_bounds__(64) k(uint64_t* res, int iters, int seed)
{
    __shared__ uint32_t lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = (uint32_t)((i * 37 + 11) & 4095);
    __syncthreads();
    unsigned long long c0, r0, c1, r1;
    uint64_t* out = res + (size_t)blockIdx.x * 16;
    uint32_t x = lane + seed, y = lane * 3 + seed, z = lane * 5, w = lane * 7;
    stamps(c0, r0);
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        asm volatile("v_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\t"
                     "v_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1" : "+v"(x));
    }
    stamps(c1, r1);
    if (lane == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
    stamps(c0, r0);
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        asm volatile("v_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %1, %1, 3, 1\n\tv_mad_u32_u24 %2, %2, 3, 1\n\tv_mad_u32_u24 %3, %3, 3, 1\n\t"
                     "v_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %1, %1, 3, 1\n\tv_mad_u32_u24 %2, %2, 3, 1\n\tv_mad_u32_u24 %3, %3, 3, 1" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
    }
    stamps(c1, r1);
    if (lane == 0) { out[2] = c1 - c0; out[3] = r1 - r0; }
    int s = seed;
    stamps(c0, r0);
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        asm volatile("s_add_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 3\n\t"
                     "s_add_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 3" : "+s"(s) :: "scc");
    }
    stamps(c1, r1);
    if (lane == 0) { out[4] = c1 - c0; out[5] = r1 - r0; }
    uint32_t p = (uint32_t)(lane + seed) & 4095;
    stamps(c0, r0);
    for (int i = 0; i < iters; i++) p = lds[p];
    stamps(c1, r1);
    if (lane == 0) { out[6] = c1 - c0; out[7] = r1 - r0; }
    // mixed: 1 SALU + 1 VALU alternating, independent of each other
    stamps(c0, r0);
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        asm volatile("s_add_i32 %0, %0, 3\n\tv_mad_u32_u24 %1, %1, 3, 1\n\ts_add_i32 %0, %0, 3\n\tv_mad_u32_u24 %1, %1, 3, 1\n\t"
                     "s_add_i32 %0, %0, 3\n\tv_mad_u32_u24 %1, %1, 3, 1\n\ts_add_i32 %0, %0, 3\n\tv_mad_u32_u24 %1, %1, 3, 1" : "+s"(s), "+v"(x) :: "scc");
    }
    stamps(c1, r1);
    if (lane == 0) { out[8] = c1 - c0; out[9] = r1 - r0; }
    // 32 dependent VALU per iteration: separates the per-instruction cost from the loop's own (3 scalar ops, one taken branch)
    stamps(c0, r0);
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++)
        asm volatile("v_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\t"
                     "v_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1\n\tv_mad_u32_u24 %0, %0, 3, 1" : "+v"(x));
    }
    stamps(c1, r1);
    if (lane == 0) { out[10] = c1 - c0; out[11] = r1 - r0; }
    // 8 TAKEN forward branches per iteration (scc is 1 after the compare)
    stamps(c0, r0);
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        asm volatile("s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:\ts_cbranch_scc1 2f\n\ts_nop 0\n2:\ts_cbranch_scc1 3f\n\ts_nop 0\n3:\ts_cbranch_scc1 4f\n\ts_nop 0\n4:\t"
                     "s_cbranch_scc1 5f\n\ts_nop 0\n5:\ts_cbranch_scc1 6f\n\ts_nop 0\n6:\ts_cbranch_scc1 7f\n\ts_nop 0\n7:\ts_cbranch_scc1 8f\n\ts_nop 0\n8:" ::: "scc");
    }
    stamps(c1, r1);
    if (lane == 0) { out[12] = c1 - c0; out[13] = r1 - r0; }
    // 8 NOT-taken branches per iteration
    stamps(c0, r0);
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        asm volatile("s_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:\ts_cbranch_scc1 2f\n\ts_nop 0\n2:\ts_cbranch_scc1 3f\n\ts_nop 0\n3:\ts_cbranch_scc1 4f\n\ts_nop 0\n4:\t"
                     "s_cbranch_scc1 5f\n\ts_nop 0\n5:\ts_cbranch_scc1 6f\n\ts_nop 0\n6:\ts_cbranch_scc1 7f\n\ts_nop 0\n7:\ts_cbranch_scc1 8f\n\ts_nop 0\n8:" ::: "scc");
    }
    stamps(c1, r1);
    if (lane == 0) { out[14] = c1 - c0; out[15] = r1 - r0; res[(size_t)gridDim.x * 16 + blockIdx.x] = x + y + z + w + s + p; }
}

int main()
{
    const int iters = 20000;
    uint64_t* d;
    const int maxg = 8192;
    hipMalloc(&d, (size_t)maxg * 17 * 8);
    std::vector<uint64_t> h((size_t)maxg * 16);
    for (int rep = 0; rep < 1; rep++)
    for (int grid : {1, 1024, 2048, 4096}) {
        hipMemset(d, 0, (size_t)maxg * 16 * 8);
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d, iters, 1);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, (size_t)grid * 16 * 8, hipMemcpyDeviceToHost);
        const char* names[8] = {"valu dep", "valu indep4", "salu dep", "lds chase", "salu+valu", "valu dep x32", "8 taken br", "8 untaken br"};
        printf("grid %5d:", grid);
        for (int t = 0; t < 8; t++) {
            std::vector<double> cyc, clk;
            for (int b = 0; b < grid; b++) {
                const double c = (double)h[(size_t)b * 16 + 2 * t], r = (double)h[(size_t)b * 16 + 2 * t + 1];
                const double n = (t == 3 ? 1.0 : (t == 5 ? 32.0 : (t >= 6 ? 1.0 : 8.0))) * iters;
                cyc.push_back(c / n);
                clk.push_back(c / r * 100.0);     // MHz if s_memtime counts shader cycles
            }
            std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
            printf("  %s %.2f cyc (memtime/realtime*100MHz = %.0f)", names[t], cyc[cyc.size() / 2], clk[clk.size() / 2]);
        }
        printf("\n");
    }
    return 0;
}
