// launch.hip -- how fast do workgroups START?  Every workgroup stamps the 100 MHz clock when its first wave begins, then
// holds its slot for ~30 us.  Diagnostic only: the two-wave lean decode launch (decode_pair.h) showed half of its
// workgroups starting 11 us late.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

extern "C" __global__ void k(uint64_t* res, int hold_ticks)
{
    extern __shared__ uint8_t lds[];
    unsigned long long t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (threadIdx.x == 0) { res[blockIdx.x] = t0; lds[0] = 1; }
    unsigned long long t = t0;
    while (t - t0 < (unsigned long long)hold_ticks) { asm volatile("s_sleep 8\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); }
    __syncthreads();
}

int main()
{
    const int grid = 4096;
    uint64_t* d;
    (void)hipMalloc(&d, grid * 8);
    std::vector<uint64_t> h(grid);
    for (int threads : {64}) for (int lds : {16992, 19184, 28000, 31744, 32000, 32256, 32512, 32768, 33000}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, d, 3000);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        auto us = [&](double q) { return (double)(h[(size_t)(q * (grid - 1))] - h[0]) / 100.0; };
        int early = 0; for (int i = 0; i < grid; i++) early += (h[i] - h[0]) < 500;
        printf("threads %3d lds %5d: workgroups started within 5 us: %d (%.2f per CU);  start percentiles us  25%% %.1f  50%% %.1f  75%% %.1f  100%% %.1f\n", threads, lds, early, early / 256.0, us(0.25), us(0.5), us(0.75), us(1.0));
    }
    return 0;
}
