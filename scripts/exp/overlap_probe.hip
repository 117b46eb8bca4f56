// Probe (GPU): do two persistent, LDS-filling kernels launched on two streams share the chip when the first is in its
// tail?  Kernel: `grid` workgroups of 1024 lanes with `lds` bytes of dynamic LDS; every workgroup spins `short_us`,
// workgroup 0 spins `long_us` (the tail).  Prints the wall time of A alone, and of A and B launched back to back on one
// stream / on two streams.  hipcc --offload-arch=gfx950 -O2 scripts/exp/overlap_probe.hip -o scripts/exp/overlap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
__global__ __launch_bounds__(1024) void spin(long long short_ticks, long long long_ticks, int* sink) {
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    const long long want = blockIdx.x == 0 ? long_ticks : short_ticks;
    int acc = 0;
    while (wall_clock64() - t0 < want) acc += lds[(threadIdx.x + acc) & 1023];
    if (acc == 0x7fffffff) *sink = acc;
}
int main(int argc, char** argv) {
    const int grid = 256, lds = argc > 1 ? std::atoi(argv[1]) : 150 * 1024;
    const double short_us = 100., long_us = 600.;
    int* sink;
    CK(hipMalloc(&sink, 4));
    CK(hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const long long st = (long long)(short_us * rate_khz / 1e3), lt = (long long)(long_us * rate_khz / 1e3);
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto run = [&](int mode) {
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin, dim3(grid), dim3(1024), lds, s1, st, lt, sink);
        if (mode == 1) hipLaunchKernelGGL(spin, dim3(grid), dim3(1024), lds, s1, st, lt, sink);
        if (mode == 2) hipLaunchKernelGGL(spin, dim3(grid), dim3(1024), lds, s2, st, lt, sink);
        CK(hipDeviceSynchronize());
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    for (int w = 0; w < 3; ++w) run(2);
    for (int rep = 0; rep < 3; ++rep)
        std::printf("lds %d B: one kernel %.0f us, two on one stream %.0f us, two on two streams %.0f us (short %g us, tail %g us)\n", lds, run(0), run(1),
                    run(2), short_us, long_us);
    return 0;
}
