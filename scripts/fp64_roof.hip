// FP64 VALU issue-rate microbenchmark for gfx950 ("state the measured roof next to the spec").
// Each kernel runs long chains of one instruction class on registers only; 8 independent chains per lane give the
// scheduler all the ILP a single wavefront can use.  Reported: G instr-lanes/s and TFLOP/s (1 flop per mul/add, 2 per fma).
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/fp64_roof.hip -o scripts/fp64_roof
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHAINS 8
#define ITERS 4096

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double a, double b) {
    double x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x * 1e-9;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) x[c] = x[c] * b;                                   // v_mul_f64
            else if (OP == 1) x[c] = x[c] + b;                              // v_add_f64
            else if (OP == 2) x[c] = __builtin_fma(x[c], b, a);             // v_fma_f64
            else if (OP == 3) x[c] = x[c] * b + a;                          // mul + add, not contracted: the path's mix
            else if (OP == 4) x[c] = fmin(x[c], b) + a;                     // v_min_f64 + v_add_f64
            else if (OP == 5) x[c] = a / x[c];                              // IEEE division sequence
            else if (OP == 6) x[c] = sqrt(x[c]) + a;                        // IEEE sqrt sequence
            else if (OP == 7) x[c] = (double)((float)x[c] * 1.0001f);       // cvt f64<->f32 + v_mul_f32
        }
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
__global__ __launch_bounds__(256) void kf(float* out, float a, float b) {
    float x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x * 1e-6f;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = x[c] * b + a;               // v_mul_f32 + v_add_f32
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K, class T>
double run(K kern, T* out, T a, T b, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, a, b);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, a, b);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int waves_per_simd = 8;
    const int blocks = p.multiProcessorCount * waves_per_simd;  // 256-thread blocks = 4 waves: one per SIMD
    double* out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    const double lanes = double(blocks) * 256, ops = lanes * ITERS * CHAINS;
    const char* names[] = {"v_mul_f64", "v_add_f64", "v_fma_f64", "mul+add (no contraction)", "v_min_f64+v_add_f64", "div f64 (IEEE)", "sqrt f64 (IEEE)+add", "cvt f64->f32, mul f32, cvt back"};
    const double flops_per[] = {1, 1, 2, 2, 1, 1, 2, 1};
    const double instr_per[] = {1, 1, 1, 2, 2, 1, 2, 3};
    double ms[8];
    ms[0] = run(k<0>, out, 1.0, 1.0000001, blocks); ms[1] = run(k<1>, out, 1.0, 1e-7, blocks);
    ms[2] = run(k<2>, out, 1e-7, 0.9999999, blocks); ms[3] = run(k<3>, out, 1e-7, 0.9999999, blocks);
    ms[4] = run(k<4>, out, 1e-7, 5.0, blocks); ms[5] = run(k<5>, out, 1.5, 0.0, blocks);
    ms[6] = run(k<6>, out, 1.5, 0.0, blocks); ms[7] = run(k<7>, out, 1.5, 0.0, blocks);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"waves_per_simd\": %d, \"rows\": [\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000, waves_per_simd);
    for (int i = 0; i < 8; ++i) {
        double t = ms[i] * 1e-3;
        printf("  {\"op\": \"%s\", \"ms\": %.4f, \"tflops\": %.2f, \"g_statements_per_s\": %.1f, \"cycles_per_wave_instr_at_clock\": %.2f}%s\n", names[i], ms[i],
               ops * flops_per[i] / t / 1e12, ops / t / 1e9,
               (t * p.clockRate * 1e3) / (double(ITERS) * CHAINS * instr_per[i] * waves_per_simd), i == 7 ? "" : ",");
    }
    float* outf = (float*)out;
    double msf = run(kf<0>, outf, 1e-6f, 0.999f, blocks);
    printf("], \"f32_mul_add\": {\"ms\": %.4f, \"tflops\": %.2f}}\n", msf, ops * 2 / (msf * 1e-3) / 1e12);
    return 0;
}
