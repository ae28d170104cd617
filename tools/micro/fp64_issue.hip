// Microbenchmark (gfx950): issue rate of v_fma_f64 / v_fma_f32 per SIMD with 1, 2 and 4 resident
// wavefronts per SIMD, independent chains (no dependency stalls).  Prices the level-1 walks, which
// are fp64-issue bound.   hipcc --offload-arch=gfx950 -O3 fp64_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T>
__global__ void __launch_bounds__(64) chains(T* out, int iters, T a, T b) {
    T x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
            x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void __launch_bounds__(64) chains32(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// one dependent chain: the latency a lone wavefront sees
__global__ void __launch_bounds__(64) chain1(double* out, int iters, double a, double b) {
    double x = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) x = __builtin_fma(x, a, b);
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int simds = p.multiProcessorCount * 4;
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("%s: %d CUs, %d SIMDs, clock attribute %.0f MHz\n", p.name, p.multiProcessorCount, simds, clk / 1e3);
    double* out; hipMalloc(&out, sizeof(double) * 64 * simds * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int kind = 0; kind < 3; ++kind)
        for (int wps : {1, 2, 4}) {
            const int grid = simds * wps;
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) chains<double><<<grid, 64>>>(out, iters, 1.0000001, 1e-9);
                else if (kind == 1) chains32<<<grid, 64>>>((float*)out, iters, 1.0000001f, 1e-9f);
                else chain1<<<grid, 64>>>(out, iters / 8, 1.0000001, 1e-9);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double n = (double)(kind == 2 ? iters / 8 : iters) * 64 * wps;     // FMA instructions per SIMD
            const double ns = best * 1e6 / n;
            printf("%-24s %d wavefront(s)/SIMD: %8.3f ms  %.3f ns per wave64 FMA and SIMD = %.2f cycles at 2.4 GHz\n",
                   kind == 0 ? "fp64, 8 chains" : kind == 1 ? "fp32, 8 chains" : "fp64, 1 dependent chain",
                   wps, best, ns, ns * 2.4);
        }
    return 0;
}
