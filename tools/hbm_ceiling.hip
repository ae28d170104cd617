// HBM ceiling probe for the F+J sweep's access pattern (MI355X / gfx950).
// Streams the same byte counts as tfk_sweep_fj on the film model at N = 1e6
// (3 state planes read, 3 + 22 planes written, 8 MB each) with trivial
// arithmetic, next to plain fill / copy / read kernels, so that the sweep's
// roofline fraction can be compared with what the memory system delivers for
// this mix.  Build: hipcc -O3 --offload-arch=gfx950 tools/hbm_ceiling.hip -o tools/_bin/hbm_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NR, int NW>
__global__ void __launch_bounds__(256) planes(const double* __restrict__ in, double* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NR; ++k) acc += in[k * n + i];
#pragma unroll
    for (int k = 0; k < NW; ++k) out[k * n + i] = acc + k;
}
// a thread walks SEG consecutive rows of width `stride` (the sweep's layout: node-major, chunk-minor)
template <int NR, int NW>
__global__ void __launch_bounds__(256) planes_walk(const double* __restrict__ in, double* __restrict__ out, long n, int stride, int seg) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long col = t % stride, row0 = (t / stride) * seg;
    for (int r = 0; r < seg; ++r) {
        const long i = (row0 + r) * stride + col;
        if (i >= n) return;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NR; ++k) acc += in[k * n + i];
#pragma unroll
        for (int k = 0; k < NW; ++k) out[k * n + i] = acc + k;
    }
}
// level-1 back-substitution pattern: few threads (one per chunk), each walking `rows`
// nodes upwards, NR planes read and NW written per node, next node requested ahead
template <int NR, int NW>
__global__ void __launch_bounds__(64) chunk_walk(const double* __restrict__ in, double* __restrict__ out, long n, int stride, int rows) {
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= stride) return;
    double cur[NR], nxt[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) cur[k] = in[k * n + (long)(rows - 1) * stride + col];
    double carry = 0.0;
    for (int r = rows - 1; r >= 0; --r) {
        if (r > 0) {
#pragma unroll
            for (int k = 0; k < NR; ++k) nxt[k] = in[k * n + (long)(r - 1) * stride + col];
        }
        double acc = carry;
#pragma unroll
        for (int k = 0; k < NR; ++k) acc = fma(cur[k], 1.0000001, acc);
        carry = acc * 0.5;
#pragma unroll
        for (int k = 0; k < NW; ++k) out[k * n + (long)r * stride + col] = acc + k;
#pragma unroll
        for (int k = 0; k < NR; ++k) cur[k] = nxt[k];
    }
}
// the same walk over a node-major layout inside a group of 64 chunks, [group][row][plane][lane]:
// a wavefront's stream is contiguous (NR * 512 bytes per row, rows back to back)
template <int NR, int NW>
__global__ void __launch_bounds__(64) chunk_walk_nm(const double* __restrict__ in, double* __restrict__ out, long n, int stride, int rows) {
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= stride) return;
    const long g = blockIdx.x, lane = threadIdx.x;
    auto at = [&](int r, int k, int npl) { return ((g * rows + r) * npl + k) * 64 + lane; };
    double cur[NR], nxt[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) cur[k] = in[at(rows - 1, k, NR)];
    double carry = 0.0;
    for (int r = rows - 1; r >= 0; --r) {
        if (r > 0) {
#pragma unroll
            for (int k = 0; k < NR; ++k) nxt[k] = in[at(r - 1, k, NR)];
        }
        double acc = carry;
#pragma unroll
        for (int k = 0; k < NR; ++k) acc = fma(cur[k], 1.0000001, acc);
        carry = acc * 0.5;
#pragma unroll
        for (int k = 0; k < NW; ++k) out[at(r, k, NW)] = acc + k;
#pragma unroll
        for (int k = 0; k < NR; ++k) cur[k] = nxt[k];
    }
}
__global__ void __launch_bounds__(256) fill2(double2* __restrict__ out, long n2) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x)
        out[i] = make_double2(1.0, 2.0);
}
__global__ void __launch_bounds__(256) copy2(const double2* __restrict__ in, double2* __restrict__ out, long n2) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x)
        out[i] = in[i];
}
__global__ void __launch_bounds__(256) read2(const double2* __restrict__ in, double* __restrict__ out, long n2) {
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
        double2 v = in[i]; acc += v.x + v.y;
    }
    if (acc == 123.456) out[0] = acc;
}

template <class F> static double time_ms(F f, int reps = 30) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b));
    CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const long n = 1000000;               // nodes per plane
    double *in, *out;
    CK(hipMalloc(&in, 32 * n * 8)); CK(hipMalloc(&out, 32 * n * 8));
    CK(hipMemset(in, 0, 32 * n * 8)); CK(hipMemset(out, 0, 32 * n * 8));
    CK(hipDeviceSynchronize());
    const int blk = 256; const int grid = (int)((n + blk - 1) / blk);
    auto report = [&](const char* name, double bytes, double ms) {
        printf("%-44s %8.2f us  %8.1f GB/s\n", name, ms * 1e3, bytes / ms * 1e-6);
    };
    double ms;
    ms = time_ms([&] { planes<3, 25><<<grid, blk>>>(in, out, n); });
    report("planes r3 w25 (sweep mix, 1 node/thread)", 28.0 * n * 8, ms);
    ms = time_ms([&] { planes<0, 25><<<grid, blk>>>(in, out, n); });
    report("planes r0 w25 (write only)", 25.0 * n * 8, ms);
    ms = time_ms([&] { planes<25, 1><<<grid, blk>>>(in, out, n); });
    report("planes r25 w1 (read mostly)", 26.0 * n * 8, ms);
    ms = time_ms([&] { planes<12, 12><<<grid, blk>>>(in, out, n); });
    report("planes r12 w12", 24.0 * n * 8, ms);
    for (int seg : {2, 4, 8, 16}) {
        const int stride = 31250;
        const long threads = (long)stride * ((n / stride + seg - 1) / seg);
        const int g2 = (int)((threads + blk - 1) / blk);
        ms = time_ms([&] { planes_walk<3, 25><<<g2, blk>>>(in, out, n, stride, seg); });
        char nm[64]; snprintf(nm, 64, "planes_walk r3 w25 seg=%d", seg);
        report(nm, 28.0 * n * 8, ms);
    }
    CK(hipFree(in)); CK(hipFree(out));
    CK(hipMalloc(&in, 40 * n * 8)); CK(hipMalloc(&out, 32 * n * 8));
    CK(hipMemset(in, 0, 40 * n * 8)); CK(hipMemset(out, 0, 32 * n * 8));
    CK(hipDeviceSynchronize());
    for (int rows : {32, 16, 8}) {
        const int stride = (int)(n / rows);
        const int g2 = (stride + 63) / 64;
        ms = time_ms([&] { chunk_walk<39, 3><<<g2, 64>>>(in, out, n, stride, rows); });
        char nm[64]; snprintf(nm, 64, "chunk_walk r39 w3, %d threads x %d rows", stride, rows);
        report(nm, 42.0 * n * 8, ms);
        ms = time_ms([&] { chunk_walk<22, 3><<<g2, 64>>>(in, out, n, stride, rows); });
        snprintf(nm, 64, "chunk_walk r22 w3, %d threads x %d rows", stride, rows);
        report(nm, 25.0 * n * 8, ms);
    }
    {   // the same walks over four alternating input sets (1.3 GB): nothing is left in the
        // 256 MB on-die cache from the previous pass, as in a time step
        double* big;
        CK(hipMalloc(&big, 4 * 40 * n * 8)); CK(hipMemset(big, 0, 4 * 40 * n * 8)); CK(hipDeviceSynchronize());
        int turn = 0;
        for (int rows : {32}) {
            const int stride = (int)(n / rows);
            const int g2 = (stride + 63) / 64;
            ms = time_ms([&] { chunk_walk<39, 3><<<g2, 64>>>(big + (long)(turn++ % 4) * 40 * n, out, n, stride, rows); }, 32);
            report("chunk_walk r39 w3, cold inputs (4 sets)", 42.0 * n * 8, ms);
            ms = time_ms([&] { chunk_walk<22, 3><<<g2, 64>>>(big + (long)(turn++ % 4) * 40 * n, out, n, stride, rows); }, 32);
            report("chunk_walk r22 w3, cold inputs (4 sets)", 25.0 * n * 8, ms);
        }
        for (int rows : {32}) {
            const int stride = (int)(n / rows) / 64 * 64;          // whole groups
            const int g2 = stride / 64;
            ms = time_ms([&] { chunk_walk_nm<39, 3><<<g2, 64>>>(big + (long)(turn++ % 4) * 40 * n, out, n, stride, rows); }, 32);
            report("chunk_walk r39 w3, node-major groups, cold inputs", 42.0 * stride * rows * 8, ms);
            ms = time_ms([&] { chunk_walk_nm<22, 3><<<g2, 64>>>(big + (long)(turn++ % 4) * 40 * n, out, n, stride, rows); }, 32);
            report("chunk_walk r22 w3, node-major groups, cold inputs", 25.0 * stride * rows * 8, ms);
            ms = time_ms([&] { chunk_walk_nm<10, 18><<<g2, 64>>>(big + (long)(turn++ % 4) * 40 * n, big + (long)((turn + 1) % 4) * 40 * n, n, stride, rows); }, 32);
            report("chunk_walk r10 w18 (factorisation mix), node-major, cold", 28.0 * stride * rows * 8, ms);
            const int stride0 = (int)(n / rows), g0 = (stride0 + 63) / 64;
            ms = time_ms([&] { chunk_walk<10, 18><<<g0, 64>>>(big + (long)(turn++ % 4) * 40 * n, big + (long)((turn + 1) % 4) * 40 * n, n, stride0, rows); }, 32);
            report("chunk_walk r10 w18 (factorisation mix), planes, cold", 28.0 * n * 8, ms);
        }
        const int blk2 = 256; const int gridp = (int)((n + blk2 - 1) / blk2);
        ms = time_ms([&] { planes<25, 1><<<gridp, blk2>>>(big + (long)(turn++ % 4) * 40 * n, out, n); }, 32);
        report("planes r25 w1, cold inputs (4 sets)", 26.0 * n * 8, ms);
        ms = time_ms([&] { planes<3, 25><<<gridp, blk2>>>(big + (long)(turn++ % 4) * 40 * n, big + (long)((turn + 1) % 4) * 40 * n, n); }, 32);
        report("planes r3 w25, rotating outputs (4 sets)", 28.0 * n * 8, ms);
        CK(hipFree(big));
    }
    const long n2 = 25 * n / 2;
    for (int g : {2048, 8192, 32768}) {
        ms = time_ms([&] { fill2<<<g, blk>>>((double2*)out, n2); });
        char nm[64]; snprintf(nm, 64, "fill 200 MB, 16 B/lane, grid=%d", g); report(nm, 25.0 * n * 8, ms);
        ms = time_ms([&] { copy2<<<g, blk>>>((const double2*)in, (double2*)out, n2 / 2); });
        snprintf(nm, 64, "copy 100->100 MB, 16 B/lane, grid=%d", g); report(nm, 25.0 * n * 8, ms);
        ms = time_ms([&] { read2<<<g, blk>>>((const double2*)in, out, n2); });
        snprintf(nm, 64, "read 200 MB, 16 B/lane, grid=%d", g); report(nm, 25.0 * n * 8, ms);
    }
    ms = time_ms([&] { CK(hipMemsetAsync(out, 0, 25 * n * 8)); });
    report("hipMemsetAsync 200 MB", 25.0 * n * 8, ms);
    ms = time_ms([&] { CK(hipMemcpyAsync(out, in, 25 * n * 4, hipMemcpyDeviceToDevice)); });
    report("hipMemcpyAsync D2D 100->100 MB", 25.0 * n * 8, ms);
    CK(hipFree(in)); CK(hipFree(out));
    return 0;
}
