// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for this package's access pattern
// (8 bytes per lane, 512 contiguous bytes per wavefront instruction, many planes per thread)
// against a known byte count, cold (inputs far larger than the 256 MB on-die cache, each
// byte read once) and hot (the same 200 MB re-read), next to 16 B/lane streams.
// Build: hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/_bin/fetch_calib
// Run:   rocprofv3 --pmc FETCH_SIZE -- tools/_bin/fetch_calib   (and --pmc WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// 25 planes of n doubles read (8 B/lane), 1 written: 200 MB read at n = 1e6
__global__ void __launch_bounds__(256) calib_read8_cold(const double* __restrict__ in, double* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 25; ++k) acc += in[k * n + i];
    out[i] = acc;
}
__global__ void __launch_bounds__(256) calib_read8_hot(const double* __restrict__ in, double* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 25; ++k) acc += in[k * n + i];
    out[i] = acc;
}
__global__ void __launch_bounds__(256) calib_read16_cold(const double2* __restrict__ in, double* __restrict__ out, long n2) {
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
        double2 v = in[i]; acc += v.x + v.y;
    }
    if (acc == 123.456) out[0] = acc;
}
// 25 planes written, 8 B/lane: 200 MB
__global__ void __launch_bounds__(256) calib_write8(double* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int k = 0; k < 25; ++k) out[k * n + i] = (double)k;
}
int main() {
    const long n = 1000000, sets = 8;                  // 8 x 200 MB of inputs
    double *in, *out;
    CK(hipMalloc(&in, sets * 25 * n * 8)); CK(hipMalloc(&out, 25 * n * 8));
    CK(hipMemset(in, 0, sets * 25 * n * 8)); CK(hipMemset(out, 0, 25 * n * 8));
    CK(hipDeviceSynchronize());
    const int blk = 256, grid = (int)((n + blk - 1) / blk);
    for (int r = 0; r < 16; ++r) calib_read8_cold<<<grid, blk>>>(in + (r % sets) * 25 * n, out, n);
    for (int r = 0; r < 16; ++r) calib_read16_cold<<<8192, blk>>>((const double2*)(in + (r % sets) * 25 * n), out, 25 * n / 2);
    for (int r = 0; r < 16; ++r) calib_read8_hot<<<grid, blk>>>(in, out, n);
    for (int r = 0; r < 16; ++r) calib_write8<<<grid, blk>>>(out, n);
    CK(hipDeviceSynchronize());
    printf("known bytes per launch: calib_read8_cold / calib_read16_cold / calib_read8_hot read 200 MB "
           "(+8 MB written by the 8 B kernels), calib_write8 writes 200 MB\n");
    return 0;
}
