// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the step kernel's access shape on gfx950: one dword per lane,
// 64 consecutive lanes = 256 contiguous bytes per wave instruction, ROWS independent row streams (struct-of-arrays).
// MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports 1/2 of the bytes of a 16-B-per-lane streaming read and other
// widths are uncalibrated.  This kernel moves a KNOWN number of bytes in the step kernel's shape; run it under
//   rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib <n_envs>     and     rocprofv3 --pmc WRITE_SIZE -- ./fetch_calib <n_envs>
// and divide the counter by the byte count printed here (scripts/summarize_pmc.py does that).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int ROWS = 32;
__global__ void soa_copy(const float* __restrict__ src, float* __restrict__ dst, long stride, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    float v[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) v[k] = src[k * stride + i];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) dst[k * stride + i] = v[k] + 1.0f;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1 << 20;
    const long stride = ((long)n + 63) / 64 * 64;
    float *a, *b;
    hipMalloc(&a, stride * ROWS * 4);
    hipMalloc(&b, stride * ROWS * 4);
    hipMemset(a, 0, stride * ROWS * 4);
    hipMemset(b, 0, stride * ROWS * 4);
    for (int it = 0; it < 20; ++it) soa_copy<<<(n + 63) / 64, 64>>>(a, b, stride, n);
    hipDeviceSynchronize();
    printf("fetch_calib: n=%d rows=%d read_bytes_per_launch=%ld write_bytes_per_launch=%ld\n", n, ROWS, (long)n * ROWS * 4, (long)n * ROWS * 4);
    return 0;
}
