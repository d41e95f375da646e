// wave_census.hip -- which SIMD does wave w of workgroup b land on?  (placement of the integrating waves)
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/wave_census.hip -o /tmp/wave_census && /tmp/wave_census [threads] [blocks] [lds_bytes]
// Every wave records HW_REG_HW_ID (wave slot, SIMD, CU, SH, SE) and HW_REG_XCC_ID; the kernel spins ~20 us so that the
// whole grid is resident at once (as the step kernel's groups are).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

__global__ void census(unsigned* out, int spin) {
    extern __shared__ unsigned char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
        out[(blockIdx.x * nw + w) * 2 + 0] = hw;
        out[(blockIdx.x * nw + w) * 2 + 1] = xcc;
    }
    if (smem[threadIdx.x] == 77 && spin < 0) out[0] = 1;
}

int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 256, blocks = argc > 2 ? atoi(argv[2]) : 1024, lds = argc > 3 ? atoi(argv[3]) : 32768;
    const int nw = threads / 64;
    unsigned* d;
    hipMalloc(&d, (size_t)blocks * nw * 2 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    std::vector<unsigned> h((size_t)blocks * nw * 2);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(census, dim3(blocks), dim3(threads), lds, 0, d, 40000);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    // gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx950: se_id wider)
    std::map<unsigned long long, std::vector<std::pair<int, int>>> per_cu;   // (xcc, se, sh, cu) -> (block * 16 + wave, simd)
    long hist[4][4] = {};   // [wave index in block][simd]
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < nw; ++w) {
            const unsigned hw = h[(b * nw + w) * 2], xcc = h[(b * nw + w) * 2 + 1] & 0xf;
            const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            per_cu[((unsigned long long)xcc << 32) | (se << 16) | (sh << 8) | cu].push_back({b * 16 + w, simd});
            if (w < 4) hist[w][simd]++;
        }
    printf("threads %d blocks %d lds %d: %zu distinct (xcc, se, sh, cu)\n", threads, blocks, lds, per_cu.size());
    int shown = 0;
    for (auto& kv : per_cu) {
        if (shown++ >= 6) break;
        printf("xcc %llu se %llu sh %llu cu %2llu:", kv.first >> 32, (kv.first >> 16) & 0xff, (kv.first >> 8) & 0xff, kv.first & 0xff);
        std::sort(kv.second.begin(), kv.second.end());
        for (auto& e : kv.second) printf(" b%d.w%d->s%d", e.first / 16, e.first % 16, e.second);
        printf("\n");
    }
    // how often do the wave-0s (and wave-1s) of the blocks that share a CU collide on a SIMD?
    long cu_n = 0, worst0 = 0, sum_max0 = 0, sum_max01 = 0;
    for (auto& kv : per_cu) {
        int c0[4] = {}, c01[4] = {};
        for (auto& e : kv.second) {
            if (e.first % 16 == 0) c0[e.second]++;
            if (e.first % 16 <= 1) c01[e.second]++;
        }
        const int m0 = std::max(std::max(c0[0], c0[1]), std::max(c0[2], c0[3]));
        const int m01 = std::max(std::max(c01[0], c01[1]), std::max(c01[2], c01[3]));
        sum_max0 += m0; sum_max01 += m01; worst0 = std::max<long>(worst0, m0); ++cu_n;
    }
    printf("per CU: max number of wave-0s on one SIMD: mean %.2f worst %ld; of wave-0s + wave-1s: mean %.2f (blocks per CU %.2f)\n",
           (double)sum_max0 / cu_n, worst0, (double)sum_max01 / cu_n, (double)blocks / cu_n);
    printf("wave index -> SIMD histogram:\n");
    for (int w = 0; w < std::min(nw, 4); ++w) printf("  w%d: %ld %ld %ld %ld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    return 0;
}
