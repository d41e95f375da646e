// wave_placement.hip -- where does the hardware put the waves of co-resident groups?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/wave_placement.hip -o /tmp/wave_placement && /tmp/wave_placement [groups] [threads] [lds_kb]
// The step kernel gives the waves of a group different roles (one integrates, the others help), and a CU holds several
// groups at BASELINE sizes.  If wave w of every group of a CU lands on the same SIMD, the integrating waves of all those
// groups share one SIMD's issue slots while three SIMDs idle through the env phase.  This prints, for the launch shape of
// the step kernel, the SIMD of each wave index for the groups that share a CU (HW_ID / XCC_ID registers), and what the
// step kernel's role rotation (role = (wave + blockIdx) % waves) makes of it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <tuple>
#include <algorithm>

__global__ void k_where(unsigned* out, unsigned long long spin) {
    extern __shared__ unsigned char smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(4);   // all groups of the launch co-resident
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = hw;
        out[2 * w + 1] = xcc;
    }
    if (spin == 1) smem[threadIdx.x] = 0;
}

int main(int argc, char** argv) {
    const int groups = argc > 1 ? atoi(argv[1]) : 1024, threads = argc > 2 ? atoi(argv[2]) : 256, lds_kb = argc > 3 ? atoi(argv[3]) : 40;
    const int W = threads / 64;
    unsigned* d = nullptr;
    hipMalloc(&d, sizeof(unsigned) * 2 * groups * W);
    hipFuncSetAttribute((const void*)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    std::vector<unsigned> h(2 * groups * W);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_where, dim3(groups), dim3(threads), lds_kb * 1024, 0, d, 20000ull);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost);
    // CU key -> groups on it
    std::map<std::tuple<int, int, int, int>, std::vector<int>> cu_groups;
    auto simd_of = [&](int g, int w) { return (int)((h[2 * (g * W + w)] >> 4) & 3u); };
    for (int g = 0; g < groups; ++g) {
        const unsigned hw = h[2 * (g * W)], xcc = h[2 * (g * W) + 1] & 15u;
        cu_groups[{(int)xcc, (int)((hw >> 13) & 7u), (int)((hw >> 12) & 1u), (int)((hw >> 8) & 15u)}].push_back(g);
    }
    printf("%d groups x %d threads, %d KB LDS per group: %zu CUs used\n", groups, threads, lds_kb, cu_groups.size());
    int shown = 0;
    long same_plain = 0, same_rot = 0, pairs = 0;
    std::vector<int> hist_plain(5, 0), hist_rot(5, 0);
    for (auto& kv : cu_groups) {
        auto& gs = kv.second;
        if (shown < 6) {
            printf("  xcc %d se %d sh %d cu %2d:", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first));
            for (int g : gs) {
                printf("  group %4d simd of waves [", g);
                for (int w = 0; w < W; ++w) printf("%d", simd_of(g, w));
                printf("]");
            }
            printf("\n");
            ++shown;
        }
        // how many of the CU's integrating waves share the busiest SIMD: wave 0 integrates (plain) / role rotation
        std::vector<int> cnt_plain(4, 0), cnt_rot(4, 0);
        for (int g : gs) {
            cnt_plain[simd_of(g, 0)]++;
            // rotation: physical wave p takes role (p + g) % W; the integrating role 0 is taken by p = (W - g % W) % W
            cnt_rot[simd_of(g, (W - g % W) % W)]++;
        }
        hist_plain[std::min(4, *std::max_element(cnt_plain.begin(), cnt_plain.end()))]++;
        hist_rot[std::min(4, *std::max_element(cnt_rot.begin(), cnt_rot.end()))]++;
    }
    printf("integrating waves on the busiest SIMD of a CU (CUs with that count): wave 0 integrates : 1:%d 2:%d 3:%d 4+:%d\n", hist_plain[1], hist_plain[2], hist_plain[3], hist_plain[4]);
    printf("                                                                      role rotation     : 1:%d 2:%d 3:%d 4+:%d\n", hist_rot[1], hist_rot[2], hist_rot[3], hist_rot[4]);
    return 0;
}
