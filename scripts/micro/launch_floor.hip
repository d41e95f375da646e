// launch_floor.hip -- what the event-timed duration of a dispatch contains besides the kernel's own instructions.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
// Times, with the start / stop events of hipExtLaunchKernelGGL (what libdockauv's dockauv_time_steps uses), kernels that
// do (a) nothing, (b) spin for a given number of s_memtime ticks in every wave, for the grids of the BASELINE configs;
// (b) calibrates the tick (shader clock) against the event clock: duration(ticks) = floor + ticks / f.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k_empty(int* p) {
    if (p && threadIdx.x == 99999) p[0] = 1;
}
__global__ void k_spin(int* p, unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
    if (p && threadIdx.x == 99999) p[0] = 1;
}

static float median_us(std::vector<float>& v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2] * 1000.0f;
}

int main() {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int* d = nullptr;
    hipMalloc(&d, 64);
    hipStream_t s;
    hipStreamCreate(&s);
    printf("event-timed duration of one dispatch (median of 200, back-to-back launches on one stream), microseconds\n");
    for (int groups : {64, 512, 1024}) {
        for (int threads : {256, 512}) {
            std::vector<float> v;
            for (int i = 0; i < 220; ++i) {
                hipExtLaunchKernelGGL(k_empty, dim3(groups), dim3(threads), 0, s, e0, e1, 0, d);
                hipStreamSynchronize(s);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                if (i >= 20) v.push_back(ms);
            }
            printf("empty kernel    %5d groups x %3d threads: %6.2f us\n", groups, threads, median_us(v));
        }
    }
    for (unsigned long long ticks : {0ull, 2000ull, 4000ull, 8000ull, 16000ull}) {
        std::vector<float> v;
        for (int i = 0; i < 220; ++i) {
            hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, e0, e1, 0, d, ticks);
            hipStreamSynchronize(s);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (i >= 20) v.push_back(ms);
        }
        printf("spin %6llu ticks, 64 groups x 256 threads: %6.2f us\n", ticks, median_us(v));
    }
    // back-to-back without host synchronisation in between (as a rollout issues its steps): 200 launches, one sync
    for (int groups : {64, 1024}) {
        std::vector<hipEvent_t> a(200), b(200);
        for (auto& e : a) hipEventCreate(&e);
        for (auto& e : b) hipEventCreate(&e);
        for (int i = 0; i < 200; ++i) hipExtLaunchKernelGGL(k_empty, dim3(groups), dim3(256), 0, s, a[i], b[i], 0, d);
        hipStreamSynchronize(s);
        std::vector<float> v, gap;
        for (int i = 20; i < 200; ++i) {
            float ms = 0;
            hipEventElapsedTime(&ms, a[i], b[i]);
            v.push_back(ms);
            hipEventElapsedTime(&ms, b[i - 1], a[i]);
            gap.push_back(ms);
        }
        printf("queued, no sync %5d groups x 256 threads: kernel %6.2f us, gap to the next %6.2f us\n", groups, median_us(v), median_us(gap));
    }
    return 0;
}
