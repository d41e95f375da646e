// kernarg_preload.hip -- does this stack honour kernarg preloading (first kernel arguments delivered in user SGPRs at
// wave launch, -mllvm -amdgpu-kernarg-preload-count=N), and what does a wave save by it?
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=14 scripts/micro/kernarg_preload.hip -o /tmp/kp && /tmp/kp
// Each wave stamps s_memtime at entry and again once its first argument-dependent value exists; run with 64 and with
// 1024 groups of 256 threads (as config 2 / config 3 launch).  Build once WITH the flag and once without and compare.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k(const float* a, const float* b, float* c, unsigned long long* stamps, int n, int stride) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int probe = n;
    asm volatile("" : "+s"(probe));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < probe) c[i] = a[i] + b[(size_t)i * stride];
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    const int groups = 1024, threads = 256, n = groups * threads;
    float *a, *b, *c;
    unsigned long long* st;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&st, groups * 4 * 8);
    std::vector<float> ha(n), hb(n);
    for (int i = 0; i < n; ++i) { ha[i] = (float)i; hb[i] = 0.5f * i; }
    hipMemcpy(a, ha.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), n * 4, hipMemcpyHostToDevice);
    for (int g : {64, 1024}) {
        std::vector<double> med;
        bool ok = true;
        for (int rep = 0; rep < 20; ++rep) {
            hipMemset(c, 0, n * 4);
            hipLaunchKernelGGL(k, dim3(g), dim3(threads), 0, 0, a, b, c, st, g * threads, 1);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            std::vector<float> hc(g * threads);
            hipMemcpy(hc.data(), c, hc.size() * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < hc.size(); ++i) ok = ok && hc[i] == ha[i] + hb[i];
            std::vector<unsigned long long> hs(g * 4);
            hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
            std::sort(hs.begin(), hs.end());
            if (rep >= 4) med.push_back((double)hs[hs.size() / 2]);
        }
        std::sort(med.begin(), med.end());
        printf("%4d groups x 256 threads: results %s; entry -> first argument usable: median wave %.0f ticks\n", g, ok ? "correct" : "WRONG",
               med[med.size() / 2]);
    }
    return 0;
}
