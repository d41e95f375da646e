// dispatch_ramp.hip -- how long does the dispatcher take to put a whole grid on the chip?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/dispatch_ramp.hip -o /tmp/dispatch_ramp && /tmp/dispatch_ramp
// At BASELINE's batch sizes every group of the step kernel is resident at once, so a launch lasts
//   (first wave starts -> last wave starts)  +  the life of the last groups.
// This measures the first term for the step kernel's grid shapes: every wave records the shader clock (s_memtime, one
// clock per XCD) and the constant 100 MHz clock (s_memrealtime, shared by the XCDs) when it starts, spins for a given
// number of ticks, and records them again.  Reported per shape: the event-timed duration (what bench.py reports), and per
// XCD the spread of the wave starts and of the wave ends.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

struct Rec {
    unsigned long long t0, t1, rt0, rt1;
    unsigned xcc, cu;
};

template <int VGPRS>
__global__ void k_spin(Rec* out, unsigned long long ticks) {
    extern __shared__ unsigned char smem[];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    if (VGPRS >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (VGPRS >= 64 && VGPRS < 128) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        Rec r;
        r.t0 = t0; r.t1 = t1; r.rt0 = rt0; r.rt1 = rt1; r.xcc = xcc & 0xf; r.cu = hwid;
        out[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;
    }
    if (ticks == 0xdeadbeefull) smem[threadIdx.x] = 1;
}

template <int VGPRS>
static void run(int groups, int threads, size_t lds, unsigned long long ticks, Rec* d, hipStream_t s) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int waves = groups * threads / 64;
    std::vector<Rec> h(waves);
    std::vector<float> dur;
    std::vector<double> ramp, endspread, total, rt_total;
    for (int it = 0; it < 60; ++it) {
        hipExtLaunchKernelGGL(k_spin<VGPRS>, dim3(groups), dim3(threads), lds, s, e0, e1, 0, d, ticks);
        hipStreamSynchronize(s);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (it < 10) continue;
        dur.push_back(ms * 1000.f);
        hipMemcpy(h.data(), d, sizeof(Rec) * waves, hipMemcpyDeviceToHost);
        // per XCD: first start, last start, last end (shader clock of that XCD)
        unsigned long long f0[16], l0[16], l1[16];
        bool seen[16] = {};
        unsigned long long rmin = ~0ull, rmax = 0;
        for (const Rec& r : h) {
            const unsigned x = r.xcc;
            if (!seen[x]) { seen[x] = true; f0[x] = r.t0; l0[x] = r.t0; l1[x] = r.t1; }
            f0[x] = std::min(f0[x], r.t0); l0[x] = std::max(l0[x], r.t0); l1[x] = std::max(l1[x], r.t1);
            rmin = std::min(rmin, r.rt0); rmax = std::max(rmax, r.rt1);
        }
        double a = 0, b = 0, c = 0;
        for (int x = 0; x < 16; ++x)
            if (seen[x]) { a = std::max(a, (double)(l0[x] - f0[x])); c = std::max(c, (double)(l1[x] - f0[x])); }
        ramp.push_back(a); total.push_back(c);
        rt_total.push_back((double)(rmax - rmin) * 10.0);   // 100 MHz ticks -> ns
        (void)b;
    }
    auto med = [](auto& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%5d groups x %3d threads (%5d waves), lds %6zu B, vgprs %3d, spin %6llu: event %6.2f us | per XCD: first->last wave start %6.0f ticks, first start->last end %6.0f ticks | all XCDs (100 MHz clock) %6.2f us\n",
           groups, threads, waves, lds, VGPRS, ticks, med(dur), med(ramp), med(total), med(rt_total) / 1000.0);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    Rec* d = nullptr;
    hipMalloc(&d, sizeof(Rec) * 1024 * 1024);
    hipStream_t s;
    hipStreamCreate(&s);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spin<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spin<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const unsigned long long T = 12000;
    printf("-- the step kernel's shapes (config 3: 1024 x 256, 40 KB; config 4: 512 x 512, 38 KB; config 2: 64 x 256)\n");
    run<128>(1024, 256, 40960, T, d, s);
    run<128>(1024, 256, 40960, 0, d, s);
    run<128>(512, 512, 38912, T, d, s);
    run<128>(64, 256, 9000, T, d, s);
    printf("-- same envs (65 536), other group shapes\n");
    run<128>(1024, 64, 40960, T, d, s);
    run<128>(1024, 128, 40960, T, d, s);
    run<128>(512, 256, 81920, T, d, s);
    run<128>(512, 512, 81920, T, d, s);
    run<128>(256, 512, 163840, T, d, s);
    run<128>(256, 1024, 163840, T, d, s);
    run<128>(2048, 128, 20480, T, d, s);
    run<128>(4096, 64, 10240, T, d, s);
    printf("-- what the ramp depends on: LDS, VGPRs\n");
    run<128>(1024, 256, 0, T, d, s);
    run<32>(1024, 256, 40960, T, d, s);
    run<32>(1024, 256, 0, T, d, s);
    run<32>(4096, 64, 0, T, d, s);
    printf("-- more groups than fit at once (two rounds)\n");
    run<128>(2048, 256, 40960, T, d, s);
    return 0;
}
