// writeback_cost.hip -- what does a kernel pay at its END for the bytes it wrote?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/writeback_cost.hip -o /tmp/writeback_cost && /tmp/writeback_cost
// MI355X has one L2 per XCD; a kernel's release at its end has to make its stores visible device-wide, i.e. write the
// dirty lines of eight L2s back.  profiles/r3/span_diagnostic_build.txt: the event-timed duration of the step kernel exceeds
// "first group starts -> last group has ended" by 1.0 us when the launch writes next to nothing and by 2.4-3.0 us when it
// writes 15+ MB.  This isolates that term: 1024 groups x 256 threads write B bytes per launch as coalesced rows (one dword
// per lane and row, the step kernel's shape) and end; variants = the cache policy of the stores.  Reported: event-timed
// duration, and first wave start -> last wave end on the 100 MHz clock (all XCDs).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

enum Mode { PLAIN = 0, NONTEMPORAL = 1, AGENT = 2, SYSTEM = 3 };

template <int MODE>
__device__ __forceinline__ void st(float* p, float v) {
    if (MODE == PLAIN) *p = v;
    else if (MODE == NONTEMPORAL) __builtin_nontemporal_store(v, p);
    else if (MODE == AGENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// rows of `stride` floats; group g writes its 64-float-per-wave slices of `rows` rows; spin = ticks of arithmetic first
template <int MODE>
__global__ void k_write(float* out, long stride, int rows, unsigned long long spin, unsigned long long* clk) {
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(1);
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < rows; ++r) st<MODE>(out + (long)r * stride + col, (float)r);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const long w = (long)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        clk[2 * w] = rt0;
        clk[2 * w + 1] = rt1;
    }
}

__global__ void k_read(const float* in, long stride, int rows, float* sink) {
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (int r = 0; r < rows; ++r) acc += in[(long)r * stride + col];
    if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE>
static void run(const char* name, int rows, unsigned long long spin, float* d, unsigned long long* clk, hipStream_t s, bool then_read) {
    const int groups = 1024, threads = 256;
    const long stride = (long)groups * threads;
    const int waves = groups * threads / 64;
    hipEvent_t e0, e1, e2, e3;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
    std::vector<float> dur, span, rdur;
    std::vector<unsigned long long> h(2 * waves);
    float* sink = d + stride * 64;
    for (int it = 0; it < 60; ++it) {
        hipExtLaunchKernelGGL(k_write<MODE>, dim3(groups), dim3(threads), 0, s, e0, e1, 0, d, stride, rows, spin, clk);
        if (then_read) hipExtLaunchKernelGGL(k_read, dim3(groups), dim3(threads), 0, s, e2, e3, 0, d, stride, rows, sink);
        hipStreamSynchronize(s);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (it < 10) continue;
        dur.push_back(ms * 1000.f);
        if (then_read) { hipEventElapsedTime(&ms, e2, e3); rdur.push_back(ms * 1000.f); }
        hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost);
        unsigned long long a = ~0ull, b = 0;
        for (int w = 0; w < waves; ++w) { a = std::min(a, h[2 * w]); b = std::max(b, h[2 * w + 1]); }
        span.push_back((float)(b - a) / 100.f);
    }
    auto med = [](std::vector<float>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.f : v[v.size() / 2]; };
    const double mb = (double)rows * stride * 4 / 1e6;
    printf("%-12s %6.1f MB written, spin %5llu: event-timed %6.2f us | first wave start -> last wave end (stores acknowledged) %6.2f us | outside the waves %5.2f us",
           name, mb, spin, med(dur), med(span), med(dur) - med(span));
    if (then_read) printf(" | a kernel reading the same bytes next: %6.2f us", med(rdur));
    printf("\n");
}

int main() {
    const long stride = 1024L * 256;
    float* d = nullptr;
    hipMalloc(&d, sizeof(float) * stride * 65);
    unsigned long long* clk = nullptr;
    hipMalloc(&clk, sizeof(unsigned long long) * 2 * 4096);
    hipStream_t s;
    hipStreamCreate(&s);
    for (unsigned long long spin : {0ull, 8000ull}) {
        for (int rows : {0, 4, 14, 32}) {
            run<PLAIN>("plain", rows, spin, d, clk, s, false);
            if (rows == 0) continue;
            run<NONTEMPORAL>("nontemporal", rows, spin, d, clk, s, false);
            run<AGENT>("agent scope", rows, spin, d, clk, s, false);
            run<SYSTEM>("system scope", rows, spin, d, clk, s, false);
        }
    }
    printf("-- followed by a reader of the same bytes (the next step reads the state this step wrote)\n");
    run<PLAIN>("plain", 14, 8000, d, clk, s, true);
    run<NONTEMPORAL>("nontemporal", 14, 8000, d, clk, s, true);
    run<AGENT>("agent scope", 14, 8000, d, clk, s, true);
    return 0;
}
