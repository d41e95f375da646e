// vmem_issue.hip -- what a wave pays to ISSUE global loads / stores on gfx950, and until the data is back.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/vmem_issue.hip -o /tmp/vmem_issue && /tmp/vmem_issue
// One workgroup on one CU; each wave issues 32 independent global loads (or stores) of a row-per-instruction SoA
// pattern (lane l reads base + k * pitch + l * width: what the step kernel's state loads look like), bracketed by
// s_memtime: cycles until the last one has ISSUED (no waitcnt), and until all have landed (s_waitcnt vmcnt(0)).
// The buffer is 64 KB and was read once before (L2-resident).  Widths: dword, dwordx2, dwordx4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP4(x, k) x(k) x(k + 1) x(k + 2) x(k + 3)
#define REP16(x, k) REP4(x, k) REP4(x, k + 4) REP4(x, k + 8) REP4(x, k + 12)
#define REP32(x, k) REP16(x, k) REP16(x, k + 16)

template <int WIDTH, bool STORE>
__global__ void k(unsigned long long* out, float* buf, float* sink) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every wave has rows of its own; row pitch = 64 lanes * WIDTH dwords
    float* base = buf;   // (wave-uniform: an SGPR pair)
    const unsigned off = wave * 32 * 64 * WIDTH * 4 + lane * WIDTH * 4;
    float acc = 0.0f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    typedef float v4 __attribute__((ext_vector_type(4)));
    // warm the lines (L2 / L1)
    for (int r = 0; r < 32; ++r) acc += base[wave * 32 * 64 * WIDTH + r * 64 * WIDTH + lane * WIDTH];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float r1[32];
    v2 r2[32];
    v4 r4[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { r1[i] = acc + i; r2[i] = v2{acc, acc + i}; r4[i] = v4{acc, acc + i, acc, acc}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (!STORE) {
#define LD1(K) asm volatile("global_load_dword %0, %1, %2 offset:0" : "=v"(r1[K]) : "v"(off + (K) * 256u), "s"(base) : "memory");
#define LD2(K) asm volatile("global_load_dwordx2 %0, %1, %2 offset:0" : "=v"(r2[K]) : "v"(off + (K) * 512u), "s"(base) : "memory");
#define LD4(K) asm volatile("global_load_dwordx4 %0, %1, %2 offset:0" : "=v"(r4[K]) : "v"(off + (K) * 1024u), "s"(base) : "memory");
        if (WIDTH == 1) { REP32(LD1, 0) }
        if (WIDTH == 2) { REP32(LD2, 0) }
        if (WIDTH == 4) { REP32(LD4, 0) }
    } else {
#define ST1(K) asm volatile("global_store_dword %0, %1, %2 offset:0" ::"v"(off + (K) * 256u), "v"(r1[K]), "s"(base) : "memory");
#define ST2(K) asm volatile("global_store_dwordx2 %0, %1, %2 offset:0" ::"v"(off + (K) * 512u), "v"(r2[K]), "s"(base) : "memory");
#define ST4(K) asm volatile("global_store_dwordx4 %0, %1, %2 offset:0" ::"v"(off + (K) * 1024u), "v"(r4[K]), "s"(base) : "memory");
        if (WIDTH == 1) { REP32(ST1, 0) }
        if (WIDTH == 2) { REP32(ST2, 0) }
        if (WIDTH == 4) { REP32(ST4, 0) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += r1[i] + r2[i].x + r4[i].y;
    if (lane == 0) {
        out[wave * 2 + 0] = t1 - t0;
        out[wave * 2 + 1] = t2 - t0;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int WIDTH, bool STORE>
void run(const char* name, unsigned long long* d_out, float* d_buf, float* d_sink) {
    printf("%-28s", name);
    for (int threads : {64, 256, 512, 1024}) {
        std::vector<double> issue, landed;
        for (int rep = 0; rep < 20; ++rep) {
            hipLaunchKernelGGL((k<WIDTH, STORE>), dim3(1), dim3(threads), 0, 0, d_out, d_buf, d_sink);
            hipDeviceSynchronize();
            unsigned long long h[32];
            hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
            if (rep >= 4) { issue.push_back((double)h[0]); landed.push_back((double)h[1]); }
        }
        std::sort(issue.begin(), issue.end());
        std::sort(landed.begin(), landed.end());
        printf("  %4d thr: issue %6.1f/instr, all landed %6.0f", threads, issue[issue.size() / 2] / 32.0, landed[landed.size() / 2]);
    }
    printf("\n");
    fflush(stdout);
}

int main() {
    unsigned long long* d_out;
    float *d_buf, *d_sink;
    hipMalloc(&d_out, 64 * sizeof(unsigned long long));
    hipMalloc(&d_buf, 16 * 32 * 64 * 4 * sizeof(float));
    hipMemset(d_buf, 0, 16 * 32 * 64 * 4 * sizeof(float));
    hipMalloc(&d_sink, 64);
    printf("wave 0 of ONE workgroup: 32 row loads / stores (lane l -> row k, element l), cycles (s_memtime)\n");
    run<1, false>("global_load_dword", d_out, d_buf, d_sink);
    run<2, false>("global_load_dwordx2", d_out, d_buf, d_sink);
    run<4, false>("global_load_dwordx4", d_out, d_buf, d_sink);
    run<1, true>("global_store_dword", d_out, d_buf, d_sink);
    run<2, true>("global_store_dwordx2", d_out, d_buf, d_sink);
    run<4, true>("global_store_dwordx4", d_out, d_buf, d_sink);
    return 0;
}
