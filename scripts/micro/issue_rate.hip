// issue_rate.hip -- what one wave's instruction stream costs on gfx950, alone on its SIMD and beside 1 / 3 other waves.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/issue_rate.hip -o gpurun_out/issue_rate && gpurun_out/issue_rate
// Each test body is 64 instructions in one asm block, looped 32 times, bracketed by s_memtime; the table prints
// shader cycles per instruction for workgroups of 64 (one wave on one SIMD), 256 (one wave per SIMD), 512 and 1024
// threads (2 / 4 waves per SIMD) on ONE CU.  Decides: packed-f32 math in the ray stage / RK stage sums, DPP lane
// exchange cost, what a scalar-cache miss costs a lone wave (parameter block), LDS round trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

enum { T_FMA_IND = 0, T_FMA_DEP, T_PKFMA_IND, T_PKFMA_DEP, T_MOV_DPP, T_FMAC_DPP, T_SIN, T_RCP, T_MIX_SALU, T_DSREAD_DEP,
       T_FMA_SGPR2, T_PKMUL_IND, T_FMA_HALF, T_FMA_QUARTER, T_COUNT };
static const char* kNames[T_COUNT] = {"v_fma_f32 x8 independent", "v_fma_f32 dependent chain", "v_pk_fma_f32 x8 independent",
                                      "v_pk_fma_f32 dependent chain", "v_mov_b32_dpp quad_perm", "v_fmac_f32_dpp quad_perm",
                                      "v_sin_f32 independent", "v_rcp_f32 independent", "v_fma + s_add alternating",
                                      "ds_read_b32 dependent chain", "v_fma_f32 (1 sgpr) + v_mov pair", "v_pk_mul_f32 x8 independent",
                                      "v_fma_f32 dep, lanes 0-31 only", "v_fma_f32 dep, lanes 0-15 only"};

template <int TEST>
__global__ void k(unsigned long long* out, float* sink, float seed) {
    __shared__ float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 0.0f;   // chain of zeros: address 0 -> 0
    __syncthreads();
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    v2 pb = {b, b}, pc = {c, c};
    int s0 = 1;
    unsigned addr = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 32; ++it) {
        if (TEST == T_FMA_IND) {
            asm volatile(REP4(REP4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n")
                              REP4("v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (TEST == T_FMA_DEP) {
            asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
        } else if (TEST == T_PKFMA_IND) {
            asm volatile(REP4(REP4("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n")
                              REP4("v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"))
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
        } else if (TEST == T_PKFMA_DEP) {
            asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(p0) : "v"(pb), "v"(pc));
        } else if (TEST == T_PKMUL_IND) {
            asm volatile(REP4(REP4("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n")
                              REP4("v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"))
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));
        } else if (TEST == T_MOV_DPP) {
            asm volatile(REP16("v_mov_b32_dpp %0, %4 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %6 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if (TEST == T_FMAC_DPP) {
            asm volatile(REP16("v_fmac_f32_dpp %0, %4, %8 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %5, %8 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %2, %6, %8 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %7, %8 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(b));
        } else if (TEST == T_SIN) {
            asm volatile(REP16("v_sin_f32 %0, %4\n v_sin_f32 %1, %5\n v_sin_f32 %2, %6\n v_sin_f32 %3, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if (TEST == T_RCP) {
            asm volatile(REP16("v_rcp_f32 %0, %4\n v_rcp_f32 %1, %5\n v_rcp_f32 %2, %6\n v_rcp_f32 %3, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if (TEST == T_MIX_SALU) {
            asm volatile(REP16("v_fma_f32 %0, %0, %5, %6\n s_add_i32 %4, %4, 1\n v_fma_f32 %1, %1, %5, %6\n s_add_i32 %4, %4, 3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0) : "v"(b), "v"(c));
        } else if (TEST == T_DSREAD_DEP) {
            asm volatile(REP64("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(addr));
        } else if (TEST == T_FMA_HALF) {
            if (threadIdx.x % 64 < 32) asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
        } else if (TEST == T_FMA_QUARTER) {
            if (threadIdx.x % 64 < 16) asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
        } else if (TEST == T_FMA_SGPR2) {
            // the constant bus takes one SGPR per VALU instruction on gfx9: a second scalar operand costs a v_mov
            asm volatile(REP16("v_mov_b32 %2, %4\n v_fma_f32 %0, %0, %5, %2\n v_mov_b32 %3, %4\n v_fma_f32 %1, %1, %5, %3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s0), "s"(__builtin_amdgcn_readfirstlane(__float_as_int(b))));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.y + (float)s0 + (float)addr;
    if (r == 12345.678f) sink[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}

// scalar loads: a fresh 64-B line per load (miss) vs the same line again (hit), dependent through the address
__global__ void k_smem(unsigned long long* out, const uint32_t* tab, int stride_dw) {
    // tab[i] = 0 everywhere: the next address depends on the loaded value
    const uint32_t __attribute__((address_space(4)))* p = (const uint32_t __attribute__((address_space(4)))*)(uintptr_t)tab;
    uint32_t v = 0;
    unsigned long long t[5];
    t[0] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 16; ++i) { v = p[v + i * stride_dw]; asm volatile("" : "+s"(v)); }          // cold lines
    t[1] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 16; ++i) { v = p[v + i * stride_dw]; asm volatile("" : "+s"(v)); }          // the same lines again
    t[2] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 16; ++i) { v = p[v + (16 + i) * stride_dw + 1]; asm volatile("" : "+s"(v)); }   // cold again
    t[3] = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = (t[1] - t[0]) / 16; out[1] = (t[2] - t[1]) / 16; out[2] = (t[3] - t[2]) / 16; out[3] = v;
    }
}

template <int TEST>
static void run(unsigned long long* d_out, float* d_sink, double res[4]) {
    const int sizes[4] = {64, 256, 512, 1024};
    for (int s = 0; s < 4; ++s) {
        std::vector<double> v;
        for (int rep = 0; rep < 7; ++rep) {
            hipLaunchKernelGGL(k<TEST>, dim3(1), dim3(sizes[s]), 0, 0, d_out, d_sink, 0.25f);
            hipDeviceSynchronize();
            unsigned long long h[16];
            hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
            double mx = 0;
            for (int w = 0; w < sizes[s] / 64; ++w) mx = std::max(mx, (double)h[w]);
            const double n_inst = (TEST == T_FMA_IND || TEST == T_PKFMA_IND || TEST == T_PKMUL_IND) ? 128.0 : 64.0;
            v.push_back(mx / (32.0 * n_inst));
        }
        std::sort(v.begin(), v.end());
        res[s] = v[3];
    }
}

int main() {
    unsigned long long* d_out;
    float* d_sink;
    hipMalloc(&d_out, 16 * sizeof(unsigned long long));
    hipMalloc(&d_sink, 1024 * sizeof(float));
    printf("cycles (s_memtime ticks) per instruction of ONE wave's stream; columns: waves per SIMD on one CU\n");
    printf("%-36s %10s %10s %10s %10s\n", "test", "1 wave", "1/SIMD x4", "2/SIMD", "4/SIMD");
    fflush(stdout);
    double r[4];
#define RUN(T) run<T>(d_out, d_sink, r); printf("%-36s %10.2f %10.2f %10.2f %10.2f\n", kNames[T], r[0], r[1], r[2], r[3]); fflush(stdout);
    RUN(T_FMA_IND) RUN(T_FMA_DEP) RUN(T_PKFMA_IND) RUN(T_PKFMA_DEP) RUN(T_PKMUL_IND) RUN(T_MOV_DPP) RUN(T_FMAC_DPP) RUN(T_SIN) RUN(T_RCP)
    RUN(T_FMA_HALF) RUN(T_FMA_QUARTER) RUN(T_DSREAD_DEP) RUN(T_FMA_SGPR2)
    uint32_t* tab;
    hipMalloc(&tab, 1 << 20);
    hipMemset(tab, 0, 1 << 20);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_smem, dim3(1), dim3(64), 0, 0, d_out, tab, 64);   // 256-B stride
        hipDeviceSynchronize();
        unsigned long long h[4];
        hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
        printf("s_load_dword dependent chain: first touch %llu, same lines again %llu, other cold lines %llu cycles per load (launch %d)\n", h[0], h[1], h[2], rep);
    }
    return 0;
}
