// accuracy of hardware transcendentals on gfx950 vs double libm (host), for the ranges the step kernel uses
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* s, float* c, float* l, float* at, const float* y2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = x[i] * 0.15915494309189535f;       // revolutions
    s[i] = __builtin_amdgcn_sinf(r);
    c[i] = __builtin_amdgcn_cosf(r);
    l[i] = __builtin_amdgcn_logf(fabsf(x[i]) + 1e-3f) * 0.6931471805599453f;   // v_log_f32 = log2
    at[i] = atan2f(y2[i], x[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> hx(n), hy(n), hs(n), hc(n), hl(n), ha(n);
    for (int i = 0; i < n; ++i) { hx[i] = -4.5f + 9.0f * i / n; hy[i] = -3.0f + 6.0f * ((i * 7919) % n) / n; }
    float *dx, *dy, *ds, *dc, *dl, *da;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dl, n * 4); hipMalloc(&da, n * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dy, hy.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, ds, dc, dl, da, dy, n);
    hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hl.data(), dl, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ha.data(), da, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, el = 0, ea = 0;
    for (int i = 0; i < n; ++i) {
        es = fmax(es, fabs(hs[i] - sin((double)hx[i]))); ec = fmax(ec, fabs(hc[i] - cos((double)hx[i])));
        double lr = log(fabs((double)hx[i]) + 1e-3); el = fmax(el, fabs(hl[i] - lr) / fmax(1.0, fabs(lr)));
        ea = fmax(ea, fabs(ha[i] - atan2((double)hy[i], (double)hx[i])));
    }
    printf("v_sin max abs err %.3e | v_cos %.3e | v_log*ln2 max rel(abs for |l|<1) %.3e | ocml atan2f %.3e\n", es, ec, el, ea);
    return 0;
}
