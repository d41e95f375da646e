// dockauv_kernels.hip -- the fused docking3d step kernel for gfx950 (MI355X).
//
// One launch = BaseDocking3d.step (reference envs/docking3d.py:346-402) for every env of the handle:
//   current speed + body-frame current -> un-normalise + low-pass -> RKF45(4) step of the Fossen 6-DOF model
//   -> angle wrap -> ray fan vs capsules/spheres -> body collision -> navigation errors -> observation
//   -> done conditions -> 13-term reward -> (optional) in-kernel episode reset.
//
// Work decomposition (no MFMA: there is no dense contraction on this path):
//   * a workgroup owns EPG consecutive envs; lane e of the first EPG threads integrates env e ("env phase"):
//     every per-env array is struct-of-arrays in HBM, so lane-consecutive = address-consecutive (256 B / wave
//     instruction), vehicle / reward parameters arrive in the kernarg segment (scalar loads, SGPR operands);
//   * the ray stage runs over ALL threads of the group on (env, ray) items: item -> env = item % EPG (lane ~ env,
//     so a wave shares one body-frame ray: scalar table loads), ray = item / EPG.  The env phase hands the new
//     pose and the ray-independent obstacle terms over through LDS ([field][EPG] => conflict-free), the ray stage
//     writes clamped distances to LDS [ray][EPG];
//   * the env phase then reads its rays back (block-max image, obstacle-avoidance sums), builds the observation
//     in an LDS tile [EPG][n_obs], and the whole group streams the tile out row-major (what the learner wants)
//     with fully coalesced stores.
// Numerics: T = float is the product path; T = double instantiates the same code for validation.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdint.h>

#include "dockauv_device.h"

namespace dockauv {

// ------------------------------------------------------------------------------------------ math wrappers
__device__ __forceinline__ void sincos_(float x, float& s, float& c) { sincosf(x, &s, &c); }
__device__ __forceinline__ void sincos_(double x, double& s, double& c) { sincos(x, &s, &c); }
__device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
__device__ __forceinline__ float log_(float x) { return logf(x); }
__device__ __forceinline__ double log_(double x) { return log(x); }
__device__ __forceinline__ float atan2_(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double atan2_(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float floor_(float x) { return floorf(x); }
__device__ __forceinline__ double floor_(double x) { return floor(x); }
__device__ __forceinline__ float abs_(float x) { return fabsf(x); }
__device__ __forceinline__ double abs_(double x) { return fabs(x); }
__device__ __forceinline__ float hypot_(float x, float y) { return hypotf(x, y); }
__device__ __forceinline__ double hypot_(double x, double y) { return hypot(x, y); }
__device__ __forceinline__ bool isnan_(float x) { return x != x; }
__device__ __forceinline__ bool isnan_(double x) { return x != x; }

template <typename T> __device__ __forceinline__ T pi_() { return T(3.14159265358979323846); }
template <typename T> __device__ __forceinline__ T inf_() { return T(INFINITY); }

// np.clip(x, lo, hi): NaN propagates
template <typename T> __device__ __forceinline__ T clip_(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }
// max that propagates NaN like np.max
template <typename T> __device__ __forceinline__ T nanmax_(T a, T b) { return (isnan_(a) || a > b) ? a : b; }

// ssa: ((a + pi) mod 2 pi) - pi in [-pi, pi), Python floor-mod.  utils/geomutils.py:4-11
template <typename T>
__device__ __forceinline__ T ssa_(T a) {
    const T two_pi = T(2) * pi_<T>();
    T t = a + pi_<T>();
    T k = floor_(t / two_pi);
    T r = t - k * two_pi;
    if (r < T(0)) r += two_pi;
    if (r >= two_pi) r -= two_pi;
    return r - pi_<T>();
}

template <typename T>
struct Trig {
    T sf, cf, st, ct, sp, cp;
};

template <typename T>
__device__ __forceinline__ Trig<T> trig_(T phi, T th, T psi) {
    Trig<T> g;
    sincos_(phi, g.sf, g.cf);
    sincos_(th, g.st, g.ct);
    sincos_(psi, g.sp, g.cp);
    return g;
}

// R_b^n(Theta), utils/geomutils.py:14-43
template <typename T>
__device__ __forceinline__ void rotmat_(const Trig<T>& g, T R[9]) {
    R[0] = g.cp * g.ct;
    R[1] = -g.sp * g.cf + g.cp * g.st * g.sf;
    R[2] = g.sp * g.sf + g.cp * g.cf * g.st;
    R[3] = g.sp * g.ct;
    R[4] = g.cp * g.cf + g.sf * g.st * g.sp;
    R[5] = -g.cp * g.sf + g.st * g.sp * g.cf;
    R[6] = -g.st;
    R[7] = g.ct * g.sf;
    R[8] = g.ct * g.cf;
}

// ------------------------------------------------------------------------------------------ Fossen RHS
// nu_dot = M^-1 (B(nu) u - D(nu) nu - C(nu) nu - g(eta)),  objects/auvsim.py:152-158, statespace.py:199-397
template <typename T, int VK>
__device__ __forceinline__ void kinetics_(const VehicleP<T>& V, const T nu[6], const Trig<T>& g, const T tau_c[6],
                                          const T* u, T nud[6]) {
    const T u_ = nu[0], v_ = nu[1], w_ = nu[2], p_ = nu[3], q_ = nu[4], r_ = nu[5];
    // --- Coriolis as cross products.  C_RB = [[m S(v2), -m S(v2) S(rG)], [m S(rG) S(v2), -S(I_b v2)]]
    const T c1x = q_ * w_ - r_ * v_, c1y = r_ * u_ - p_ * w_, c1z = p_ * v_ - q_ * u_;              // v2 x v1
    const T t1x = V.rg[1] * r_ - V.rg[2] * q_, t1y = V.rg[2] * p_ - V.rg[0] * r_, t1z = V.rg[0] * q_ - V.rg[1] * p_;
    const T c2x = q_ * t1z - r_ * t1y, c2y = r_ * t1x - p_ * t1z, c2z = p_ * t1y - q_ * t1x;         // v2 x (rG x v2)
    const T c3x = V.rg[1] * c1z - V.rg[2] * c1y, c3y = V.rg[2] * c1x - V.rg[0] * c1z, c3z = V.rg[0] * c1y - V.rg[1] * c1x;
    const T ivx = V.Ib[0] * p_ + V.Ib[1] * q_ + V.Ib[2] * r_;
    const T ivy = V.Ib[3] * p_ + V.Ib[4] * q_ + V.Ib[5] * r_;
    const T ivz = V.Ib[6] * p_ + V.Ib[7] * q_ + V.Ib[8] * r_;
    const T c4x = ivy * r_ - ivz * q_, c4y = ivz * p_ - ivx * r_, c4z = ivx * q_ - ivy * p_;         // (I_b v2) x v2
    // C_A = [[0, -S(a1)], [-S(a1), -S(a2)]],  a = M_A nu (diagonal M_A)
    const T a1x = V.ma[0] * u_, a1y = V.ma[1] * v_, a1z = V.ma[2] * w_;
    const T a2x = V.ma[3] * p_, a2y = V.ma[4] * q_, a2z = V.ma[5] * r_;
    T C[6];
    C[0] = V.m * (c1x - c2x) - (a1y * r_ - a1z * q_);
    C[1] = V.m * (c1y - c2y) - (a1z * p_ - a1x * r_);
    C[2] = V.m * (c1z - c2z) - (a1x * q_ - a1y * p_);
    C[3] = V.m * c3x - c4x - (a1y * w_ - a1z * v_) - (a2y * r_ - a2z * q_);
    C[4] = V.m * c3y - c4y - (a1z * u_ - a1x * w_) - (a2z * p_ - a2x * r_);
    C[5] = V.m * c3z - c4z - (a1x * v_ - a1y * u_) - (a2x * q_ - a2y * p_);
    // --- damping force D(nu) nu
    T Dn[6];
    const T au = abs_(u_), av = abs_(v_), aw = abs_(w_), ap = abs_(p_), aq = abs_(q_), ar = abs_(r_);
    if (VK == VK_LAUV) {
        const T* L = V.lauv;
        Dn[0] = -(V.dl[0] + V.dq[0] * au) * u_;
        Dn[1] = -(V.dl[1] + V.dq[1] * av + L[L_Y_uv] * au) * v_ - (L[L_Y_r] + L[L_Y_rr] * ar + L[L_Y_urf] * au) * r_;
        Dn[2] = -(V.dl[2] + V.dq[2] * aw + L[L_Z_uw] * au) * w_ - (L[L_Z_q] + L[L_Z_qq] * aq + L[L_Z_uqf] * au) * q_;
        Dn[3] = -(V.dl[3] + V.dq[3] * ap) * p_;
        Dn[4] = -(V.dl[4] + V.dq[4] * aq + L[L_M_uqf] * au) * q_ - (L[L_M_w] + L[L_M_ww] * aw + L[L_M_uw] * au) * w_;
        Dn[5] = -(V.dl[5] + V.dq[5] * ar + L[L_N_urf] * au) * r_ - (L[L_N_v] + L[L_N_vv] * av + L[L_N_uv] * au) * v_;
    } else {
        Dn[0] = -(V.dl[0] + V.dq[0] * au) * u_;
        Dn[1] = -(V.dl[1] + V.dq[1] * av) * v_;
        Dn[2] = -(V.dl[2] + V.dq[2] * aw) * w_;
        Dn[3] = -(V.dl[3] + V.dq[3] * ap) * p_;
        Dn[4] = -(V.dl[4] + V.dq[4] * aq) * q_;
        Dn[5] = -(V.dl[5] + V.dq[5] * ar) * r_;
    }
    // --- restoring forces g(eta)
    T G[6];
    G[0] = V.gWB * g.st;
    G[1] = -V.gWB * g.ct * g.sf;
    G[2] = -V.gWB * g.ct * g.cf;
    G[3] = -V.gy * g.ct * g.cf + V.gz * g.ct * g.sf;
    G[4] = V.gz * g.st + V.gx * g.ct * g.cf;
    G[5] = -V.gx * g.ct * g.sf - V.gy * g.st;
    // --- control forces
    T tau[6];
    if (VK == VK_LAUV) {
        const T uu = u_ * u_;   // LAUV.py:60: fins scale with u**2
        tau[0] = u[0];
        tau[1] = V.lauv[L_Y_uudr] * uu * u[1];
        tau[2] = V.lauv[L_Z_uuds] * uu * u[2];
        tau[3] = T(0);
        tau[4] = V.lauv[L_M_uuds] * uu * u[2];
        tau[5] = V.lauv[L_N_uudr] * uu * u[1];
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) tau[i] = tau_c[i];
    }
    T rhs[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) rhs[i] = tau[i] - Dn[i] - C[i] - G[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < 6; ++j) acc += V.Minv[i * 6 + j] * rhs[j];
        nud[i] = acc;
    }
}

// one RHS evaluation: derivative of (pos, att, nu) at stage values (att, nu).  The RHS does not depend on position.
template <typename T, int VK, bool WANT_POS>
__device__ __forceinline__ void rhs_(const VehicleP<T>& V, const T att[3], const T nu[6], const T nuc[3],
                                     const T tau_c[6], const T* u, T pd[3], T ad[3], T nd[6]) {
    const Trig<T> g = trig_(att[0], att[1], att[2]);
    if (WANT_POS) {
        T R[9];
        rotmat_(g, R);
        const T vx = nu[0] + nuc[0], vy = nu[1] + nuc[1], vz = nu[2] + nuc[2];
        pd[0] = R[0] * vx + R[1] * vy + R[2] * vz;
        pd[1] = R[3] * vx + R[4] * vy + R[5] * vz;
        pd[2] = R[6] * vx + R[7] * vy + R[8] * vz;
    }
    // T_Theta, utils/geomutils.py:46-75
    const T tt = g.st / g.ct;
    const T sq_cr = g.sf * nu[4] + g.cf * nu[5];
    ad[0] = nu[3] + sq_cr * tt;
    ad[1] = g.cf * nu[4] - g.sf * nu[5];
    ad[2] = sq_cr / g.ct;
    kinetics_<T, VK>(V, nu, g, tau_c, u, nd);
}

// AUVSim.step: un-normalise, low-pass, Fehlberg 4th-order step, wrap.  objects/auvsim.py:77-108,
// utils/odesolver45.py:18-27 (stage 6 and the 5th-order result are dead in the reference and not computed).
template <typename T, int VK>
__device__ __forceinline__ void vehicle_step_(const VehicleP<T>& V, const EnvP<T>& E, T st[12], T u[kMaxU],
                                              const T act[kMaxU], const T nuc[3]) {
    const int n_u = (VK == VK_LAUV) ? 3 : (VK == VK_JOY ? 6 : V.n_u);
#pragma unroll
    for (int i = 0; i < kMaxU; ++i) {
        if (i < n_u) {
            const T a = clip_(act[i], T(-1), T(1));
            const T x = V.ulo[i] + V.uhalf[i] * (a + T(1)) / T(2);
            u[i] = E.lp_alpha * x + (T(1) - E.lp_alpha) * u[i];
        }
    }
    T tau_c[6] = {0, 0, 0, 0, 0, 0};
    if (VK == VK_JOY) {
#pragma unroll
        for (int i = 0; i < 6; ++i) tau_c[i] = V.B[i * kMaxU + i] * u[i];
    } else if (VK == VK_DENSEB) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < kMaxU; ++j)
                if (j < n_u) acc += V.B[i * kMaxU + j] * u[j];
            tau_c[i] = acc;
        }
    }
    const T h = E.h;
    T k1[12], k2[9], k3[12], k4[12], k5[12];  // k2 carries no position derivative (never used)
    T a_[3], n_[6];
    rhs_<T, VK, true>(V, st + 3, st + 6, nuc, tau_c, u, k1, k1 + 3, k1 + 6);
#pragma unroll
    for (int i = 0; i < 3; ++i) a_[i] = st[3 + i] + h * k1[3 + i] / T(4);
#pragma unroll
    for (int i = 0; i < 6; ++i) n_[i] = st[6 + i] + h * k1[6 + i] / T(4);
    rhs_<T, VK, false>(V, a_, n_, nuc, tau_c, u, nullptr, k2, k2 + 3);
#pragma unroll
    for (int i = 0; i < 3; ++i) a_[i] = st[3 + i] + T(3) * h * k1[3 + i] / T(32) + T(9) * h * k2[i] / T(32);
#pragma unroll
    for (int i = 0; i < 6; ++i) n_[i] = st[6 + i] + T(3) * h * k1[6 + i] / T(32) + T(9) * h * k2[3 + i] / T(32);
    rhs_<T, VK, true>(V, a_, n_, nuc, tau_c, u, k3, k3 + 3, k3 + 6);
#pragma unroll
    for (int i = 0; i < 3; ++i)
        a_[i] = st[3 + i] + T(1932) * h * k1[3 + i] / T(2197) - T(7200) * h * k2[i] / T(2197) +
                T(7296) * h * k3[3 + i] / T(2197);
#pragma unroll
    for (int i = 0; i < 6; ++i)
        n_[i] = st[6 + i] + T(1932) * h * k1[6 + i] / T(2197) - T(7200) * h * k2[3 + i] / T(2197) +
                T(7296) * h * k3[6 + i] / T(2197);
    rhs_<T, VK, true>(V, a_, n_, nuc, tau_c, u, k4, k4 + 3, k4 + 6);
#pragma unroll
    for (int i = 0; i < 3; ++i)
        a_[i] = st[3 + i] + T(439) * h * k1[3 + i] / T(216) - T(8) * h * k2[i] + T(3680) * h * k3[3 + i] / T(513) -
                T(845) * h * k4[3 + i] / T(4104);
#pragma unroll
    for (int i = 0; i < 6; ++i)
        n_[i] = st[6 + i] + T(439) * h * k1[6 + i] / T(216) - T(8) * h * k2[3 + i] + T(3680) * h * k3[6 + i] / T(513) -
                T(845) * h * k4[6 + i] / T(4104);
    rhs_<T, VK, true>(V, a_, n_, nuc, tau_c, u, k5, k5 + 3, k5 + 6);
#pragma unroll
    for (int i = 0; i < 12; ++i)
        st[i] = st[i] + h * (T(25) * k1[i] / T(216) + T(1408) * k3[i] / T(2565) + T(2197) * k4[i] / T(4104) - k5[i] / T(5));
    st[3] = ssa_(st[3]);
    st[4] = ssa_(st[4]);
    st[5] = ssa_(st[5]);
}

// ------------------------------------------------------------------------------------------ LDS layout
// (field counts and lds_bytes live in dockauv_device.h, shared with the host side)

// ------------------------------------------------------------------------------------------ ray kernels
// One ray vs one capsule: same case structure and same results as the vectorised reference routine
// (objects/shape.py:327-390), but evaluated in a form that is well conditioned in float32.  The reference works with
// the un-normalised axis ba and the products baba*oaoa - baoa^2, b^2 - a*c (40 m pillars: terms ~1e6, catastrophic
// cancellation in fp32).  With the unit axis d = ba/|ba| and components perpendicular to it,
//   a = |rd_perp|^2, b = rd . oa_perp, c = |oa_perp|^2 - r^2   (reference a, b, c divided by baba),
// t = (-b - sqrt(b^2 - a c)) / a and y/|ba| = oa_par + t (rd . d) are unchanged.  Cap spheres use
// h2 = r^2 - |oc - (rd . oc) rd|^2 (= b2^2 - c2 for unit rd).
// LDS record: d(3) oa_perp(3) oa_par len c r2
template <typename T>
__device__ __forceinline__ T ray_capsule_(const T* __restrict__ cp, int stride, T rx, T ry, T rz) {
    const T dx = cp[0 * stride], dy = cp[1 * stride], dz = cp[2 * stride];
    const T px = cp[3 * stride], py = cp[4 * stride], pz = cp[5 * stride];
    const T oa_par = cp[6 * stride], len = cp[7 * stride], cc = cp[8 * stride], r2 = cp[9 * stride];
    const T bard = rx * dx + ry * dy + rz * dz;
    const T qx = rx - bard * dx, qy = ry - bard * dy, qz = rz - bard * dz;   // rd_perp
    const T a = qx * qx + qy * qy + qz * qz;
    const T b = rx * px + ry * py + rz * pz;
    const T hh = b * b - a * cc;
    T res = -inf_<T>();
    if (hh > T(0)) {   // h <= 0 (tangent included) and NaN -> -inf  (shape.py:389)
        const T t = (-b - sqrt_(hh)) / a;
        const T y = oa_par + t * bard;
        if (y > T(0) && y < len) {
            res = t;
        } else {
            T ocx, ocy, ocz, rr = r2;
            if (y >= T(0)) {         // top cap wins at y == 0 (second assignment, shape.py:378-379)
                const T s = oa_par - len;
                ocx = px + s * dx; ocy = py + s * dy; ocz = pz + s * dz;
            } else if (y <= T(0)) {
                ocx = px + oa_par * dx; ocy = py + oa_par * dy; ocz = pz + oa_par * dz;
            } else {                 // y is NaN: oc stays the zero vector (shape.py:377)
                ocx = ocy = ocz = T(0);
            }
            const T b2 = rx * ocx + ry * ocy + rz * ocz;
            const T ex = ocx - b2 * rx, ey = ocy - b2 * ry, ez = ocz - b2 * rz;
            const T h2 = rr - (ex * ex + ey * ey + ez * ez);
            res = (h2 > T(0)) ? (-b2 - sqrt_(h2)) : T(0);
        }
        if (res == T(0)) res = -inf_<T>();
    }
    return res;
}

// ------------------------------------------------------------------------------------------ device-side reset
// Philox4x32-10 (Salmon et al., SC'11): counter-based, integer only -> bit-exact against the NumPy restatement in
// oracle/philox_ref.py.  Counter = (env, episode, block, 0), key = config.seed.
__device__ __forceinline__ void philox4x32_10_(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// Scenario generators of the reference (envs/docking3d.py:687-703, 795-988) fed with 12 uniforms U[k] = (x >> 8) 2^-24
// in the reference's draw order (same mapping as gym_dockauv_amd/scenarios.py: episodes_from_uniforms).
// Writes pose (6), goal (4), current rows (8) and the capsule slots of env `env`.
template <typename T>
__device__ __forceinline__ void generate_episode_(const EnvP<T>& E, const Buffers& B, int env, int episode, T pose[6],
                                                  T goal[4], T cur[8]) {
    T U[12];
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
        uint32_t c[4] = {(uint32_t)env, (uint32_t)episode, (uint32_t)blk, 0u};
        philox4x32_10_(c, (uint32_t)(E.seed & 0xffffffffull), (uint32_t)(E.seed >> 32));
#pragma unroll
        for (int j = 0; j < 4; ++j) U[blk * 4 + j] = T(c[j] >> 8) * T(1.0 / 16777216.0);
    }
    const T pi = pi_<T>();
    const int scn = E.scenario;
    goal[0] = goal[1] = goal[2] = T(0);
    goal[3] = (U[0] - T(0.5)) * pi;                                        // docking3d.py:814
    T rx = U[1] - T(0.5), ry = U[2] - T(0.5), rz = U[3] - T(0.5);          // :694
    const T sgn = rz > T(0) ? T(1) : (rz < T(0) ? T(-1) : T(0));
    rz = abs_(rx + ry) / T(3) * sgn;                                       // :695
    const T sc = T(15) / sqrt_(rx * rx + ry * ry + rz * rz);               // :696, :809
    pose[0] = rx * sc; pose[1] = ry * sc; pose[2] = rz * sc;
    pose[3] = (U[4] - T(0.5)) * T(2) * (E.max_att * T(0.7));               // :699-703
    pose[4] = (U[5] - T(0.5)) * T(2) * (E.max_att * T(0.7));
    pose[5] = (U[6] - T(0.5)) * T(2) * pi;
    T Vc = T(0), vmin = T(0), vmax = T(0), al = T(0), be = T(0);           // :820-822
    int k = 7;
    if (scn == 1) {                                                        // SimpleCurrent :844-848
        al = (U[7] - T(0.5)) * T(2) * (pi / T(2));
        be = (U[8] - T(0.5)) * T(2) * pi;
        vmin = vmax = U[9];
        Vc = T(0.5);
        k = 10;
    }
    T* g_caps = static_cast<T*>(B.caps);
    const long S = B.stride;
    if (scn >= 2 && scn <= 6) {                                            // Capsule* / Obstacles* :860-886
        T sth, cth;
        sincos_(U[k] * T(2) * pi, sth, cth);
        goal[0] = cth * T(2);                                              // CAPSULE_RADIUS + safety radius
        goal[1] = sth * T(2);
        goal[2] = (U[k + 1] - T(0.5)) * T(4);
        goal[3] = ssa_(atan2_(-goal[1], -goal[0]));                        // :884-886
        k += 2;
        int slot = 0;
        if (scn != 5 && slot < E.max_cap) {                                // centre capsule (popped by NoCap, :964)
            T* cg = g_caps + (size_t)slot * 7 * S + env;
            cg[0 * S] = T(0); cg[1 * S] = T(0); cg[2 * S] = T(2);
            cg[3 * S] = T(0); cg[4 * S] = T(0); cg[5 * S] = T(-2);
            cg[6 * S] = T(1);
            ++slot;
        }
        if (scn >= 4) {                                                    // four pillars :923-946
            T th = U[k] * T(2) * pi;
            k += 1;
            const T half = E.dmax;                                         // 2 * max_dist_from_goal / 2
            for (int i = 0; i < 4; ++i) {
                T s_, c_;
                sincos_(th, s_, c_);
                th += pi / T(2);
                if (slot < E.max_cap) {
                    T* cg = g_caps + (size_t)slot * 7 * S + env;
                    cg[0 * S] = c_ * T(6); cg[1 * S] = s_ * T(6); cg[2 * S] = half;
                    cg[3 * S] = c_ * T(6); cg[4 * S] = s_ * T(6); cg[5 * S] = -half;
                    cg[6 * S] = T(1);
                    ++slot;
                }
            }
        }
        for (; slot < E.max_cap; ++slot) g_caps[((size_t)slot * 7 + 6) * S + env] = T(-1);
        if (scn == 3 || scn == 6) {                                        // *Current :904-906, :984-986
            al = (U[k] - T(0.5)) * T(2) * (pi / T(2));
            be = (U[k + 1] - T(0.5)) * T(2) * pi;
            Vc = vmin = vmax = T(0.5);
        }
    }
    T sa, ca, sb, cb;
    sincos_(al, sa, ca);
    sincos_(be, sb, cb);
    cur[0] = Vc; cur[1] = ca * cb; cur[2] = sb; cur[3] = sa * cb;         // objects/current.py:70-74
    cur[4] = vmin; cur[5] = vmax; cur[6] = al; cur[7] = be;
}

// ------------------------------------------------------------------------------------------ the step kernel
template <typename T, int VK, bool RAYS, int EPG, int NT>
__global__ __launch_bounds__(NT) void step_kernel(const KernelArgs<T, 2> A) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const EnvP<T>& E = A.E;
    const Buffers& B = A.B;
    const long S = B.stride;
    const int tid = threadIdx.x;
    const int env0 = blockIdx.x * EPG;
    const int env = env0 + tid;
    const bool owner = (tid < EPG) && (env < E.n_envs);
    const int n_valid = min(EPG, E.n_envs - env0);

    // LDS carve-up
    T* lds_pose = reinterpret_cast<T*>(smem_raw);
    T* lds_cap = lds_pose + (RAYS ? kPoseFields * EPG : 0);
    T* lds_sph = lds_cap + (RAYS ? kCapFields * E.max_cap * EPG : 0);
    T* lds_dist = lds_sph + (RAYS ? kSphFields * E.max_sph * EPG : 0);
    size_t t_bytes = RAYS ? (size_t)(lds_dist + (size_t)E.n_rays * EPG - lds_pose) * sizeof(T) : 0;
    t_bytes = (t_bytes + 15) & ~(size_t)15;
    float* lds_obs = reinterpret_cast<float*>(smem_raw + t_bytes);

    T* g_state = static_cast<T*>(B.state);
    T* g_u = static_cast<T*>(B.u);
    T* g_goal = static_cast<T*>(B.goal);
    T* g_cur = static_cast<T*>(B.cur);
    T* g_cum = static_cast<T*>(B.cum_reward);
    T* g_caps = static_cast<T*>(B.caps);
    T* g_sph = static_cast<T*>(B.sph);

    // registers that live across the ray stage (env phase lanes only)
    T st[12], u[kMaxU], act[kMaxU], nuc[3], goal[4];
    T Vc = T(0), edot_norm2 = T(0);
    Trig<T> g1{};
    bool collision = false;
    int n_u_env = E.n_u_max;

    if (owner) {
        // ---------------- load (coalesced SoA) ----------------
#pragma unroll
        for (int k = 0; k < 12; ++k) st[k] = g_state[k * S + env];
        const T* actp = static_cast<const T*>(A.io.actions) + (size_t)env * E.n_u_max;
#pragma unroll
        for (int k = 0; k < kMaxU; ++k) {
            if (k < E.n_u_max) {
                u[k] = g_u[k * S + env];
                act[k] = actp[k];
            } else {
                u[k] = T(0);
                act[k] = T(0);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) goal[k] = g_goal[k * S + env];
        Vc = g_cur[0 * S + env];
        const T cdx = g_cur[1 * S + env], cdy = g_cur[2 * S + env], cdz = g_cur[3 * S + env];
        const T vmin = g_cur[4 * S + env], vmax = g_cur[5 * S + env];

        // ---------------- 1. Current.sim (objects/current.py:78-96) ----------------
        const T w = A.io.noise ? static_cast<const T*>(A.io.noise)[env] : T(0);
        Vc += (-E.mu * Vc + w) * E.h;
        Vc = clip_(Vc, vmin, vmax);
        // ---------------- 2. nu_c = R(Theta_pre)^T v_c^n (objects/current.py:33-53) ----------------
        {
            const Trig<T> g0 = trig_(st[3], st[4], st[5]);
            T R0[9];
            rotmat_(g0, R0);
            const T vx = Vc * cdx, vy = Vc * cdy, vz = Vc * cdz;
            nuc[0] = R0[0] * vx + R0[3] * vy + R0[6] * vz;
            nuc[1] = R0[1] * vx + R0[4] * vy + R0[7] * vz;
            nuc[2] = R0[2] * vx + R0[5] * vy + R0[8] * vz;
        }
        // ---------------- 3. vehicle ----------------
        if (VK == VK_MIXED) {
            if (B.veh_id[env] == 0) {
                vehicle_step_<T, VK_JOY>(A.V[0], E, st, u, act, nuc);
                n_u_env = 6;
            } else {
                vehicle_step_<T, VK_LAUV>(A.V[1], E, st, u, act, nuc);
                n_u_env = 3;
            }
        } else {
            vehicle_step_<T, VK>(A.V[0], E, st, u, act, nuc);
            n_u_env = (VK == VK_LAUV) ? 3 : (VK == VK_JOY ? 6 : A.V[0].n_u);
        }
        // post-step trig of the wrapped attitude: euler_dot (Q4, auvsim.py:108), ray rotation, obs[8:10]
        g1 = trig_(st[3], st[4], st[5]);
        {
            const T tt = g1.st / g1.ct;
            const T sq_cr = g1.sf * st[10] + g1.cf * st[11];
            const T e0 = st[9] + sq_cr * tt, e1 = g1.cf * st[10] - g1.sf * st[11], e2 = sq_cr / g1.ct;
            edot_norm2 = e0 * e0 + e1 * e1 + e2 * e2;
        }
        if (RAYS) {
            // ---------------- hand the pose + obstacle terms to the ray stage; body collision ----------------
            T R1[9];
            rotmat_(g1, R1);
            lds_pose[0 * EPG + tid] = st[0];
            lds_pose[1 * EPG + tid] = st[1];
            lds_pose[2 * EPG + tid] = st[2];
#pragma unroll
            for (int k = 0; k < 9; ++k) lds_pose[(3 + k) * EPG + tid] = R1[k];
            int ncap = 0;
            bool open = true;
            for (int c = 0; c < E.max_cap; ++c) {
                const T* cg = g_caps + (size_t)c * 7 * S + env;
                const T c1x = cg[0 * S], c1y = cg[1 * S], c1z = cg[2 * S];
                const T c2x = cg[3 * S], c2y = cg[4 * S], c2z = cg[5 * S];
                const T rad = cg[6 * S];
                open = open && (rad > T(0));
                if (open) {
                    ncap = c + 1;
                    const T bax = c2x - c1x, bay = c2y - c1y, baz = c2z - c1z;
                    const T oax = st[0] - c1x, oay = st[1] - c1y, oaz = st[2] - c1z;
                    const T ocx = st[0] - c2x, ocy = st[1] - c2y, ocz = st[2] - c2z;
                    const T baba = bax * bax + bay * bay + baz * baz;
                    const T len = sqrt_(baba);
                    const T inv_len = T(1) / len;
                    const T dx = bax * inv_len, dy = bay * inv_len, dz = baz * inv_len;
                    const T oa_par = oax * dx + oay * dy + oaz * dz;
                    const T px = oax - oa_par * dx, py = oay - oa_par * dy, pz = oaz - oa_par * dz;
                    T* cp = lds_cap + (size_t)c * kCapFields * EPG + tid;
                    cp[0 * EPG] = dx; cp[1 * EPG] = dy; cp[2 * EPG] = dz;
                    cp[3 * EPG] = px; cp[4 * EPG] = py; cp[5 * EPG] = pz;
                    cp[6 * EPG] = oa_par;
                    cp[7 * EPG] = len;
                    cp[8 * EPG] = px * px + py * py + pz * pz - rad * rad;
                    cp[9 * EPG] = rad * rad;
                    // collision_capsule_sphere / dist_line_point (objects/shape.py:195-210, 393-417)
                    const T s = -oa_par;
                    const T t = ocx * dx + ocy * dy + ocz * dz;
                    T hh = s > t ? s : t;
                    hh = hh > T(0) ? hh : T(0);
                    const T cxx = oay * dz - oaz * dy, cxy = oaz * dx - oax * dz, cxz = oax * dy - oay * dx;
                    const T dist = hypot_(hh, sqrt_(cxx * cxx + cxy * cxy + cxz * cxz));
                    collision = collision || (dist <= rad + E.safety);
                }
            }
            int nsph = 0;
            open = true;
            for (int s = 0; s < E.max_sph; ++s) {
                const T* sg = g_sph + (size_t)s * 4 * S + env;
                const T cx = sg[0 * S], cy = sg[1 * S], cz = sg[2 * S], rad = sg[3 * S];
                open = open && (rad > T(0));
                if (open) {
                    nsph = s + 1;
                    const T ocx = st[0] - cx, ocy = st[1] - cy, ocz = st[2] - cz;
                    const T d2 = ocx * ocx + ocy * ocy + ocz * ocz;
                    T* sp = lds_sph + (size_t)s * kSphFields * EPG + tid;
                    sp[0 * EPG] = ocx; sp[1 * EPG] = ocy; sp[2 * EPG] = ocz;
                    sp[3 * EPG] = rad * rad;
                    // collision_sphere_spheres (objects/shape.py:182-192)
                    collision = collision || (sqrt_(d2) <= E.safety + rad);
                }
            }
            lds_pose[12 * EPG + tid] = T(ncap);
            lds_pose[13 * EPG + tid] = T(nsph);
        }
    }

    if (RAYS) {
        __syncthreads();
        // ---------------- 4-6. ray stage over (env, ray) items ----------------
        const T* rays = static_cast<const T*>(B.rays);
        const int items = EPG * E.n_rays;
        for (int item = tid; item < items; item += NT) {
            const int e = item % EPG;
            const int r = item / EPG;
            if (env0 + e >= E.n_envs) continue;
            // body-frame ray: same address for the whole wave (EPG is a multiple of 64) -> scalar loads
            const T bx = rays[r * 4 + 0], by = rays[r * 4 + 1], bz = rays[r * 4 + 2];
            const T* P = lds_pose + e;
            T rx = P[3 * EPG] * bx + P[4 * EPG] * by + P[5 * EPG] * bz;   // Radar.update, objects/sensor.py:97-102
            T ry = P[6 * EPG] * bx + P[7 * EPG] * by + P[8 * EPG] * bz;
            T rz = P[9 * EPG] * bx + P[10 * EPG] * by + P[11 * EPG] * bz;
            const T inv_n = T(1) / sqrt_(rx * rx + ry * ry + rz * rz);
            rx *= inv_n; ry *= inv_n; rz *= inv_n;
            const int ncap = (int)P[12 * EPG], nsph = (int)P[13 * EPG];
            // update_radar_collision, envs/docking3d.py:415-442: smallest positive over groups, else group 0
            T best = inf_<T>(), first = T(0);
            bool have_first = false;
            for (int c = 0; c < ncap; ++c) {
                const T v = ray_capsule_<T>(lds_cap + (size_t)c * kCapFields * EPG + e, EPG, rx, ry, rz);
                if (!have_first) { first = v; have_first = true; }
                if (v > T(0) && v < best) best = v;
            }
            if (nsph > 0) {
                // intersec_dist_lines_spheres_vectorized, objects/shape.py:235-264
                T sbest = inf_<T>(), sfirst = T(0);
                for (int s = 0; s < nsph; ++s) {
                    const T* sp = lds_sph + (size_t)s * kSphFields * EPG + e;
                    // h = b^2 - (|oc|^2 - r^2) evaluated as r^2 - |oc - b rd|^2 (no cancellation far from the sphere)
                    const T ox = sp[0 * EPG], oy = sp[1 * EPG], oz = sp[2 * EPG];
                    const T b = ox * rx + oy * ry + oz * rz;
                    const T ex = ox - b * rx, ey = oy - b * ry, ez = oz - b * rz;
                    const T hh = sp[3 * EPG] - (ex * ex + ey * ey + ez * ez);
                    const T v = (hh < T(0)) ? -inf_<T>() : (-b - sqrt_(hh));
                    if (s == 0) sfirst = v;
                    if (v > T(0) && v < sbest) sbest = v;
                }
                const T v = (sbest < inf_<T>()) ? sbest : sfirst;
                if (!have_first) { first = v; have_first = true; }
                if (v > T(0) && v < best) best = v;
            }
            T d;
            if (!have_first) {
                d = E.ray_max;                       // no obstacle at all: fallback (sensor.py:110-111)
            } else {
                d = (best < inf_<T>()) ? best : first;
                if (d < T(0) || d > E.ray_max) d = E.ray_max;   // sensor.py:117 (NaN survives)
            }
            lds_dist[(size_t)r * EPG + e] = d;
        }
        __syncthreads();
    }

    uint8_t done_flag = 0;
    if (owner) {
        // ---------------- 9. navigation errors (envs/docking3d.py:404-413) ----------------
        const T dx = goal[0] - st[0], dy = goal[1] - st[1], dz = goal[2] - st[2];
        const T dd = sqrt_(dx * dx + dy * dy + dz * dz);
        const T dxy = sqrt_(dx * dx + dy * dy);
        const T dth = st[4] + ssa_(atan2_(dz, dxy));
        const T dpsi = ssa_(atan2_(dy, dx) - st[5]);
        const T dhg = ssa_(goal[3] - st[5]);

        // ---------------- 10. observation (envs/docking3d.py:462-488) ----------------
        float* ot = lds_obs + (size_t)tid * E.n_obs;
        const T log_ratio = log_(E.dtol / E.dmax);
        T o[16];
        o[0] = clip_(T(1) - log_(dd / E.dmax) / log_ratio, T(0), T(1));
        o[1] = clip_(dth / (pi_<T>() / T(2)), T(-1), T(1));
        o[2] = clip_(dpsi / pi_<T>(), T(-1), T(1));
        o[3] = clip_(st[6] / E.vel_max[0], T(-1), T(1));
        o[4] = clip_(st[7] / E.vel_max[1], T(-1), T(1));
        o[5] = clip_(st[8] / E.vel_max[2], T(-1), T(1));
        o[6] = clip_(st[3] / E.max_att, T(-1), T(1));
        o[7] = clip_(st[4] / E.max_att, T(-1), T(1));
        o[8] = clip_(g1.sp, T(-1), T(1));
        o[9] = clip_(g1.cp, T(-1), T(1));
        o[10] = clip_(st[9] / E.vel_max[3], T(-1), T(1));
        o[11] = clip_(st[10] / E.vel_max[4], T(-1), T(1));
        o[12] = clip_(st[11] / E.vel_max[5], T(-1), T(1));
        o[13] = clip_(nuc[0] / T(2), T(-1), T(1));
        o[14] = clip_(nuc[1] / T(2), T(-1), T(1));
        o[15] = clip_(nuc[2] / T(2), T(-1), T(1));
#pragma unroll
        for (int k = 0; k < 16; ++k) ot[k] = (float)o[k];

        // rays: block max image (sensor.py:131-137) + obstacle avoidance sums (docking3d.py:766-792)
        T oa = T(0);
        if (RAYS) {
            const T* rays = static_cast<const T*>(B.rays);
            T sum_beta = T(0), denom = T(0);
            for (int cv = 0; cv < E.n_vr; ++cv) {
                for (int ch = 0; ch < E.n_hr; ++ch) {
                    T cell = T(0);   // zero padding (skimage block_reduce cval = 0)
                    for (int dv = 0; dv < E.blk; ++dv) {
                        const int iv = cv * E.blk + dv;
                        if (iv >= E.n_v) continue;
                        for (int dh = 0; dh < E.blk; ++dh) {
                            const int ih = ch * E.blk + dh;
                            if (ih >= E.n_h) continue;
                            const int r = iv * E.n_h + ih;
                            const T d = lds_dist[(size_t)r * EPG + tid];
                            cell = nanmax_(d, cell);
                            const T beta = rays[r * 4 + 3];
                            const T c = clip_(T(1) - d / E.ray_max, T(0), T(1));
                            const T one_c = T(1) - c;   // gamma_c = 1
                            T pen = one_c * one_c;
                            pen = (pen > T(0.001) || isnan_(pen)) ? pen : T(0.001);   // np.maximum(.., epsilon_c)
                            sum_beta += beta;
                            denom += pen * beta;
                            if (A.io.ray_dist) static_cast<T*>(A.io.ray_dist)[(size_t)env * E.n_rays + r] = d;
                        }
                    }
                    ot[16 + cv * E.n_hr + ch] = (float)clip_(cell / E.ray_max, T(0), T(1));
                }
            }
            oa = sum_beta / denom - T(1);
        } else {
            for (int k = 0; k < E.n_red; ++k) ot[16 + k] = 1.0f;   // every ray reports max_dist
            if (A.io.ray_dist)
                for (int r = 0; r < E.n_rays; ++r) static_cast<T*>(A.io.ray_dist)[(size_t)env * E.n_rays + r] = E.ray_max;
        }

        // ---------------- 11. done conditions (envs/docking3d.py:597-631) ----------------
        const int t_steps = B.t_steps[env];
        const bool c0 = dd < E.dtol;
        const bool c1 = dd > E.dmax;
        const bool c2 = (abs_(st[3]) > E.max_att) || (abs_(st[4]) > E.max_att);
        const bool c3 = t_steps >= E.max_timesteps;
        const bool c4 = collision;
        const bool done = c0 || c1 || c2 || c3 || c4;
        done_flag = done ? 1 : 0;

        // ---------------- 12. reward (envs/docking3d.py:490-595, 706-792) ----------------
        T rw[kNRew];
        const T eps = T(0.001);
        const T dd_e = dd > eps ? dd : eps;
        const T dtol_e = E.dtol > eps ? E.dtol : eps;
        const T lp_d = T(1) - clip_(log_(dd_e / E.dmax) / log_(dtol_e / E.dmax), T(0), T(1));
        rw[0] = -E.w_d * lp_d;
        const T half_pi = pi_<T>() / T(2);
        if (E.reward_set == 2) {
            // cont_goal_constraints with x_des = 0 -> max(x_goal, eps) = eps; exponents 4 (docking3d.py:523-548)
            const T ath = abs_(dth), aps = abs_(dpsi), aoa = abs_(oa);
            const T l_th = T(1) - clip_(log_((ath > eps ? ath : eps) / half_pi) / log_(eps / half_pi), T(0), T(1));
            const T l_ps = T(1) - clip_(log_((aps > eps ? aps : eps) / pi_<T>()) / log_(eps / pi_<T>()), T(0), T(1));
            const T l_oa = T(1) - clip_(log_((aoa > eps ? aoa : eps) / T(1)) / log_(eps / T(1)), T(0), T(1));
            const T ld2 = lp_d * lp_d, ld4 = ld2 * ld2;
            const T t2 = l_th * l_th, p2 = l_ps * l_ps, o2 = l_oa * l_oa;
            rw[1] = -E.w_dth * (t2 * t2) * ld4;
            rw[2] = -E.w_dpsi * (p2 * p2) * ld4;
            rw[6] = -E.w_oa * (o2 * o2) * ld4;
        } else {
            const T a = dth / half_pi, b = dpsi / pi_<T>();
            rw[1] = -E.w_dth * (a * a);
            rw[2] = -E.w_dpsi * (b * b);
            rw[6] = -E.w_oa * oa;
        }
        {
            const T a = st[3] / half_pi, b = st[4] / half_pi;
            rw[3] = -E.w_phi * (a * a);
            rw[4] = -E.w_th * (b * b);
            const T e = sqrt_(edot_norm2) / E.vel_max[3];
            rw[5] = -E.w_thdot * (e * e);
        }
        {
            T acc = T(0);
            const T nu_f = T(n_u_env);
#pragma unroll
            for (int k = 0; k < kMaxU; ++k) {
                if (k < n_u_env) {
                    const T a = abs_(act[k]) / nu_f;   // raw, un-clipped action (docking3d.py:584)
                    acc += a * a * E.w_act[k];
                }
            }
            rw[7] = -acc;
        }
        rw[8] = c0 ? E.w_done[0] : T(0);
        rw[9] = c1 ? E.w_done[1] : T(0);
        rw[10] = c2 ? E.w_done[2] : T(0);
        rw[11] = c3 ? E.w_done[3] : T(0);
        rw[12] = c4 ? E.w_done[4] : T(0);
        T reward = T(0);
#pragma unroll
        for (int k = 0; k < kNRew; ++k) reward += rw[k];

        // ---------------- outputs ----------------
        static_cast<T*>(A.io.reward)[env] = reward;
        A.io.done[env] = done_flag;
        if (A.io.conditions)
            A.io.conditions[env] = (uint8_t)((c0 ? 1 : 0) | (c1 ? 2 : 0) | (c2 ? 4 : 0) | (c3 ? 8 : 0) | (c4 ? 16 : 0));
        if (A.io.reward_terms) {
            T* rt = static_cast<T*>(A.io.reward_terms) + (size_t)env * kNRew;
#pragma unroll
            for (int k = 0; k < kNRew; ++k) rt[k] = rw[k];
        }
        if (A.io.nav) {
            T* nv = static_cast<T*>(A.io.nav) + (size_t)env * 4;
            nv[0] = dd; nv[1] = dth; nv[2] = dpsi; nv[3] = dhg;
        }

        // ---------------- state write-back / in-kernel episode reset ----------------
        T cum = g_cum[env] + reward;
        int tnext = t_steps + 1;
        if (done && E.reset_mode != 0) {
            // VecEnv auto-reset; the returned observation is the reference's reset observation = all zeros
            // (Q8, docking3d.py:269,322); the terminal one goes to terminal_obs.
            if (A.io.terminal_obs) {
                float* to = A.io.terminal_obs + (size_t)env * E.n_obs;
                for (int k = 0; k < E.n_obs; ++k) to[k] = ot[k];
            }
            for (int k = 0; k < E.n_obs; ++k) ot[k] = 0.0f;
            const int ep_next = B.episode[env] + 1;
            T curv[8];
            if (E.reset_mode == 1) {   // from the host-staged pool
                const T* pp = static_cast<const T*>(B.p_pose);
                const T* pg = static_cast<const T*>(B.p_goal);
                const T* pc = static_cast<const T*>(B.p_cur);
#pragma unroll
                for (int k = 0; k < 6; ++k) st[k] = pp[k * S + env];
#pragma unroll
                for (int k = 0; k < 4; ++k) goal[k] = pg[k * S + env];
#pragma unroll
                for (int k = 0; k < 8; ++k) curv[k] = pc[k * S + env];
                const T* pcap = static_cast<const T*>(B.p_caps);
                for (int k = 0; k < E.max_cap * 7; ++k) g_caps[(size_t)k * S + env] = pcap[(size_t)k * S + env];
                const T* psp = static_cast<const T*>(B.p_sph);
                for (int k = 0; k < E.max_sph * 4; ++k) g_sph[(size_t)k * S + env] = psp[(size_t)k * S + env];
            } else {                   // generated in-kernel (sphere fields are static per env and stay)
                generate_episode_<T>(E, B, env, ep_next, st, goal, curv);
            }
#pragma unroll
            for (int k = 6; k < 12; ++k) st[k] = T(0);
#pragma unroll
            for (int k = 0; k < kMaxU; ++k) u[k] = T(0);
#pragma unroll
            for (int k = 0; k < 4; ++k) g_goal[k * S + env] = goal[k];
            Vc = curv[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) g_cur[k * S + env] = curv[k];
            cum = T(0);
            tnext = 0;
            B.episode[env] = ep_next;
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) g_state[k * S + env] = st[k];
#pragma unroll
        for (int k = 0; k < kMaxU; ++k)
            if (k < E.n_u_max) g_u[k * S + env] = u[k];
        g_cur[0 * S + env] = Vc;
        g_cum[env] = cum;
        B.t_steps[env] = tnext;
    }

    // ---------------- coalesced row-major observation store ----------------
    __syncthreads();
    {
        const int total = n_valid * E.n_obs;
        float* dst = A.io.obs + (size_t)env0 * E.n_obs;
        for (int idx = tid; idx < total; idx += NT) dst[idx] = lds_obs[idx];
    }
}

// ------------------------------------------------------------------------------------------ launchers
template <typename T, int VK, bool RAYS, int EPG, int NT>
static int launch_one(const KernelArgs<T, 2>& a, void* stream, void* ev0, void* ev1) {
    const int groups = (a.E.n_envs + EPG - 1) / EPG;
    const size_t lds = lds_bytes<T>(EPG, a.E.max_cap, a.E.max_sph, a.E.n_rays, a.E.n_obs, RAYS);
    if (lds > 64 * 1024) {
        // gfx950 has 160 KiB of LDS per CU; more than 64 KiB per group must be requested explicitly
        static size_t granted = 0;
        if (lds > granted) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&step_kernel<T, VK, RAYS, EPG, NT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            granted = lds;
        }
    }
    if (ev0 || ev1) {
        // start/stop events attached to THIS dispatch: the elapsed time is the kernel's own duration
        hipExtLaunchKernelGGL((step_kernel<T, VK, RAYS, EPG, NT>), dim3(groups), dim3(NT), lds, (hipStream_t)stream,
                              (hipEvent_t)ev0, (hipEvent_t)ev1, 0, a);
    } else {
        hipLaunchKernelGGL((step_kernel<T, VK, RAYS, EPG, NT>), dim3(groups), dim3(NT), lds, (hipStream_t)stream, a);
    }
    return (int)hipGetLastError();
}

template <typename T, int VK>
static int launch_vk(const KernelArgs<T, 2>& a, bool has_rays, int threads, void* stream, void* ev0, void* ev1) {
    if (!has_rays) return launch_one<T, VK, false, 64, 64>(a, stream, ev0, ev1);
    if (threads >= 256) return launch_one<T, VK, true, 64, 256>(a, stream, ev0, ev1);
    return launch_one<T, VK, true, 64, 64>(a, stream, ev0, ev1);
}

template <typename T>
static int launch_t(const KernelArgs<T, 2>& a, int vk, bool has_rays, int threads, void* stream, void* ev0, void* ev1) {
    switch (vk) {
        case VK_JOY: return launch_vk<T, VK_JOY>(a, has_rays, threads, stream, ev0, ev1);
        case VK_DENSEB: return launch_vk<T, VK_DENSEB>(a, has_rays, threads, stream, ev0, ev1);
        case VK_LAUV: return launch_vk<T, VK_LAUV>(a, has_rays, threads, stream, ev0, ev1);
        case VK_MIXED: return launch_vk<T, VK_MIXED>(a, has_rays, threads, stream, ev0, ev1);
    }
    return (int)hipErrorInvalidValue;
}

int launch_step_f32(const KernelArgs<float, 2>& a, int vk, bool has_rays, int envs_per_group, int threads, void* stream,
                    void* ev0, void* ev1) {
    (void)envs_per_group;
    return launch_t<float>(a, vk, has_rays, threads, stream, ev0, ev1);
}
int launch_step_f64(const KernelArgs<double, 2>& a, int vk, bool has_rays, int envs_per_group, int threads, void* stream,
                    void* ev0, void* ev1) {
    (void)envs_per_group;
    return launch_t<double>(a, vk, has_rays, threads, stream, ev0, ev1);
}

}  // namespace dockauv
