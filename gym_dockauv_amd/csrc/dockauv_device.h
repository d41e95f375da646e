// dockauv_device.h -- parameter blocks passed BY VALUE to the step kernel (kernarg segment => scalar loads,
// wave-uniform SGPR operands) and the struct-of-arrays buffer table.  Internal to libdockauv.so.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace dockauv {

constexpr int kMaxU = 8;
constexpr int kNRew = 13;
constexpr int kLoRows = 4;                       // rows of Buffers::pos_lo
constexpr int kLoState[kLoRows] = {0, 1, 2, 5};  // the state rows they belong to

// indices into VehicleP::lauv (same order as dockauv_vehicle::lauv in include/dockauv.h)
enum LauvIdx {
    L_Y_r = 0, L_Y_rr, L_Y_urf, L_Z_q, L_Z_qq, L_Z_uqf, L_M_w, L_M_ww, L_M_uw, L_N_v, L_N_vv, L_N_uv,
    L_Y_uv, L_Z_uw, L_M_uqf, L_N_urf, L_Y_uudr, L_Z_uuds, L_M_uuds, L_N_uudr, L_COUNT
};

// kernel-side vehicle kinds (template parameter VK)
enum VehKind { VK_JOY = 0, VK_DENSEB = 1, VK_LAUV = 2, VK_MIXED = 3 };

template <typename T>
struct VehicleP {
    T m;
    T gWB, gx, gy, gz;   // W-B, x_G W - x_B B, y_G W - y_B B, z_G W - z_B B   (statespace.py:353-397)
    T rg[3];
    T Ib[9];
    T ma[6];
    T dl[6], dq[6];
    T Minv[36];
    T B[6 * kMaxU];      // VK_DENSEB: dense row-major; VK_JOY: only the diagonal B[i*kMaxU+i] is read
    T ulo[kMaxU], uhalf[kMaxU];  // u = ulo + (uhi-ulo) * (clip(a)+1)/2 ; uhalf = (uhi-ulo)
    T lauv[L_COUNT];
    // structural fast path (kinetics_, SYM): coefficients of the Coriolis + added-mass polynomial, combined on the
    // host in float64.  kc = { m+ma0, m+ma1, m+ma2, m*z_G,
    //                          Iy-Iz+ma4-ma5, ma1-ma2,  Iz-Ix+ma5-ma3, ma2-ma0,  Ix-Iy+ma3-ma4, ma0-ma1 }
    T kc[10];
    int n_u;
};

template <typename T>
struct EnvP {
    int n_envs, max_timesteps, reward_set, reset_mode, scenario;
    int n_v, n_h, blk, n_vr, n_hr, n_rays, n_red, n_obs;
    int max_cap, max_sph, n_u_max;
    unsigned long long seed;
    T h, lp_alpha, mu;
    T dmax, dtol, max_att, safety;
    T vel_max[6];
    // reciprocals of configuration constants, prepared on the host in float64
    T inv_dmax, inv_log_tol, inv_log_tol_eps, inv_max_att, inv_ray_max;
    T inv_vel[6];
    T w_d, w_dth, w_dpsi, w_phi, w_th, w_thdot, w_oa;
    T w_done[5];
    T w_act[kMaxU];      // action_reward_factors[i]
    T ray_max, alpha_max, beta_max;
    // ray stage with one lane per ray (fans of <= 64 rays): lanes per env (power of two >= n_rays), the circular cone
    // that contains the fan (cos / sin of its half-angle, widened by 1e-3 rad), the sum of the obstacle-avoidance
    // weights over all rays
    // packed rows with bfloat16 observation columns (dockauv_step_io::pack_reward_done == 2): 32-bit words per row
    // (ceil(n_obs / 2) pairs + reward + done) and the multiplier that divides a word index by it (mul_hi)
    int bf16_wpr;
    unsigned bf16_magic;
    int ray_pad, ray_pad_log2;
    int device_noise;    // 1: the kernel draws the current's white noise itself (dockauv_config::device_noise)
    T fan_cos, fan_sin, sum_beta;
};

// All per-env arrays are struct-of-arrays: element (field k, env i) lives at base[k * stride + i], stride = n_envs
// rounded up to a multiple of 64 so that every row starts 256-B aligned.
struct Buffers {
    void* state;      // T [12][S]
    void* pos_lo;     // T [4][S]   low-order words of the position and of the heading (compensated accumulation, float
                      //            path: x y z psi = state[{0, 1, 2, 5}] + pos_lo[{0, 1, 2, 3}]; zero in float64)
    void* u;          // T [kMaxU][S]
    void* goal;       // T [4][S]   x y z heading
    void* cur;        // T [6][S]   V_c, dir_x, dir_y, dir_z (NED unit vector), V_min, V_max
    void* cum_reward; // T [S]
    int32_t* t_steps; // [S]
    int32_t* episode; // [S]
    uint8_t* veh_id;  // [S]
    void* cur_sigma;  // T [S]      white_noise_std of the env's current (device_noise)
    void* caps;       // T [max_cap][7][S]
    void* sph;        // T [max_sph][4][S]
    // next-episode pool (DOCKAUV_RESET_POOL)
    void* p_pose;     // T [6][S]
    void* p_goal;     // T [4][S]
    void* p_cur;      // T [6][S]
    void* p_caps;     // T [max_cap][7][S]
    void* p_sph;      // T [max_sph][4][S]
    // ray table: T [n_rays][4] = body-frame unit direction xyz, obstacle-avoidance weight beta_oa
    const void* rays;
    // fans of <= 64 rays, lane = ray: what lane l of a wave needs to know about its ray r = l % ray_pad (unit direction
    // (1, 0, 0) and weight 0 for the padding lanes r >= n_rays): T [64][4] = direction xyz, beta_oa; int32 [64] =
    // block-max cell of the ray (sensor.py:131-137)
    const void* lane_tab;
    const int32_t* lane_cell;
    // sticky status word of the handle (device memory, zero = healthy): bit 0 = a tail role of the step kernel gave up
    // waiting for its group's integrating wave (dockauv_step.hip.inc: wait_nav_).  Read by the host wherever it
    // synchronises with the stream anyway; reported as DOCKAUV_E_KERNEL.
    unsigned int* status;
    long stride;
};

struct StepIO {
    const void* actions;
    const void* noise;
    float* obs;
    void* reward;
    uint8_t* done;
    void* reward_terms;
    uint8_t* conditions;
    void* nav;
    void* ray_dist;
    float* terminal_obs;
    void* state_dot;          // T [N][12] or null
    const void* trace;        // TraceDev in device memory or null (library-owned, dockauv_trace_enable)
    long long trace_step;     // index of this step in the trace
    int pack;                 // 0: separate reward / done; 1: packed float32 rows; 2: packed rows, observation columns bfloat16
    int device_noise;         // 1: no noise array given and the handle draws the current's white noise itself
};

// ring of the last `capacity` steps of `n_rows` selected envs (include/dockauv.h: dockauv_trace_*); row-major
// [capacity][n_rows][width]
struct TraceDev {
    const int32_t* slot_of_env;   // [n_envs]: row of the env in the ring, -1 = not selected
    int n_rows, capacity;
    void* state_pre;      // T [..][12]
    void* state;          // T [..][12]
    void* state_dot;      // T [..][12]
    void* u;              // T [..][kMaxU]
    void* nu_c;           // T [..][3]
    float* obs;           // [..][n_obs]
    void* reward_terms;   // T [..][kNRew]
    uint8_t* cond;        // [..]
};

// Vehicle / reward / fan parameters (~1.7 KB in f32).  They live in a DEVICE buffer that persists across launches
// (uploaded once by dockauv_create) and are read through a constant-address-space pointer: wave-uniform scalar loads
// (s_load -> SGPR operands) that hit in L2 from the second launch on.  Passing them by value in the kernarg segment
// made every launch re-fetch 27 fresh cache lines from memory, one exposed miss per first touch (measured: a lone
// wave spent ~45 % of its life in s_waitcnt).
template <typename T, int NV>
struct ParamBlock {
    EnvP<T> E;
    VehicleP<T> V[NV];
};

// copy groups riding in the launch (dockauv_ride.h); plan == nullptr: none
struct RideLaunch {
    const void* plan = nullptr;   // dockauv_p2p_plan in DEVICE memory
    const void* src = nullptr;    // previous step's rows
    uint32_t stamp = 0, wait_stamp = 0;
    int groups = 0;               // copy groups appended to the grid
};

// Resident step sequence (dockauv_step_sequence's fast path, dockauv_step.hip.inc: step_seq_kernel): the action / row
// pointers of up to kSeqMax consecutive steps travel in the kernarg segment of ONE launch.
constexpr int kSeqMax = 64;
struct SeqArgs {
    const void* actions[kSeqMax];
    float* obs[kSeqMax];
    int n;
};

// host-side bundle of everything a launch needs
template <typename T, int NV>
struct KernelArgs {
    ParamBlock<T, NV> P;      // host copy (grid / LDS sizing, and the source of the device copy)
    const void* params_dev;   // device copy of P
    Buffers B;
    StepIO io;
    RideLaunch ride;
};

// what actually travels in the kernarg segment (29 pointers + 2 ints)
struct DevArgs {
    const void* params;
    Buffers B;
    StepIO io;
    int n_envs;   // also in the parameter block; here so that the first state loads do not wait for that block
};

// LDS hand-over layout between the env phase and the ray stage
constexpr int kCapFields = 10;   // body-frame unit axis d(3), oa_perp(3), oa_par, |ba|, r^2, oa_par - |ba|
constexpr int kSphFields = 4;    // body-frame origin - centre (3), r^2
constexpr int kPoseFields = 16;  // n_cap, n_sph, position (3), body -> NED rotation (9), may-be-hit bit masks (capsules, spheres)
constexpr int kSpecFields = 19;  // pre-drawn next episode of an env: pose (6), goal (4), current rows (8), pillar-ring draw
constexpr int kNavFields = 5;    // integrating wave -> tail roles (SHARE_NAV): distance, delta_theta, delta_psi, condition bits, ready flags;
                                 // they live in spare rows of the obstacle-avoidance sums (dockauv_step.hip.inc: lds_nav)
constexpr int kHxFields = 21;    // env phase -> tail waves: state (12), V_c, action penalty, |euler_dot|^2, collision, nu_c (3), sin/cos psi

// one-wave groups whose completed capsule records stay in registers (dockauv_step.hip.inc: regrec): the 63-ray fan (one env
// per 64-lane pass) against at most kRegCaps capsules and no spheres, float32
constexpr int kRegCaps = 5;
constexpr bool solo_regrec(int max_cap, int max_sph, int ray_pad_log2) {
    return max_sph == 0 && max_cap >= 1 && max_cap <= kRegCaps && ray_pad_log2 == 6;
}

template <typename T>
inline size_t lds_bytes(int epg, int nt, int max_cap, int max_sph, int n_obs, bool rays, int ray_pad_log2 = 0) {
    if (rays && nt / epg == 1 && sizeof(T) == 4 && solo_regrec(max_cap, max_sph, ray_pad_log2)) {
        // raw capsule rows [max_cap][7][epg]; over them, once the records are complete, the tile
        const size_t raw = (size_t)epg * 7 * max_cap * sizeof(T);
        const size_t tile = (size_t)epg * (n_obs + 2) * sizeof(float);
        return ((raw > tile ? raw : tile) + 15) & ~(size_t)15;
    }
    if (rays && nt / epg == 1) {
        // one-wave ray groups (dockauv_step.hip.inc: SOLO): no pose rows (registers); the observation tile overlays the
        // obstacle records; behind the larger of the two: obstacle-avoidance sums (2 rows), the list of active envs (1 row),
        // and the ray cells [epg][n_red] the passes accumulate
        const size_t rec = (size_t)epg * (kCapFields * max_cap + kSphFields * max_sph) * sizeof(T);
        const size_t tile = (size_t)epg * (n_obs + 2) * sizeof(float);
        size_t bytes = ((rec > tile ? rec : tile) + 15) & ~(size_t)15;
        bytes = (bytes + (size_t)epg * 3 * sizeof(T) + 15) & ~(size_t)15;
        return bytes + (size_t)epg * (n_obs - 16) * sizeof(float);
    }
    size_t t_elems = rays ? (size_t)epg * (kPoseFields + kCapFields * max_cap + kSphFields * max_sph + 4 * (nt / epg)) : 0;
    if (nt / epg >= 2) t_elems += (size_t)epg * (kHxFields + kSpecFields);
    size_t bytes = t_elems * sizeof(T);
    bytes = (bytes + 15) & ~(size_t)15;
    return bytes + (size_t)epg * (n_obs + 2) * sizeof(float);   // +2: packed reward | done columns
}

// launch one step; implemented in dockauv_kernels_f32.hip / _f64.hip.  vk = VehKind, sym = structural fast path
// (see kinetics_), has_rays = obstacles present.  ev0 / ev1: optional hipEvent_t recorded at the start / end of this
// very dispatch (hipExtLaunchKernelGGL).  Returns a hipError_t as int.
int launch_step_f32(const KernelArgs<float, 2>& a, int vk, bool sym, bool has_rays, int threads, void* stream,
                    void* ev0 = nullptr, void* ev1 = nullptr);
int launch_step_f64(const KernelArgs<double, 2>& a, int vk, bool sym, bool has_rays, int threads, void* stream,
                    void* ev0 = nullptr, void* ev1 = nullptr);

// n <= kSeqMax steps of the handle in ONE launch (every group walks its 64 envs through all of them): float32 product
// kernels of the structural fast path only (dockauv_kernels_seq.hip); hipErrorNotSupported = the caller launches the steps
// one by one.  a.io: everything but actions / obs, which come from `seq`.
int launch_sequence_f32(const KernelArgs<float, 2>& a, int vk, bool sym, bool has_rays, int threads, const SeqArgs& seq, void* stream);

#ifdef DOCKAUV_STAMPS
int read_stamps(unsigned long long* out);   // diagnostic build only
int read_span(unsigned long long* out, int groups);
#endif

}  // namespace dockauv
