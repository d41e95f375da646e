/*
 * dockauv.h -- C ABI of libdockauv.so: batched docking3d step() on MI355X (gfx950).
 *
 * The reference (Erikx3/gym_dockauv) is pure Python and has no FFI; the boundary it offers is the Python class
 * API of gym_dockauv/envs/docking3d.py.  This header is the C-ABI a maintainer would bind underneath that API
 * (ctypes stub: INTEGRATION.md).  Each entry point names the reference interface it replaces.
 *
 * Conventions: every function returns 0 on success and a negative DOCKAUV_E_* code on failure;
 * dockauv_last_error() gives the message.  No exceptions cross the boundary.  All pointers are plain C pointers
 * with explicit sizes; no torch / numpy types.  One host thread drives one handle; work is stream-ordered on the
 * HIP stream passed to dockauv_step (NULL = the default stream).  The library owns the per-env state in HBM until
 * dockauv_destroy; the caller owns every buffer it passes in.
 *
 * Host-side field I/O (dockauv_set_field / dockauv_get_field) is always float64, row-major [count][width]
 * ("array of envs"); the library converts to its struct-of-arrays device layout and device precision.
 */
#ifndef DOCKAUV_H
#define DOCKAUV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DOCKAUV_ABI_VERSION 3
#define DOCKAUV_MAX_U 8          /* inputs: BlueROV2 joystick 6, BlueROV2 direct 8, LAUV 3 */
#define DOCKAUV_N_REWARDS 13     /* envs/docking3d.py:152 */
#define DOCKAUV_N_CONDITIONS 5   /* envs/docking3d.py:597-619 */
#define DOCKAUV_N_OBS_BASE 16    /* envs/docking3d.py:114 */
#define DOCKAUV_MAX_RAYS 1024
#define DOCKAUV_MAX_CAPSULES 8
#define DOCKAUV_MAX_SPHERES 16

/* error codes */
#define DOCKAUV_OK 0
#define DOCKAUV_E_INVALID (-1)   /* bad argument / config */
#define DOCKAUV_E_HIP (-2)       /* HIP runtime error (message has hipGetErrorString) */
#define DOCKAUV_E_NODEVICE (-3)  /* no usable gfx950 device */
#define DOCKAUV_E_RANGE (-4)     /* first/count outside [0, n_envs) */
#define DOCKAUV_E_KERNEL (-5)    /* a step kernel reported an internal time-out in the handle's sticky status word (an
                                    intra-group wait gave up instead of hanging the GPU); reported by the calls that
                                    synchronise: dockauv_synchronize, dockauv_get_field, dockauv_step_host,
                                    dockauv_time_steps, dockauv_trace_read -- and by dockauv_poll_status, which does
                                    not.  Results since are invalid. */

/* device arithmetic type of the path */
#define DOCKAUV_F32 0            /* product path ("within 1e-5 of the float64 reference") */
#define DOCKAUV_F64 1            /* validation path: same kernels instantiated in double */

/* vehicle model kinds */
#define DOCKAUV_VEH_CONSTB 0     /* constant B, diagonal damping: objects/vehicles/BlueROV2.py:27-88 */
#define DOCKAUV_VEH_LAUV 1       /* B(nu) ~ u^2, cross-coupled damping + lift: objects/vehicles/LAUV.py:59-110 */

/* what happens to an env whose episode ended inside dockauv_step */
#define DOCKAUV_RESET_NONE 0     /* nothing: caller resets (single-env gym.Env semantics, docking3d.py:222) */
#define DOCKAUV_RESET_POOL 1     /* in-kernel reset from the host-staged next-episode pool (VecEnv auto-reset) */
#define DOCKAUV_RESET_DEVICE 2   /* in-kernel scenario generation with a counter RNG (throughput mode) */

/* scenario ids for DOCKAUV_RESET_DEVICE (envs/docking3d.py:795-988) */
#define DOCKAUV_SCN_SIMPLE 0
#define DOCKAUV_SCN_SIMPLE_CURRENT 1
#define DOCKAUV_SCN_CAPSULE 2
#define DOCKAUV_SCN_CAPSULE_CURRENT 3
#define DOCKAUV_SCN_OBSTACLES 4
#define DOCKAUV_SCN_OBSTACLES_NOCAP 5
#define DOCKAUV_SCN_OBSTACLES_CURRENT 6
#define DOCKAUV_SCN_SPHERES 7    /* build-defined: SimpleDocking3d + max_spheres spheres in a 3..12 m shell */

/*
 * One vehicle type.  Replaces the per-instance constants of objects/statespace.py:58-197 (StateSpace) and the
 * B / D / u_bound overrides of the two vehicle classes.  The host computes M_inv exactly like the reference
 * (numpy.linalg.inv of M_RB + M_A in float64, statespace.py:190-197) and hands the numbers over.
 */
typedef struct dockauv_vehicle {
    int32_t kind;                 /* DOCKAUV_VEH_* */
    int32_t n_u;                  /* number of inputs, <= DOCKAUV_MAX_U */
    double m;                     /* mass */
    double W, BY;                 /* weight m*g (statespace.py:86-88), buoyancy */
    double r_G[3], r_B[3];        /* CG / CB offsets from CO */
    double I_b[9];                /* inertia about CO, row-major (statespace.py:105-117) */
    double ma_diag[6];            /* diagonal of M_A = -(X_udot..N_rdot) (statespace.py:164-187) */
    double d_lin[6], d_quad[6];   /* X_u..N_r, X_uu..N_rr (statespace.py:288-351) */
    double M_inv[36];             /* row-major */
    double B[6 * DOCKAUV_MAX_U];  /* row-major 6 x DOCKAUV_MAX_U, constant-B kinds only (BlueROV2.py:34-72) */
    double u_lo[DOCKAUV_MAX_U], u_hi[DOCKAUV_MAX_U]; /* u_bound columns (BlueROV2.py:44-50, LAUV.py:103-110) */
    /* LAUV extras (LAUV.py:32-55), order: Y_r Y_rr Y_urf | Z_q Z_qq Z_uqf | M_w M_ww M_uwb+M_uwf |
       N_v N_vv N_uvb+N_uvf | Y_uvb+Y_uvf  Z_uwb+Z_uwf  M_uqf  N_urf | Y_uudr Z_uuds M_uuds N_uudr */
    double lauv[20];
} dockauv_vehicle;

/*
 * Environment batch configuration.  Replaces the reads of the config dict in BaseDocking3d.__init__
 * (envs/docking3d.py:48-220; key schema config/env_config.py:20-91) and Radar.__init__ (objects/sensor.py:43-87).
 */
typedef struct dockauv_config {
    uint32_t struct_size;          /* sizeof(dockauv_config): ABI check */
    uint32_t abi_version;          /* DOCKAUV_ABI_VERSION */
    int32_t n_envs;                /* envs owned by this handle (this GPU's shard) */
    int32_t precision;             /* DOCKAUV_F32 / DOCKAUV_F64 */
    int32_t n_vehicles;            /* 1, or 2 for a per-env vehicle id (mixed batch) */
    int32_t reset_mode;            /* DOCKAUV_RESET_* */
    int32_t scenario;              /* DOCKAUV_SCN_* (used by DOCKAUV_RESET_DEVICE only) */
    int32_t max_timesteps;         /* "max_timesteps" */
    int32_t reward_set;            /* "reward_set": 1 or 2 (docking3d.py:519-582) */
    int32_t max_capsules;          /* per-env capsule slots, 0..DOCKAUV_MAX_CAPSULES */
    int32_t max_spheres;           /* per-env sphere slots, 0..DOCKAUV_MAX_SPHERES */
    int32_t n_v, n_h;              /* ray fan: vertical x horizontal rays (sensor.py:56-63) */
    int32_t blocksize_reduce;      /* "blocksize_reduce" (sensor.py:136-137) */
    int32_t envs_per_group;        /* 0 = auto (64); -1 = test hook: general (non-structural) kinetics expressions */
    int32_t threads_per_group;     /* 0 = auto; 64/128 without obstacles, 64/256/512 with: waves per 64-env group */
    uint64_t seed;                 /* DOCKAUV_RESET_DEVICE: counter-RNG key */
    double t_step_size;            /* "t_step_size" */
    double lowpass_T1;             /* 0.2 (objects/auvsim.py:40) */
    double current_mu;             /* Gauss-Markov mu, 0.005 in every shipped scenario (docking3d.py:820) */
    double max_dist_from_goal, max_attitude, dist_goal_reached_tol;
    double vel_max[6];             /* u_max v_max w_max p_max q_max r_max */
    double safety_radius;          /* 1.0, hard-wired in the reference (objects/auvsim.py:43) */
    double w_d, w_delta_theta, w_delta_psi, w_phi, w_theta, w_Thetadot, w_oa; /* "reward_factors" */
    double w_done[DOCKAUV_N_CONDITIONS];  /* w_goal w_deltad_max w_Theta_max w_t_max w_col (docking3d.py:181-187) */
    double action_reward_factors[DOCKAUV_MAX_U]; /* scalar config value broadcast by the host (docking3d.py:584) */
    double radar_max_dist;         /* "radar.max_dist" */
    double radar_alpha_max, radar_beta_max; /* alpha/2, beta/2 (sensor.py:53-54) */
    /* [n_v*n_h][4] row-major, ray index = iv*n_h + ih: unit body-frame direction normalise(1, sin beta, sin alpha)
     * (sensor.py:66-71) and the obstacle-avoidance weight beta_oa (docking3d.py:786-788).  Read during create only. */
    const double* ray_table;
    dockauv_vehicle vehicle[2];
    /* Gauss-Markov current with sigma > 0 (objects/current.py:88, w = np.random.normal(0, sigma)) when the caller
     * passes no noise array: 0 = w = 0 (what every shipped scenario has: white_noise_std = 0, docking3d.py:820);
     * 1 = the kernel draws w = sigma_env * N(0, 1) itself -- Philox4x32-10 counter (env, episode, t_steps, 1), key =
     * seed, Box-Muller on the first two words (oracle/philox_ref.py: philox_normal); sigma_env = field
     * DOCKAUV_F_CURRENT_SIGMA.  dockauv_step_io.noise, when given, always wins (parity mode). */
    int32_t device_noise;
    int32_t reserved0;
} dockauv_config;

typedef struct dockauv_env_s* dockauv_handle;

/* per-env fields addressable from the host (width = doubles per env) */
#define DOCKAUV_F_STATE 0        /* 12: eta(6), nu_r(6)                       (objects/auvsim.py:37,162-246) */
#define DOCKAUV_F_U 1            /* DOCKAUV_MAX_U: filtered input u           (objects/auvsim.py:277-284) */
#define DOCKAUV_F_GOAL 2         /* 4: goal x y z, heading_goal_reached       (docking3d.py:194,202) */
#define DOCKAUV_F_CURRENT 3      /* 5: V_c V_min V_max alpha beta             (objects/current.py:20-31) */
#define DOCKAUV_F_TSTEPS 4       /* 1: steps in this episode                  (docking3d.py:139) */
#define DOCKAUV_F_CAPSULES 5     /* max_capsules*7: bot xyz, top xyz, radius (radius <= 0: unused slot) */
#define DOCKAUV_F_SPHERES 6      /* max_spheres*4: centre xyz, radius (radius <= 0: unused slot) */
#define DOCKAUV_F_VEHICLE_ID 7   /* 1: index into config.vehicle[] (mixed batches) */
#define DOCKAUV_F_CUM_REWARD 8   /* 1: cumulative reward of the running episode (docking3d.py:156) */
#define DOCKAUV_F_EPISODE 9      /* 1: episode counter (docking3d.py:141) */
#define DOCKAUV_F_CURRENT_SIGMA 10 /* 1: white_noise_std of the env's current (objects/current.py:31); read by device_noise */
/* next-episode pool (DOCKAUV_RESET_POOL): same layouts */
#define DOCKAUV_F_POOL_POSE 16       /* 6: position, attitude */
#define DOCKAUV_F_POOL_GOAL 17       /* 4 */
#define DOCKAUV_F_POOL_CURRENT 18    /* 5 */
#define DOCKAUV_F_POOL_CAPSULES 19   /* max_capsules*7 */
#define DOCKAUV_F_POOL_SPHERES 20    /* max_spheres*4 */

/*
 * Inputs / outputs of one step.  Replaces the arguments and return tuple of BaseDocking3d.step
 * (envs/docking3d.py:346-402) for a batch.  "T" = float (DOCKAUV_F32) or double (DOCKAUV_F64).
 * In dockauv_step every pointer is a DEVICE pointer; in dockauv_step_host every pointer is a HOST pointer.
 * Nullable members may be NULL.
 */
typedef struct dockauv_step_io {
    const void* actions;     /* T [n_envs][n_u_max] row-major, raw policy output (clipped inside, auvsim.py:74) */
    const void* noise;       /* nullable, T [n_envs]: w_k ~ N(0, sigma) of Current.sim (current.py:88); NULL = 0 */
    float* obs;              /* float32 [n_envs][n_obs] row-major (docking3d.py:462-488); with pack_reward_done = 1:
                                float32 [n_envs][n_obs + 2] = obs | reward | done(0.0/1.0), one dense buffer so that a
                                single all-gather ships everything a learner needs; with pack_reward_done = 2 the
                                observation columns are bfloat16 (round to nearest even), two per 32-bit word:
                                uint32 [n_envs][ceil(n_obs / 2) + 2] = obs pairs (low half first; an odd n_obs is padded
                                with 0) | reward (float32) | done (float32) -- half the bytes over xGMI */
    void* reward;            /* T [n_envs] (docking3d.py:593); nullable when pack_reward_done */
    uint8_t* done;           /* [n_envs] 0/1 (docking3d.py:630); nullable when pack_reward_done */
    void* reward_terms;      /* nullable, T [n_envs][13]: last_reward_arr (docking3d.py:513-588) */
    uint8_t* conditions;     /* nullable, [n_envs]: bit i = condition i (docking3d.py:608-619) */
    void* nav;               /* nullable, T [n_envs][4]: delta_d, delta_theta, delta_psi, delta_heading_goal */
    void* ray_dist;          /* nullable, T [n_envs][n_rays]: clamped intersec_dist (sensor.py:113-118) */
    float* terminal_obs;     /* nullable, float32 [n_envs][n_obs]: written only where done (auto-reset modes) */
    void* state_dot;         /* nullable, T [n_envs][12]: AUVSim._state_dot, the right-hand side at the new state with the
                                new input (objects/auvsim.py:108), what EpisodeDataStorage logs as "states_dot"
                                (utils/datastorage.py:272,299) */
    int32_t pack_reward_done; /* 0 / 1 / 2, see obs */
    int32_t reserved;
} dockauv_step_io;

/* library / build info; callable without a GPU */
int dockauv_abi_version(void);
const char* dockauv_build_info(void);
/* message of the last failure on this handle (h may be NULL: last failure of create) */
const char* dockauv_last_error(dockauv_handle h);

/* BaseDocking3d.__init__ (docking3d.py:48-220) for a batch: allocates the SoA state in HBM of `device` */
int dockauv_create(const dockauv_config* cfg, int device, dockauv_handle* out);
int dockauv_destroy(dockauv_handle h);

/* derived sizes: n_obs = 16 + n_rays_reduced (docking3d.py:114-115), n_rays (sensor.py:63), n_u_max */
int dockauv_n_obs(dockauv_handle h);
/* waves x 64 = threads per 64-env group the handle's step kernels run with (dockauv_config::threads_per_group, or the
 * library's choice for this workload and batch size when that was 0).  No reference counterpart: a tuning read-out. */
int dockauv_threads_per_group(dockauv_handle h);
int dockauv_n_rays(dockauv_handle h);
int dockauv_n_u(dockauv_handle h);

/* state access: auv.state / position / attitude setters, goal_location, Current(...), capsules, spheres
 * (docking3d.py:803-988 generate_environment; objects/auvsim.py:174-195).  src/dst: double [count][width]. */
int dockauv_field_width(dockauv_handle h, int field);
int dockauv_set_field(dockauv_handle h, int field, int first, int count, const double* src);
int dockauv_get_field(dockauv_handle h, int field, int first, int count, double* dst);

/* AUVSim.reset + counters of BaseDocking3d.reset (objects/auvsim.py:55-65, docking3d.py:262-276) for envs
 * [first, first+count): state, u, t_steps, cumulative reward -> 0; episode += 1.  Pose/goal/... are then set
 * with dockauv_set_field. */
int dockauv_reset_envs(dockauv_handle h, int first, int count);

/* BaseDocking3d.step (docking3d.py:346-402) for all envs of the handle; device pointers, asynchronous on stream */
int dockauv_step(dockauv_handle h, const dockauv_step_io* io, void* hip_stream);
/* `n` consecutive steps, step i with ios[i] (device pointers), queued back-to-back on `hip_stream` by one call:
 * an open-loop action sequence (the manual / scripted loops of train.py:108-117, 238) without a host round trip per
 * step.  Equivalent to n calls of dockauv_step. */
int dockauv_step_sequence(dockauv_handle h, const dockauv_step_io* ios, int n, void* hip_stream);
/* Fast path of dockauv_step_sequence (ABI 3): when the steps are what the float32 product kernels serve (mandatory outputs as
 * packed rows of one kind, reset mode NONE / DEVICE, reward set 1, fans of 9-16 or 33-64 rays, no logging) they run as
 * RESIDENT launches of up to 64 steps each -- every 64-env group walks its envs through all steps of the launch, step k
 * reading ios[k].actions and writing ios[k].obs, with no launch boundary in between.  The bytes written are exactly those of
 * n single launches (tests/test_gpu_reset.py); what differs is WHEN: rows of different groups belong to different steps
 * while the call is in flight, so the buffers must not be consumed before the call has completed on the stream (an
 * open-loop sequence; a policy in the loop uses dockauv_step).  On by default; dockauv_set_option switches it per handle. */
#define DOCKAUV_OPT_SEQUENCE_RESIDENT 1   /* value 0: dockauv_step_sequence launches its steps one by one */
int dockauv_set_option(dockauv_handle h, int option, int value);
/* same with host pointers (staged through the library's pinned buffers; synchronous) */
int dockauv_step_host(dockauv_handle h, const dockauv_step_io* io);
/* block until everything queued on the handle's last-used stream is done */
int dockauv_synchronize(dockauv_handle h);
/* the handle's sticky kernel status WITHOUT any synchronisation (the word lives in host-coherent memory): 0, or
 * DOCKAUV_E_KERNEL once a step kernel that has already run gave up an internal wait.  For device-resident rollouts that
 * never call a synchronising entry point (the reference has no counterpart: its step() raises in the caller's thread);
 * cheap enough for every step. */
int dockauv_poll_status(dockauv_handle h);

/*
 * Episode storage for selected envs of a batch (utils/datastorage.py:164-343 EpisodeDataStorage, hooked at
 * docking3d.py:252-259,363-364): a ring of the last `capacity` steps of `n_rows` chosen envs, kept in HBM and written
 * by the step kernel itself, so that a device-resident rollout (no host round trip per step) can still hand the
 * reference's per-step arrays to its post-analysis.  Per step and selected env the kernel records: the state the step
 * started from, the new state, _state_dot, the filtered input u, nu_c (body frame, first three), the observation
 * BEFORE any auto-reset zeroing, the 13 reward terms and the condition bits.  Row of step k: k % capacity.
 * dockauv_trace_enable(h, env_ids, n_rows, capacity): env_ids host array, strictly increasing; n_rows = 0 switches
 *   the trace off and frees the ring.  The step counter restarts at 0.  Every call first releases the ring of an earlier
 *   enable -- also a call that is then refused (DOCKAUV_E_INVALID: no ids / capacity < 1; DOCKAUV_E_RANGE: ids not strictly
 *   increasing inside [0, n_envs)): after a refused call the trace is off.
 * dockauv_trace_steps(h): steps recorded since enable (or a negative error code).
 * dockauv_trace_read(h, first_step, n_steps, ...): copies steps [first_step, first_step + n_steps) -- they must still be
 *   in the ring -- to host arrays [n_steps][n_rows][width] (float64, obs float32, conditions uint8); any of the output
 *   pointers may be NULL.  Synchronises with the handle's last-used stream.
 */
int dockauv_trace_enable(dockauv_handle h, const int32_t* env_ids, int n_rows, int capacity);
long long dockauv_trace_steps(dockauv_handle h);
int dockauv_trace_read(dockauv_handle h, long long first_step, int n_steps, double* state_pre /*12*/, double* state /*12*/,
                       double* state_dot /*12*/, double* u /*DOCKAUV_MAX_U*/, double* nu_c /*3*/, float* obs /*n_obs*/,
                       double* reward_terms /*13*/, uint8_t* conditions /*1*/);

/* measurement helper (bench.py): run `steps` step launches back-to-back on `stream` re-using the same io, each
 * dispatch carrying its own start/stop HIP events ON THAT STREAM; returns the average KERNEL duration in microseconds
 * (launch gaps excluded -- comparable with rocprofv3 --kernel-trace). */
int dockauv_time_steps(dockauv_handle h, const dockauv_step_io* io, void* hip_stream, int steps, double* avg_us);

/*
 * Multi-GPU: peer-to-peer gather of the packed [obs | reward | done] rows over xGMI ("keep the
 * collective pluggable"; the default transport is one RCCL all-gather issued by the host through torch.distributed,
 * gym_dockauv_amd/parallel.py).  The reference is single-process and has no counterpart; these entry points replace
 * the concatenation of per-env observations a vectorised caller does on the host (train.py:64-71 consumes it).
 * One process per GPU.  Every rank owns a gather buffer and a flag array, exports them as IPC handles (the host
 * exchanges the 64-byte handles over any channel it has), opens its peers' handles, and per step
 *   1. dockauv_p2p_push: copies its rows into its slice of every rank's gather buffer (one kernel, system-scope
 *      write-through 16-byte stores over the fabric, each wave waits for its acknowledgements);
 *   2. dockauv_p2p_signal_wait: raises stamp t in every peer's flag array and waits -- bounded -- until every peer's
 *      stamp has reached `wait_stamp` in its own.
 * All calls are asynchronous on `hip_stream`.  A wait that runs out of `max_spins` sets bit r (r = late rank) in
 * status[0] and stores the stamp in status[1]; every later wait then returns at once (the grid always drains).
 */
#define DOCKAUV_P2P_HANDLE_BYTES 64
#define DOCKAUV_P2P_MAX_PEERS 15
/* device memory a peer process can map; uncached != 0: fine-grained (what a peer stores is seen by a kernel that is
 * already running: required for flag arrays; far too slow for gather buffers, which are read by later kernels only).
 * `handle` (nullable) receives DOCKAUV_P2P_HANDLE_BYTES bytes. */
int dockauv_p2p_alloc(int device, size_t bytes, int uncached, void** dev_ptr, unsigned char* handle);
int dockauv_p2p_free(void* dev_ptr);
/* map / unmap a peer's allocation on `device` */
int dockauv_p2p_open(int device, const unsigned char* handle, void** dev_ptr);
int dockauv_p2p_close(void* dev_ptr);
/* copy `bytes` from src (16-byte aligned) to each of dsts[0..n_dsts) (local or peer-mapped, 16-byte aligned) */
int dockauv_p2p_push(const void* src, size_t bytes, void* const* dsts, int n_dsts, void* hip_stream);
/* peer_slots[p] = &flags_of_peer_p[my_rank]; my_flags = this rank's flag array [world]; status = uint32 [2] in device
 * memory of this rank; stamp / wait_stamp: 0 = skip that half; stamps compare modulo 2^32 */
int dockauv_p2p_signal_wait(uint32_t* const* peer_slots, int n_peers, const uint32_t* my_flags, int world, int my_rank,
                            uint32_t stamp, uint32_t wait_stamp, uint64_t max_spins, uint32_t* status,
                            void* hip_stream);

/*
 * The same gather as ONE kernel: the blocks copy; the block that finishes last (device counter) raises `stamp` at the
 * peers and waits for `wait_stamp`.  A plan is a plain description of one rank's view of one gather buffer.
 */
typedef struct dockauv_p2p_plan {
    void* dsts[DOCKAUV_P2P_MAX_PEERS + 1];        /* this rank's slice in every rank's gather buffer (own included) */
    uint32_t* peer_slots[DOCKAUV_P2P_MAX_PEERS];  /* &flags_of_peer_p[my_rank] */
    const uint32_t* my_flags;                     /* [world], written by the peers */
    uint32_t* status;                             /* [2], this rank */
    uint32_t* counter;                            /* [1], this rank, zero between gathers */
    uint64_t bytes;                               /* size of the slice */
    uint64_t max_spins;
    int32_t n_dsts, n_peers, world, my_rank;
} dockauv_p2p_plan;
int dockauv_p2p_gather(const dockauv_p2p_plan* plan, const void* src, uint32_t stamp, uint32_t wait_stamp,
                       void* hip_stream);
/*
 * n steps with their gathers, queued by one host call: step i (global step number t0 + i) writes its packed rows to
 * ios[i].obs, which must be row buffer (t0 + i) % 2 of the caller's two; its gather uses plan (t0 + i) % n_plans and
 * stamp t0 + i + 1 (raised at the peers and awaited from them).  On return (asynchronous) `compute_stream` is ordered
 * after every gather queued here.
 * gather_stream == compute_stream, lag 0: step kernel, gather kernel, step kernel, ... in order: every rank holds all
 *   rows of step t before step t + 1 starts.
 * gather_stream == compute_stream, lag 1: the gather of step t RIDES in the grid of step kernel t + 1 (extra workgroups
 *   behind the step groups push the previous rows while the step groups integrate: the fabric transfer is hidden
 *   behind the arithmetic, one launch per step); the last gather gets a kernel of its own.  Needs >= 4 plans, float
 *   kernels, slices that are multiples of 16 bytes.
 * gather_stream != compute_stream (lag 0 or 1 = which stamp a gather awaits): gather kernels on a second stream beside the next step kernel; the step
 *   kernel that next writes the same row buffer waits for that gather.  Five stream/event calls per step on the host
 *   and two cross-stream dependencies: measured slower than one stream at every size on one GPU.
 */
int dockauv_step_gather_sequence(dockauv_handle h, const dockauv_step_io* ios, int n, const dockauv_p2p_plan* plans,
                                 int n_plans, uint64_t t0, int lag, void* compute_stream, void* gather_stream);

#ifdef __cplusplus
}
#endif
#endif /* DOCKAUV_H */
