"""
ctypes binding of libdockauv.so (include/dockauv.h).  This is the ONLY compute path of the package: if the HIP
library is missing or no MI355X is visible, loading / creating fails loudly -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdockauv.so")

ABI_VERSION = 3
OPT_SEQUENCE_RESIDENT = 1
MAX_U = 8
N_REWARDS = 13
N_CONDITIONS = 5
N_OBS_BASE = 16
MAX_CAPSULES = 8
MAX_SPHERES = 16

F32, F64 = 0, 1
VEH_CONSTB, VEH_LAUV = 0, 1
RESET_NONE, RESET_POOL, RESET_DEVICE = 0, 1, 2

(F_STATE, F_U, F_GOAL, F_CURRENT, F_TSTEPS, F_CAPSULES, F_SPHERES, F_VEHICLE_ID, F_CUM_REWARD, F_EPISODE,
 F_CURRENT_SIGMA) = range(11)
(F_POOL_POSE, F_POOL_GOAL, F_POOL_CURRENT, F_POOL_CAPSULES, F_POOL_SPHERES) = range(16, 21)

SCN = {"SimpleDocking3d": 0, "SimpleCurrentDocking3d": 1, "CapsuleDocking3d": 2, "CapsuleCurrentDocking3d": 3,
       "ObstaclesDocking3d": 4, "ObstaclesNoCapDocking3d": 5, "ObstaclesCurrentDocking3d": 6, "SphereDocking3d": 7}


class Vehicle(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("n_u", C.c_int32),
        ("m", C.c_double), ("W", C.c_double), ("BY", C.c_double),
        ("r_G", C.c_double * 3), ("r_B", C.c_double * 3),
        ("I_b", C.c_double * 9),
        ("ma_diag", C.c_double * 6),
        ("d_lin", C.c_double * 6), ("d_quad", C.c_double * 6),
        ("M_inv", C.c_double * 36),
        ("B", C.c_double * (6 * MAX_U)),
        ("u_lo", C.c_double * MAX_U), ("u_hi", C.c_double * MAX_U),
        ("lauv", C.c_double * 20),
    ]


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
        ("n_envs", C.c_int32), ("precision", C.c_int32), ("n_vehicles", C.c_int32), ("reset_mode", C.c_int32),
        ("scenario", C.c_int32), ("max_timesteps", C.c_int32), ("reward_set", C.c_int32),
        ("max_capsules", C.c_int32), ("max_spheres", C.c_int32),
        ("n_v", C.c_int32), ("n_h", C.c_int32), ("blocksize_reduce", C.c_int32),
        ("envs_per_group", C.c_int32), ("threads_per_group", C.c_int32),
        ("seed", C.c_uint64),
        ("t_step_size", C.c_double), ("lowpass_T1", C.c_double), ("current_mu", C.c_double),
        ("max_dist_from_goal", C.c_double), ("max_attitude", C.c_double), ("dist_goal_reached_tol", C.c_double),
        ("vel_max", C.c_double * 6),
        ("safety_radius", C.c_double),
        ("w_d", C.c_double), ("w_delta_theta", C.c_double), ("w_delta_psi", C.c_double), ("w_phi", C.c_double),
        ("w_theta", C.c_double), ("w_Thetadot", C.c_double), ("w_oa", C.c_double),
        ("w_done", C.c_double * N_CONDITIONS),
        ("action_reward_factors", C.c_double * MAX_U),
        ("radar_max_dist", C.c_double), ("radar_alpha_max", C.c_double), ("radar_beta_max", C.c_double),
        ("ray_table", C.POINTER(C.c_double)),
        ("vehicle", Vehicle * 2),
        ("device_noise", C.c_int32), ("reserved0", C.c_int32),
    ]


class StepIO(C.Structure):
    _fields_ = [
        ("actions", C.c_void_p), ("noise", C.c_void_p), ("obs", C.c_void_p), ("reward", C.c_void_p),
        ("done", C.c_void_p), ("reward_terms", C.c_void_p), ("conditions", C.c_void_p), ("nav", C.c_void_p),
        ("ray_dist", C.c_void_p), ("terminal_obs", C.c_void_p), ("state_dot", C.c_void_p),
        ("pack_reward_done", C.c_int32), ("reserved", C.c_int32),
    ]


P2P_HANDLE_BYTES = 64
P2P_MAX_PEERS = 15


class P2PPlan(C.Structure):
    _fields_ = [
        ("dsts", C.c_void_p * (P2P_MAX_PEERS + 1)), ("peer_slots", C.c_void_p * P2P_MAX_PEERS),
        ("my_flags", C.c_void_p), ("status", C.c_void_p), ("counter", C.c_void_p),
        ("bytes", C.c_uint64), ("max_spins", C.c_uint64),
        ("n_dsts", C.c_int32), ("n_peers", C.c_int32), ("world", C.c_int32), ("my_rank", C.c_int32),
    ]


# every symbol include/dockauv.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("dockauv_abi_version", C.c_int, []),
    ("dockauv_build_info", C.c_char_p, []),
    ("dockauv_last_error", C.c_char_p, [C.c_void_p]),
    ("dockauv_create", C.c_int, [C.POINTER(Config), C.c_int, C.POINTER(C.c_void_p)]),
    ("dockauv_destroy", C.c_int, [C.c_void_p]),
    ("dockauv_n_obs", C.c_int, [C.c_void_p]),
    ("dockauv_threads_per_group", C.c_int, [C.c_void_p]),
    ("dockauv_n_rays", C.c_int, [C.c_void_p]),
    ("dockauv_n_u", C.c_int, [C.c_void_p]),
    ("dockauv_field_width", C.c_int, [C.c_void_p, C.c_int]),
    ("dockauv_set_field", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("dockauv_get_field", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("dockauv_reset_envs", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("dockauv_step", C.c_int, [C.c_void_p, C.POINTER(StepIO), C.c_void_p]),
    ("dockauv_step_sequence", C.c_int, [C.c_void_p, C.POINTER(StepIO), C.c_int, C.c_void_p]),
    ("dockauv_set_option", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("dockauv_step_host", C.c_int, [C.c_void_p, C.POINTER(StepIO)]),
    ("dockauv_synchronize", C.c_int, [C.c_void_p]),
    ("dockauv_poll_status", C.c_int, [C.c_void_p]),
    ("dockauv_trace_enable", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    ("dockauv_trace_steps", C.c_longlong, [C.c_void_p]),
    ("dockauv_trace_read", C.c_int, [C.c_void_p, C.c_longlong, C.c_int] + [C.c_void_p] * 8),
    ("dockauv_time_steps", C.c_int, [C.c_void_p, C.POINTER(StepIO), C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    ("dockauv_p2p_alloc", C.c_int, [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.c_char_p]),
    ("dockauv_p2p_free", C.c_int, [C.c_void_p]),
    ("dockauv_p2p_open", C.c_int, [C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    ("dockauv_p2p_close", C.c_int, [C.c_void_p]),
    ("dockauv_p2p_push", C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    ("dockauv_p2p_signal_wait", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_uint32,
                                          C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]),
    ("dockauv_p2p_gather", C.c_int, [C.POINTER(P2PPlan), C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    ("dockauv_step_gather_sequence", C.c_int, [C.c_void_p, C.POINTER(StepIO), C.c_int, C.POINTER(P2PPlan), C.c_int,
                                               C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]),
]

_lib: Optional[C.CDLL] = None


class DockAUVError(RuntimeError):
    pass


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libdockauv.so and bind every declared symbol.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("DOCKAUV_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise DockAUVError(f"{p} not found: build it first (python -c 'import __graft_entry__ as g; g.build()' or "
                           f"make -C gym_dockauv_amd/csrc).  There is no CPU fallback.")
    lib = C.CDLL(p)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.dockauv_abi_version() != ABI_VERSION:
        raise DockAUVError(f"libdockauv ABI {lib.dockauv_abi_version()} != binding ABI {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(lib: C.CDLL, handle, rc: int, what: str) -> None:
    if rc != 0:
        msg = lib.dockauv_last_error(handle)
        raise DockAUVError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
