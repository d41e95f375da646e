"""Error behaviour of a LIVE handle through the C ABI (include/dockauv.h: "every function returns 0 on success and a negative
DOCKAUV_E_* code on failure"): ranges outside [0, n_envs), null buffers, unknown field / option ids, a trace that was never
enabled -- each call is refused with the documented code and a message, nothing is launched, and the handle goes on stepping
exactly as an untouched twin does.  (The reference has no such surface: its envs are Python objects and raise `IndexError` /
`AttributeError` at the same places; `envs/docking3d.py:346-402` for step.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

E_INVALID, E_RANGE = -1, -4


def make(n):
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    env = BatchedDocking3d(num_envs=n, scenario="ObstaclesDocking3d", auto_reset=True)
    env.seed(11)
    env.reset()
    return env


def test_live_handle_refuses_bad_calls_and_keeps_working():
    from gym_dockauv_amd import _capi
    N = 130   # (two full groups and a ragged one)
    env, twin = make(N), make(N)
    lib, h = env._lib, env._handle
    try:
        w = lib.dockauv_field_width(h, _capi.F_STATE)
        assert w == 12
        buf = np.zeros((N + 8, w))
        p = buf.ctypes.data_as(C.c_void_p)

        def err():
            return lib.dockauv_last_error(h)

        # ---- env ranges
        for first, count in ((-1, 1), (0, N + 1), (N, 1), (N - 1, 2), (5, -1)):
            assert lib.dockauv_get_field(h, _capi.F_STATE, first, count, p) == E_RANGE and b"outside [0, 130)" in err()
            assert lib.dockauv_set_field(h, _capi.F_STATE, first, count, p) == E_RANGE
            assert lib.dockauv_reset_envs(h, first, count) == E_RANGE
        # an empty range is a no-op, at both ends of the batch
        assert lib.dockauv_get_field(h, _capi.F_STATE, 0, 0, p) == 0 and lib.dockauv_get_field(h, _capi.F_STATE, N, 0, p) == 0
        assert lib.dockauv_set_field(h, _capi.F_STATE, N, 0, p) == 0 and lib.dockauv_reset_envs(h, N, 0) == 0
        # ---- unknown ids, null buffers
        assert lib.dockauv_field_width(h, 9999) == E_INVALID and b"unknown field id" in err()
        assert lib.dockauv_get_field(h, 9999, 0, 1, p) == E_INVALID
        assert lib.dockauv_get_field(h, _capi.F_STATE, 0, 1, None) == E_INVALID and b"null argument" in err()
        assert lib.dockauv_set_field(h, _capi.F_STATE, 0, 1, None) == E_INVALID
        assert lib.dockauv_set_option(h, 9999, 1) == E_INVALID and b"unknown option" in err()
        # ---- step entry points: nothing is launched for an incomplete io
        io = _capi.StepIO()
        assert lib.dockauv_step(h, C.byref(io), None) == E_INVALID and b"actions/obs must not be NULL" in err()
        assert lib.dockauv_step(h, None, None) == E_INVALID
        assert lib.dockauv_step_host(h, C.byref(io)) == E_INVALID
        assert lib.dockauv_step_sequence(h, C.byref(io), 1, None) == E_INVALID and b"step 0" in err()
        assert lib.dockauv_step_sequence(h, C.byref(io), -1, None) == E_INVALID
        assert lib.dockauv_step_sequence(h, C.byref(io), 0, None) == 0          # an empty sequence is a no-op
        avg = C.c_double()
        assert lib.dockauv_time_steps(h, C.byref(io), None, 0, C.byref(avg)) == E_INVALID
        host_io = _capi.StepIO()
        a = np.zeros((N, env.n_u)); o = np.zeros((N, env.n_observations)); r = np.zeros(N); d = np.zeros(N, np.uint8)
        host_io.actions, host_io.obs = a.ctypes.data, o.ctypes.data
        assert lib.dockauv_step_host(h, C.byref(host_io)) == E_INVALID          # reward / done missing
        host_io.reward, host_io.done, host_io.pack_reward_done = r.ctypes.data, d.ctypes.data, 1
        assert lib.dockauv_step_host(h, C.byref(host_io)) == E_INVALID and b"device-pointer feature" in err()
        # ---- the episode-storage trace
        assert lib.dockauv_trace_read(h, 0, 1, *([None] * 8)) == E_INVALID and b"trace is not enabled" in err()
        ids = (C.c_int32 * 3)(5, 5, 7)                                          # not strictly increasing
        assert lib.dockauv_trace_enable(h, ids, 3, 16) == E_RANGE
        ids = (C.c_int32 * 2)(5, N)                                             # beyond the batch
        assert lib.dockauv_trace_enable(h, ids, 2, 16) == E_RANGE
        assert lib.dockauv_trace_enable(h, ids, 2, 0) == E_INVALID              # no capacity
        # ---- none of it touched the handle: it steps like a twin that was never asked anything
        assert lib.dockauv_poll_status(h) == 0 and lib.dockauv_synchronize(h) == 0
        np.testing.assert_array_equal(env.get_field(_capi.F_STATE), twin.get_field(_capi.F_STATE))
        rs = np.random.RandomState(3)
        for _ in range(40):
            act = rs.uniform(-1, 1, size=(N, env.n_u))
            o1, r1, d1, _ = env.step(act)
            o2, r2, d2, _ = twin.step(act)
            np.testing.assert_array_equal(o1, o2)
            np.testing.assert_array_equal(r1, r2)
            np.testing.assert_array_equal(d1, d2)
        assert np.isfinite(o1).all() and np.isfinite(r1).all()
    finally:
        env.close()
        twin.close()
