"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/dockauv.h declares, the ctypes
struct mirrors the C struct, and creating a handle without a GPU fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dockauv.h")


@pytest.fixture(scope="module")
def lib():
    from gym_dockauv_amd.csrc import build
    build.build()
    from gym_dockauv_amd import _capi
    return _capi.load_library()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dockauv_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    from gym_dockauv_amd import _capi
    names = declared_symbols()
    assert len(names) >= 15
    bound = {s[0] for s in _capi.SYMBOLS}
    assert set(names) == bound, f"binding and header disagree: {set(names) ^ bound}"
    for n in names:
        assert hasattr(lib, n), f"{n} not exported by libdockauv.so"
    assert lib.dockauv_abi_version() == _capi.ABI_VERSION
    assert b"gfx950" in lib.dockauv_build_info()


def test_struct_layout_matches_c(tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof with the ctypes mirror."""
    from gym_dockauv_amd import _capi
    src = tmp_path / "layout.c"
    src.write_text(f'''
#include <stdio.h>
#include <stddef.h>
#include "{HEADER}"
int main(void) {{
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(dockauv_config), sizeof(dockauv_vehicle), sizeof(dockauv_step_io),
         offsetof(dockauv_config, seed), offsetof(dockauv_config, ray_table), offsetof(dockauv_config, vehicle),
         offsetof(dockauv_vehicle, lauv), sizeof(dockauv_p2p_plan), offsetof(dockauv_p2p_plan, my_flags),
         offsetof(dockauv_p2p_plan, n_dsts));
  return 0;
}}''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)]).decode().split()
    got = [C.sizeof(_capi.Config), C.sizeof(_capi.Vehicle), C.sizeof(_capi.StepIO), _capi.Config.seed.offset,
           _capi.Config.ray_table.offset, _capi.Config.vehicle.offset, _capi.Vehicle.lauv.offset,
           C.sizeof(_capi.P2PPlan), _capi.P2PPlan.my_flags.offset, _capi.P2PPlan.n_dsts.offset]
    assert [int(x) for x in out] == got


def test_create_without_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    from gym_dockauv_amd import _capi
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    with pytest.raises(_capi.DockAUVError) as ei:
        BatchedDocking3d(num_envs=4)
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value)


def test_bad_config_rejected(lib):
    from gym_dockauv_amd import _capi
    cfg = _capi.Config()
    cfg.struct_size = 1
    h = C.c_void_p()
    rc = lib.dockauv_create(C.byref(cfg), 0, C.byref(h))
    assert rc == -1 and b"mismatch" in lib.dockauv_last_error(None)


def test_oversized_batch_rejected(lib):
    """SoA row offsets are 32-bit in the kernel: a handle whose arrays would reach 4 GiB is refused at create."""
    from gym_dockauv_amd import _capi
    cfg = _capi.Config()
    cfg.struct_size = C.sizeof(_capi.Config)
    cfg.abi_version = _capi.ABI_VERSION
    cfg.n_envs = 2_000_000_000
    cfg.precision = _capi.F32
    cfg.n_vehicles = 1
    h = C.c_void_p()
    rc = lib.dockauv_create(C.byref(cfg), 0, C.byref(h))
    assert rc == -1 and b"too large for one handle" in lib.dockauv_last_error(None)
    # the p2p entry points validate their arguments before they touch a device
    assert lib.dockauv_p2p_push(None, 16, None, 1, None) == -1
    assert lib.dockauv_p2p_gather(None, None, 1, 1, None) == -1
    assert b"dockauv_p2p_gather" in lib.dockauv_last_error(None)


def test_empty_batch_and_bad_shapes_rejected_before_any_device_call(lib):
    """dockauv_create validates its configuration before it touches a device: an empty batch, an unknown precision, too
    many obstacle slots, an empty or oversized ray fan and a zero step size all come back as DOCKAUV_E_INVALID with a
    message naming the field (the reference has no batch: its envs are built one at a time, `envs/docking3d.py:74-220`)."""
    from gym_dockauv_amd import _capi

    def good():
        cfg = _capi.Config()
        cfg.struct_size = C.sizeof(_capi.Config)
        cfg.abi_version = _capi.ABI_VERSION
        cfg.n_envs = 64
        cfg.precision = _capi.F32
        cfg.n_vehicles = 1
        cfg.n_v, cfg.n_h, cfg.blocksize_reduce = 7, 9, 2
        cfg.reward_set = 1
        cfg.t_step_size = 0.1
        return cfg

    def rejected(cfg, needle):
        h = C.c_void_p()
        rc = lib.dockauv_create(C.byref(cfg), 0, C.byref(h))
        msg = lib.dockauv_last_error(None)
        assert rc == -1 and not h.value and needle in msg, (rc, msg)

    for n in (0, -5):
        cfg = good(); cfg.n_envs = n
        rejected(cfg, b"n_envs must be > 0")
    cfg = good(); cfg.precision = 7
    rejected(cfg, b"bad precision")
    cfg = good(); cfg.n_vehicles = 3
    rejected(cfg, b"n_vehicles")
    cfg = good(); cfg.max_capsules = 1000
    rejected(cfg, b"max_capsules")
    cfg = good(); cfg.max_spheres = -1
    rejected(cfg, b"max_spheres")
    cfg = good(); cfg.n_v = 0
    rejected(cfg, b"bad ray fan")
    cfg = good(); cfg.n_v, cfg.n_h = 1000, 1000
    rejected(cfg, b"bad ray fan")
    cfg = good(); cfg.blocksize_reduce = 0
    rejected(cfg, b"blocksize_reduce")
    cfg = good(); cfg.reward_set = 3
    rejected(cfg, b"reward_set")
    cfg = good(); cfg.t_step_size = 0.0
    rejected(cfg, b"t_step_size")
    cfg = good()   # everything valid but the ray table: still no device call
    rejected(cfg, b"ray_table is NULL")
    # null arguments
    h = C.c_void_p()
    assert lib.dockauv_create(None, 0, C.byref(h)) == -1 and b"null argument" in lib.dockauv_last_error(None)


def test_every_entry_point_refuses_a_null_handle(lib):
    """No entry point dereferences a NULL handle: each hands back DOCKAUV_E_INVALID (-1) -- destroy / free / close of nothing
    are no-ops (0) -- without touching a device."""
    from gym_dockauv_amd import _capi
    io = _capi.StepIO()
    d = C.c_double()
    invalid = {
        "n_obs": lambda: lib.dockauv_n_obs(None),
        "threads_per_group": lambda: lib.dockauv_threads_per_group(None),
        "n_rays": lambda: lib.dockauv_n_rays(None),
        "n_u": lambda: lib.dockauv_n_u(None),
        "field_width": lambda: lib.dockauv_field_width(None, 0),
        "set_field": lambda: lib.dockauv_set_field(None, 0, 0, 1, None),
        "get_field": lambda: lib.dockauv_get_field(None, 0, 0, 1, None),
        "reset_envs": lambda: lib.dockauv_reset_envs(None, 0, 1),
        "step": lambda: lib.dockauv_step(None, C.byref(io), None),
        "step (null io)": lambda: lib.dockauv_step(None, None, None),
        "step_sequence": lambda: lib.dockauv_step_sequence(None, C.byref(io), 1, None),
        "set_option": lambda: lib.dockauv_set_option(None, 1, 1),
        "step_host": lambda: lib.dockauv_step_host(None, C.byref(io)),
        "synchronize": lambda: lib.dockauv_synchronize(None),
        "poll_status": lambda: lib.dockauv_poll_status(None),
        "trace_enable": lambda: lib.dockauv_trace_enable(None, None, 0, 0),
        "trace_steps": lambda: lib.dockauv_trace_steps(None),
        "trace_read": lambda: lib.dockauv_trace_read(None, 0, 1, *([None] * 8)),
        "time_steps": lambda: lib.dockauv_time_steps(None, C.byref(io), None, 1, C.byref(d)),
        "step_gather_sequence": lambda: lib.dockauv_step_gather_sequence(None, C.byref(io), 1, None, 0, 0, 0, None, None),
    }
    for name, call in invalid.items():
        assert call() == -1, name
    for name, call in {"destroy": lambda: lib.dockauv_destroy(None), "p2p_free": lambda: lib.dockauv_p2p_free(None),
                       "p2p_close": lambda: lib.dockauv_p2p_close(None)}.items():
        assert call() == 0, name
