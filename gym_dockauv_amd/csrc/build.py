#!/usr/bin/env python3
"""Build libdockauv.so for gfx950 in-tree (gym_dockauv_amd/lib/).  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(os.path.dirname(HERE), "lib")
OUT = os.path.join(LIB_DIR, "libdockauv.so")
SOURCES = ["dockauv_kernels.hip", "dockauv_capi.hip"]
DEPS = SOURCES + ["dockauv_device.h", os.path.join("..", "..", "include", "dockauv.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def up_to_date() -> bool:
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(os.path.join(HERE, d)) <= t for d in DEPS + ["build.py"])


def build(force: bool = False, verbose: bool = False, extra=()) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    if not force and not extra and up_to_date():
        return OUT
    cmd = [HIPCC, *FLAGS, *extra, *SOURCES, "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, cwd=HERE, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libdockauv.so")
    if verbose and (r.stdout or r.stderr):
        print(r.stdout + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True,
                extra=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else ()))
