#!/usr/bin/env python3
"""Build libdockauv.so for gfx950 in-tree (gym_dockauv_amd/lib/).  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(os.path.dirname(HERE), "lib")
OUT = os.path.join(LIB_DIR, os.environ.get("DOCKAUV_LIB_NAME", "libdockauv.so"))
SOURCES = ["dockauv_kernels_f32.hip", "dockauv_kernels_f64.hip", "dockauv_kernels_seq.hip", "dockauv_capi.hip", "dockauv_p2p.hip"]
DEPS = SOURCES + ["dockauv_step.hip.inc", "dockauv_device.h", "dockauv_ride.h", os.path.join("..", "..", "include", "dockauv.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: the SLP vectoriser packs independent f32 ops into v_pk_* pairs; on this kernel that costs ~370
# v_mov to build the aligned register pairs and pushes the step kernel from 113 to ~200 VGPRs (4 -> 2 waves/SIMD)
# -ffp-contract=on: a multiply-add is fused where the SOURCE writes it inside one expression, and nowhere else.  hipcc's default
# ("fast": the backend fuses whatever it finds after inlining and CSE) gave two instantiations of the same source -- the step
# kernel and the resident sequence kernel of the mixed batch -- results that differed in the last bit (round 4); with "on" what
# a step computes is a property of the source: 12-17 more VALU instructions of ~2 500 per kernel, no change in registers.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
         "-fno-slp-vectorize", "-Werror=uninitialized", "-ffp-contract=on"]

# The float64 validation kernels of 512-thread groups need all 256 VGPRs, spill ~100 of them to scratch AND spill
# 200-460 SGPRs.  With the default "SGPR spills live in lanes of reserved VGPRs" the general-expression (non-SYM) ray
# kernel step_kernel<double, VK_JOY, false, true, 64, 512> computed a wrong wave-uniform coefficient on gfx950 /
# ROCm 7.2: every env of the batch got the same wrong surge / pitch-rate derivative (rows 0 and 4, the two rows coupled
# by z_G and M^-1[0][4]), deterministically.  Same source, same box: 256-thread groups (no VGPR spills) exact; SGPR spills
# to scratch memory (the flag below) exact (profiles/r2_f64_general_path.txt, scripts/diag/general_vs_sym.py).  Nothing in
# the source depends on the spill strategy, so the validation translation unit is built with SGPR spills in memory;
# the float32 product kernels (<= 128 VGPRs, no VGPR spills) keep the default.
PER_SOURCE_FLAGS = {"dockauv_kernels_f64.hip": ["-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0"]}


# Test library with ONE injected fault (never the product): the float32 kernels built with
# -DDOCKAUV_FAULT_INJECT_DROP_NAV_FLAG=1 (group 0's integrating wave never raises its hand-over flag), linked with the
# product's other objects.  tests/test_gpu_status.py uses it to watch a bounded wait give up, the grid drain and
# DOCKAUV_E_KERNEL reach the host.
FAULT_LIB = os.path.join(LIB_DIR, "libdockauv_faultinject.so")
FAULT_SRC, FAULT_FLAGS = "dockauv_kernels_f32.hip", ["-DDOCKAUV_FAULT_INJECT_DROP_NAV_FLAG=1"]


def up_to_date() -> bool:
    variant = os.environ.get("DOCKAUV_LIB_NAME") is not None      # (scripts/build_variant.py: no fault library)
    for lib in ([OUT] if variant else [OUT, FAULT_LIB]):
        if not os.path.exists(lib):
            return False
        t = os.path.getmtime(lib)
        if not all(os.path.getmtime(os.path.join(HERE, d)) <= t for d in DEPS + ["build.py"]):
            return False
    return True


def build(force: bool = False, verbose: bool = False, extra=()) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    if not force and not extra and up_to_date():
        return OUT
    # one hipcc per translation unit, in parallel, then link
    obj_dir = os.path.join(LIB_DIR, "obj" + os.environ.get("DOCKAUV_OBJ_TAG", ""))
    os.makedirs(obj_dir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"]
    procs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        cmd = [HIPCC, *cflags, *PER_SOURCE_FLAGS.get(src, []), *extra, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd, cwd=HERE, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    fault = None
    if os.environ.get("DOCKAUV_LIB_NAME") is None and not extra:   # the product build also makes the fault-injection test library
        fobj = os.path.join(obj_dir, "dockauv_kernels_f32_faultinject.o")
        cmd = [HIPCC, *cflags, *FAULT_FLAGS, "-c", FAULT_SRC, "-o", fobj]
        if verbose:
            print(" ".join(cmd), flush=True)
        fault = (fobj, subprocess.Popen(cmd, cwd=HERE, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    objs = []
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError(f"hipcc failed on {src}")
        if verbose and out:
            print(out)
        objs.append(obj)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT], cwd=HERE,
                       capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed linking libdockauv.so")
    if fault is not None:
        fobj, p = fault
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError("hipcc failed on the fault-injection build of " + FAULT_SRC)
        fobjs = [fobj if os.path.basename(o) == FAULT_SRC.replace(".hip", ".o") else o for o in objs]
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *fobjs, "-o", FAULT_LIB], cwd=HERE,
                           capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("hipcc failed linking libdockauv_faultinject.so")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True,
                extra=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else ()))
