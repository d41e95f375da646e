"""`python bench.py --gpus 2` end to end on ONE GPU (runs last in the suite: the file name sorts behind the parity tests).

The driver starts the scaling runs as `python bench.py --gpus N`; the launcher half of that is covered on the CPU
(tests/test_bench_contract.py).  Here both self-launched rank processes get as far as the JSON line: they share the box's
one GPU (DOCKAUV_RANKS_SHARE_GPU=1) and use gloo for the all-gather, because RCCL refuses two ranks on one device -- everything
else (sharding, packed rows written into the rank's slice of the gather buffer, the overlapped gather, max over ranks, the
sub-measurements of the N > 1 line) is the code the N-GPU run executes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_self_launched_ranks_print_one_line():
    env = dict(os.environ, DOCKAUV_RANKS_SHARE_GPU="1", DOCKAUV_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, out.stdout[-2000:]          # the launcher hands on the JSON line alone
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
    c = d["config"]
    assert c["world_size"] == 2 and c["backend"] == "gloo" and len(c["ranks"]) == 2
    assert c["workload"].startswith("config4") and c["envs_per_gpu"] == 32768 and c["total_envs"] == 65536
    assert c["obs_finite"] and c["gather_bytes_per_rank_per_step"] == 32768 * (c["n_obs"] + 2) * 4
    assert d["value"] > 0 and abs(d["value"] - 65536 * 20 / (d["ms_per_step"] * 1e-3 * 20)) <= 1e-6 * d["value"]
    # the line explains itself: the same shards without the gather, and what the gather asked of the links
    assert d["same_workload_without_gather"]["per_gpu_value"] > 0 and d["link_GBps_achieved"]["per_link_direction"] > 0
    for r in (0, 1):
        assert f"[rank {r}] rendezvous of 2 ranks complete" in out.stderr
