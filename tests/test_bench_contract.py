"""bench.py's contract with the driver: the workloads are BASELINE.json's configs, the algorithmic bytes per env-step are
SURVEY.md section 8d's formula, and (on a GPU) the default invocation prints ONE JSON line with the fields the round
prompt names."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def algorithmic_bytes(n_u, n_cap, n_sph, n_obs):
    # read state, filtered u, action, goal(3)+heading... = (12 + n_u + n_u + 3 + 4 + 1) + obstacles;
    # written state, u, t_steps, obs, reward, done-as-word
    return 4 * ((12 + n_u + n_u + 3 + 4 + 1) + 7 * n_cap + 4 * n_sph) + 4 * (12 + n_u + 1 + n_obs + 1 + 1)


def test_workloads_are_baselines_configs():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    cfgs = base["configs"]
    w2, w3, w4, w5 = (bench.workload(c, 0) for c in (2, 3, 4, 5))
    assert "4 096 envs" in cfgs[1] and w2["envs"] == 4096 and w2["scenario"] == "SimpleDocking3d"
    assert "65 536 envs" in cfgs[2] and "16-beam" in cfgs[2] and "8 sphere" in cfgs[2]
    assert w3["envs"] == 65536 and w3["scenario"] == "SphereDocking3d"
    # 16 beams: 4 x 4 rays from the fan's half-angles and resolution (objects/sensor.py:43-87)
    r = w3["cfg"]["radar"]
    n_side = int(round(r["alpha"] / r["ray_per_deg"])) + 1
    assert n_side * n_side == 16
    assert "262 144 envs sharded over 8" in cfgs[3] and w4["envs"] * 8 == 262144 and w4["cfg"]["vehicle"] == "LAUV"
    assert "524 288 envs on 8" in cfgs[4] and w5["envs"] * 8 == 524288
    assert sorted(set(w5["vehicles"])) == ["BlueROV2", "LAUV"] and w5["vehicles"][:2] == ["BlueROV2", "LAUV"]
    assert abs(w5["vehicles"].count("LAUV") / len(w5["vehicles"]) - 0.5) < 1e-9


def test_algorithmic_bytes_follow_the_survey_formula():
    assert bench.ALGO_BYTES[2] == algorithmic_bytes(6, 0, 0, 16 + 20) == 356          # default 7 x 9 fan -> 4 x 5 cells, all at max_dist
    assert bench.ALGO_BYTES[3] == algorithmic_bytes(6, 0, 8, 16 + 4) == 420            # 16 rays -> 2 x 2 cells
    assert bench.ALGO_BYTES[4] == algorithmic_bytes(3, 5, 0, 16 + 20) == 460           # LAUV, 63 rays -> 4 x 5 cells
    # config 5: half BlueROV2 (V_c written back as well), half LAUV (SURVEY.md 8d's two figures)
    assert bench.ALGO_BYTES[5] == (500 + 464) // 2


def test_roofline_entry_shape():
    r = bench.roofline_of(3, 65536, 10.0, 100, "deadbeef0000")
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["limited_by"] == "latency"
    assert bench.roofline_of(2, 4096, 4.5, 10, "x")["limited_by"] == "launch"
    assert bench.roofline_of(4, 32768, 20.0, 10, "x", dense=True)["limited_by"] == "valu"
    assert np.isclose(r["achieved"], 420 * 65536 / 10e-6 / 1e9) and np.isclose(r["frac"], r["achieved"] / 8000.0)
    assert "traffic" in r
    if r["traffic"] is not None:   # committed counters: they must say where they come from, and that this is another kernel
        assert "profiles/" in r["traffic_source"] and "later kernel version" in r["traffic_source"]
        assert 0 < r["valu_frac"] < r["valu_frac_at_4_cycles_per_inst"] < 4
    assert "not derived from this run" in r["limited_by_source"]


def test_committed_counters_name_their_kernel():
    """profiles/pmc_counters.json is taken by scripts/profile_r4.sh at the end of a round; every entry names the kernel-source
    hash it was counted on, and the bench line says so whenever the library is a later kernel version (never silently)."""
    import warnings
    path = os.path.join(ROOT, "profiles", "pmc_counters.json")
    if not os.path.exists(path):
        pytest.skip("no committed counters")
    d = json.load(open(path))
    sha = bench.kernel_source_sha()
    sizes = {2: 4096, 3: 65536, 4: 32768, 5: 65536}
    for cid, envs in sizes.items():
        key = f"config{cid}_envs{envs}"
        assert key in d and d[key]["traffic_bytes"] > 0 and len(d[key]["kernel_sha"]) == 12
        r = bench.roofline_of(cid, envs, 10.0, 100, sha)
        if d[key]["kernel_sha"] != sha:
            warnings.warn(f"{key}: counters of kernel {d[key]['kernel_sha']}, tree is {sha}: re-run scripts/profile_r4.sh")
            assert "later kernel version" in r["traffic_source"] and "valu_frac_note" in r
        else:
            assert "later kernel version" not in r["traffic_source"]


def test_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (the shape of the driver's N = 1 command) starts two rank processes
    itself: both make the rendezvous (gloo here: no device needed for it) and get as far as the first thing that needs a
    GPU; in this container that is where they stop, and the launcher hands the failure on."""
    env = dict(os.environ, DOCKAUV_DIST_BACKEND="gloo", HIP_VISIBLE_DEVICES="")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    err = out.stderr
    assert "launch N > 1 with" not in err
    for r in (0, 1):
        assert f"[rank {r}] rendezvous of 2 ranks complete" in err, err[-2000:]
    import torch
    if not torch.cuda.is_available():
        assert out.returncode != 0 and "needs an MI355X" in err and out.stdout.strip() == ""
        assert "stopping the other ranks" in err


def test_launcher_relays_exactly_the_json_line():
    """rank 0's stdout may carry a collective library's banner in front of the line (gloo prints one): the launcher hands
    on the JSON line alone."""
    line, others = bench.split_json_line('[Gloo] Rank 0 is connected to 1 peer ranks.\n{"metric": "m", "value": 1.0}\n')
    assert json.loads(line) == {"metric": "m", "value": 1.0} and others == ["[Gloo] Rank 0 is connected to 1 peer ranks."]
    assert bench.split_json_line("no line here\n") == (None, [])
    line, others = bench.split_json_line('{"a": 1}\nnoise\n{"b": 2}\n')
    assert json.loads(line) == {"b": 2} and others == ['{"a": 1}', "noise"]


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", HIP_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "WORLD_SIZE = 2" in out.stderr


@pytest.mark.gpu
def test_default_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "50", "--min-seconds", "0.05",
                          "--no-sweep", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "reps", "configs"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 200 and d["warmup"] == 50 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["config"]["workload"].startswith("config3") and d["config"]["envs_per_gpu"] == 65536
    assert np.isclose(d["value"], 65536 * 200 / (d["ms_per_step"] * 1e-3 * 200), rtol=1e-9)
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and np.isclose(r["frac"], r["achieved"] / r["peak"]) and 0.05 < r["frac"] < 1.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and "SphereDocking3d" in c["sample"]
    dense = [s for s in d["configs"] if "ray-dense" in s["workload"]]
    assert sorted(s["workload"][:7] for s in dense) == ["config3", "config4"]
    for s in dense:   # the ray stage where it is busy: >= 90 % of the fans have a hit, before and after the timed region
        a = s["active_fraction"]
        assert min(a["region_start"], a["region_end_min"], a["after_kernel_timing"]) >= 0.9, a
        assert s["kernel_us"] > 0 and s["obs_finite"]
    assert sorted(s.get("layout") for s in d["configs"] if s["workload"].startswith("config5")) == ["interleaved", "vehicle_sorted"]
    subs = {s["workload"][:7]: s for s in d["configs"] if "ray-dense" not in s["workload"] and s.get("layout") != "vehicle_sorted"}
    assert set(subs) == {"config2", "config3", "config4", "config5"}
    for k in ("config2", "config4", "config5"):   # the CPU path on the same scenario, per config
        assert subs[k]["cpu_baseline"]["kind"] == "port" and subs[k]["cpu_baseline"]["value"] > 0
    assert subs["config2"]["roofline"]["limited_by"] == "launch" and subs["config3"]["roofline"]["limited_by"] == "latency"
    for s in subs.values():
        assert s["kernel_us"] > 0 and 0 < s["roofline"]["frac"] < 1 and "traffic" in s["roofline"]
    assert subs["config2"]["envs"] == 4096 and subs["config4"]["envs"] == 32768 and subs["config5"]["envs"] == 65536
