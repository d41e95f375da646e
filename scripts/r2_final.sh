#!/bin/bash
# Round-2 closing run on the GPU box: smoke, the whole GPU suite (p2p rehearsal included), the micro-benchmarks, the
# in-kernel time lines, the share of envs that take a ray pass, and the default bench line.  Outputs -> gpurun_out/r2_final/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_final; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "== smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "== tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/micro/issue_rate.hip -o /tmp/issue_rate && timeout -k 10 120 /tmp/issue_rate > $O/issue_rate.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/micro/wave_census.hip -o /tmp/wave_census && timeout -k 10 60 /tmp/wave_census > $O/wave_census.txt 2>&1
for c in "2 4096" "3 65536" "4 32768"; do set -- $c
  DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/libdockauv_stamps.so timeout -k 10 120 python scripts/stamps.py --config $1 --envs $2 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|nanmedian" >> $O/stamps.txt
done
timeout -k 10 300 python scripts/active_fraction.py 2>&1 | grep -v amdgpu.ids > $O/active_fraction.txt
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "== bench rc=$?"; tail -2 $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_final/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("headline", d["config"]["workload"], "value %.3e" % d["value"], "us/step %.2f" % (d["ms_per_step"] * 1e3), "reps", d.get("reps"),
      "kernel_us %.2f" % r["kernel_us"], "frac %.4f" % r["frac"], "traffic", r.get("traffic"), r.get("traffic_source"), "valu", r.get("valu_frac"))
for c in d.get("configs", []):
    cl = c.get("closed_loop") or {}
    print(c["workload"][:44], "N", c["envs"], "value %.3e" % c["value"], "kernel_us %.2f" % c["kernel_us"], "frac %.4f" % c["roofline"]["frac"],
          "traffic", c["roofline"].get("traffic"), "closed py %.1f graph %s" % (cl.get("python_issued_us_per_step", 0), cl.get("hip_graph_us_per_step")))
print("cpu", d.get("cpu_baseline")); print("sweep", d.get("sweep"))
PY
