#!/bin/bash
# round 2, first GPU call: micro-benchmark of issue costs, the GPU test-suite, in-kernel time lines, the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/micro/issue_rate.hip -o /tmp/issue_rate && timeout -k 10 120 /tmp/issue_rate > gpurun_out/issue_rate.txt 2>&1
echo "== micro done"; cat gpurun_out/issue_rate.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_p2p.py > gpurun_out/r2c1_tests.log 2>&1; echo "== tests rc=$?"; tail -15 gpurun_out/r2c1_tests.log
for c in "2 4096" "3 65536" "4 32768" "5 65536"; do set -- $c
  DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/libdockauv_stamps.so timeout -k 10 120 python scripts/stamps.py --config $1 --envs $2 >> gpurun_out/r2c1_stamps.txt 2>&1
done; echo "== stamps"; cat gpurun_out/r2c1_stamps.txt
timeout -k 10 600 python bench.py > gpurun_out/r2c1_bench.json 2> gpurun_out/r2c1_bench.err; echo "== bench rc=$?"; tail -3 gpurun_out/r2c1_bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2c1_bench.json").read().strip().splitlines()[-1])
print("headline", d["config"]["workload"], "value %.3e"%d["value"], "us/step %.2f"%(d["ms_per_step"]*1e3), "reps", d["reps"], "kernel_us %.2f"%d["roofline"]["kernel_us"], "frac %.4f"%d["roofline"]["frac"])
for c in d.get("configs", []):
    cl=c["closed_loop"] or {}
    print(c["workload"][:40], "N",c["envs"], "value %.3e"%c["value"], "us/step %.2f"%(c["ms_per_step"]*1e3), "kernel_us %.2f"%c["kernel_us"], "frac %.4f"%c["roofline"]["frac"], "closed py %.1f graph %s"%(cl.get("python_issued_us_per_step",0), cl.get("hip_graph_us_per_step")))
print("cpu", d.get("cpu_baseline"))
print("sweep", d.get("sweep"))
PY
