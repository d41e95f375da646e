#!/bin/bash
# closing run of round 3 on the GPU box: profiles of the current kernels, the un-profiled default bench line, smoke
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
bash scripts/profile_r3.sh > gpurun_out/prof_r3.log 2>&1 || { tail -20 gpurun_out/prof_r3.log; exit 1; }
grep -E "^== |step_kernel" gpurun_out/prof_r3.log | cut -c1-150
cp gpurun_out/prof_r3/pmc_counters.json profiles/pmc_counters.json   # (the box's copy: the default line below cites the counters just taken)
timeout -k 10 600 python bench.py > gpurun_out/r3/bench_default_unprofiled.json 2> gpurun_out/r3/bench_default.err || { tail -5 gpurun_out/r3/bench_default.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3/bench_default_unprofiled.json").read().strip().splitlines()[-1])
print("headline value %.3e us/step %.2f kernel_us %.2f frac %.3f" % (d["value"], d["ms_per_step"] * 1e3, d["roofline"]["kernel_us"], d["roofline"]["frac"]))
for s in d["configs"]:
    print("%-70s %-14s kernel_us=%6.2f frac=%.3f us_step=%6.2f" % (s["workload"][:70], s.get("layout", ""), s["kernel_us"], s["roofline"]["frac"], s["ms_per_step"] * 1e3))
PY
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()"
