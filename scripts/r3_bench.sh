#!/bin/bash
# the default bench line (no sweep unless SWEEP=1) + a table of its sub-results -> gpurun_out/r3/bench_line.json
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
EXTRA="--no-sweep"; [ "$SWEEP" = "1" ] && EXTRA=""
timeout -k 10 900 python bench.py $EXTRA --cpu-seconds ${CPU_SECONDS:-4} "$@" > gpurun_out/r3/bench_line.json 2> gpurun_out/r3/bench_err.txt || { tail -20 gpurun_out/r3/bench_err.txt; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3/bench_line.json").read().strip().splitlines()[-1])
print("headline value %.3e  us/step %.2f  kernel_us %.2f  frac %.3f  cpu %.0f" % (d["value"], d["ms_per_step"] * 1e3, d["roofline"]["kernel_us"], d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("value", 0)))
for s in d.get("configs", []):
    a = s.get("active_fraction")
    print("%-72s %-14s kernel_us=%6.2f frac=%.3f us_step=%6.2f %-8s %s cpu=%s" % (s["workload"][:72], s.get("layout", ""), s["kernel_us"], s["roofline"]["frac"], s["ms_per_step"] * 1e3,
          s["roofline"]["limited_by"], ("active %.2f/%.2f/%.2f" % (a["region_start"], a["region_end_min"], a["after_kernel_timing"])) if a else "", (s.get("cpu_baseline") or {}).get("value")))
for s in d.get("sweep", []):
    print("sweep", s)
PY
