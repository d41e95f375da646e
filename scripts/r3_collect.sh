#!/bin/bash
# copy the closing run's results from gpurun_out/ into profiles/ (run in the build container after scripts/r3_final.sh)
cd "$(dirname "$0")/.."
for c in config2 config3 config4 config5 config5_sorted config3_dense config4_dense config2_1M config3_1M; do
  mkdir -p profiles/r3/$c
  cp gpurun_out/prof_r3/$c/{kernel_stats.csv,kernel_trace_split.txt,bench_line.json,pmc_summary.txt,sq_summary.txt} profiles/r3/$c/
done
mkdir -p profiles/r3/calib; cp gpurun_out/prof_r3/calib/calib.txt profiles/r3/calib/
cp gpurun_out/prof_r3/pmc_counters.json profiles/r3/pmc_counters.json
cp gpurun_out/prof_r3/pmc_counters.json profiles/pmc_counters.json
cp gpurun_out/r3/bench_default_unprofiled.json profiles/r3/bench_default_unprofiled.json
python -m pytest tests/test_bench_contract.py -q -m "not gpu" 2>&1 | tail -1
