#!/bin/bash
# Round-4 profiles (produced on the GPU box; copy gpurun_out/prof_r4 -> profiles/r4).  For each BASELINE config:
#   kernel_stats.csv   rocprofv3 --kernel-trace --stats of bench.py (average duration of dockauv::step_kernel)
#   pmc_summary.txt    FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs), corrected with the calibration copy kernel
#   sq_summary.txt     SQ instruction counters (own --pmc pass)
# and pmc_counters.json, which bench.py reads for roofline.traffic / valu_frac (entries carry their source + kernel sha).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT/gpurun_out/prof_r4
rm -rf $ROOT; mkdir -p $ROOT/calib
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/micro/fetch_calib.hip -o $ROOT/calib/fetch_calib 2> /dev/null
$ROOT/calib/fetch_calib 1048576 > $ROOT/calib/calib.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/calib/calib_fetch -- $ROOT/calib/fetch_calib 1048576 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/calib/calib_write -- $ROOT/calib/fetch_calib 1048576 > /dev/null 2>&1
cat $ROOT/calib/calib.txt
prof() {   # name, bench args...
  local name=$1; shift
  local OUT=$ROOT/$name
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu --no-sweep --no-configs --no-closed-loop --no-resident --steps 1000 --warmup 200 --min-seconds 0.05 "$@" > $OUT/bench_line.json 2> $OUT/bench_trace.err || { tail -5 $OUT/bench_trace.err; return 1; }
  find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
  # the same trace per dispatch: launches queued back to back (the timed regions) against launches after an idle gap
  # (the per-dispatch event timing loop) -- rocprofv3's average mixes the two
  python3 scripts/diag/trace_split.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > $OUT/kernel_trace_split.txt 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu --no-sweep --no-configs --no-closed-loop --no-resident --steps 100 --warmup 20 --min-seconds 0.001 "$@" > /dev/null 2> $OUT/pmc_fetch.err || tail -3 $OUT/pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu --no-sweep --no-configs --no-closed-loop --no-resident --steps 100 --warmup 20 --min-seconds 0.001 "$@" > /dev/null 2> $OUT/pmc_write.err || tail -3 $OUT/pmc_write.err
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py --no-cpu --no-sweep --no-configs --no-closed-loop --no-resident --steps 100 --warmup 20 --min-seconds 0.001 "$@" > /dev/null 2> $OUT/pmc_sq.err || tail -3 $OUT/pmc_sq.err
  python3 scripts/summarize_pmc.py $OUT $ROOT/calib > $OUT/pmc_summary.txt 2>&1
  python3 - $OUT <<'PY' > $OUT/sq_summary.txt
import csv, glob, os, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "step_kernel" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k in sorted(acc):
    v = acc[k][len(acc[k]) // 4:]
    res[k] = sum(v) / len(v)
    print(f"{k:24s} avg/dispatch {res[k]:16.1f}  (n={len(v)})")
print(json.dumps(res))
PY
  echo "== $name"; head -3 $OUT/kernel_stats.csv; tail -2 $OUT/pmc_summary.txt | head -1; tail -1 $OUT/sq_summary.txt
  rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
}
prof config2 --config 2
prof config3 --config 3
prof config4 --config 4
prof config5 --config 5
prof config5_sorted --config 5 --layout vehicle_sorted
prof config3_dense --only-ray-dense 3
prof config4_dense --only-ray-dense 4
prof config2_1M --config 2 --envs 1048576 --steps 200 --warmup 20
prof config3_1M --config 3 --envs 1048576 --steps 200 --warmup 20
prof config4_1M --config 4 --envs 1048576 --steps 100 --warmup 20
prof config5_1M --config 5 --envs 1048576 --steps 100 --warmup 20
# BASELINE's totals of configs 4 / 5 (what their eight GPUs share) on ONE GPU: one-wave groups, capsule records in registers
prof config4_262k --config 4 --envs 262144 --steps 200 --warmup 20
prof config5_524k --config 5 --envs 524288 --steps 200 --warmup 20
# the resident step sequence (dockauv_step_sequence's fast path): rocprofv3 kernel stats of step_seq_kernel (one dispatch = up to
# 64 steps) next to the bench line's own per-step figure
for c in 2 3 4 5; do
  OUT=$ROOT/resident_config$c; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --config $c --no-cpu --no-sweep --no-configs --no-closed-loop --steps 1024 --warmup 128 --min-seconds 0.05 > $OUT/bench_line.json 2> $OUT/bench_trace.err || tail -5 $OUT/bench_trace.err
  find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
  rm -rf $OUT/trace
  python3 - $OUT <<'PY'
import csv, json, sys, os
out = sys.argv[1]
d = json.loads(open(os.path.join(out, "bench_line.json")).read().strip().splitlines()[-1])
r = d.get("sequence_resident") or {}
rows = [x for x in csv.DictReader(open(os.path.join(out, "kernel_stats.csv"))) if "step_seq_kernel" in x["Name"]]
for x in rows:
    print("step_seq_kernel: calls %s, average %.2f us per dispatch of 64 steps = %.3f us per step (rocprofv3); bench line: %.3f us per step (HIP events), per-launch kernel %.3f us" % (
        x["Calls"], float(x["AverageNs"]) / 1e3, float(x["AverageNs"]) / 1e3 / 64, r.get("us_per_step_events", float("nan")), d["roofline"]["kernel_us"]))
PY
done | tee $ROOT/resident_summary.txt
python3 - $ROOT <<'PY'
import json, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
root = sys.argv[1]
sha = bench.kernel_source_sha()
out = {}
for name, key in (("config2", "config2_envs4096"), ("config3", "config3_envs65536"), ("config4", "config4_envs32768"),
                  ("config5", "config5_envs65536"), ("config5_sorted", "config5_sorted_envs65536"), ("config3_dense", "config3_dense_envs65536"),
                  ("config4_dense", "config4_dense_envs32768"), ("config2_1M", "config2_envs1048576"), ("config3_1M", "config3_envs1048576"),
                  ("config4_1M", "config4_envs1048576"), ("config5_1M", "config5_envs1048576"),
                  ("config4_262k", "config4_envs262144"), ("config5_524k", "config5_envs524288")):
    e = {"source": f"profiles/r4/{name}/", "kernel_sha": sha}
    try:
        d = json.loads(open(os.path.join(root, name, "pmc_summary.txt")).read().strip().splitlines()[-1])
        e.update(traffic_bytes=d["traffic_bytes"], read_bytes=d["read_bytes"], write_bytes=d["write_bytes"])
    except Exception as ex:
        print("no traffic for", name, ex)
    try:
        d = json.loads(open(os.path.join(root, name, "sq_summary.txt")).read().strip().splitlines()[-1])
        e.update(sq_insts_valu=d.get("SQ_INSTS_VALU"), sq_insts_salu=d.get("SQ_INSTS_SALU"), sq_waves=d.get("SQ_WAVES"))
    except Exception as ex:
        print("no SQ counters for", name, ex)
    out[key] = e
json.dump(out, open(os.path.join(root, "pmc_counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
