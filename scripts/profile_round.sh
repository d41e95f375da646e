#!/bin/bash
# Everything the round's profiles/ directory holds, produced on the GPU box.  Usage: scripts/profile_round.sh <tag>
# -> gpurun_out/prof_<tag>/{config2,config2_1M,config3}/..., calib/, pmc_traffic.json
TAG=${1:-r1}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $ROOT/calib
# --- calibration of the byte counters for the kernel's access shape
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/micro/fetch_calib.hip -o $ROOT/calib/fetch_calib 2> /dev/null
$ROOT/calib/fetch_calib 1048576 > $ROOT/calib/calib.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/calib/calib_fetch -- $ROOT/calib/fetch_calib 1048576 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/calib/calib_write -- $ROOT/calib/fetch_calib 1048576 > /dev/null 2>&1
cat $ROOT/calib/calib.txt
prof() {   # name, bench args...
  local name=$1; shift
  local OUT=$ROOT/$name
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu --no-sweep "$@" > $OUT/bench_line.json 2> $OUT/bench_trace.err || { tail -5 $OUT/bench_trace.err; return 1; }
  find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu --no-sweep --steps 100 --warmup 20 "$@" > /dev/null 2> $OUT/pmc_fetch.err || tail -3 $OUT/pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu --no-sweep --steps 100 --warmup 20 "$@" > /dev/null 2> $OUT/pmc_write.err || tail -3 $OUT/pmc_write.err
  python3 scripts/summarize_pmc.py $OUT $ROOT/calib > $OUT/pmc_summary.txt 2>&1
  echo "== $name"; head -3 $OUT/kernel_stats.csv; cat $OUT/pmc_summary.txt
}
prof config2 --config 2
prof config2_1M --config 2 --envs 1048576 --steps 200 --warmup 20
prof config3 --config 3 --steps 300 --warmup 30
prof config4 --config 4 --steps 300 --warmup 30
# --- the multi-GPU step with ONE rank (no fabric): step kernels with riding copy groups (dockauv::step_ride_kernel)
(
  export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 DOCKAUV_FORCE_DIST=1
  OUT=$ROOT/p2p_one_rank; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu --no-sweep --gather p2p > $OUT/bench_line.json 2> $OUT/bench_trace.err || tail -5 $OUT/bench_trace.err
  find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
  echo "== p2p_one_rank"; head -6 $OUT/kernel_stats.csv
)
python3 - $ROOT <<'PY'
import json, os, sys
root = sys.argv[1]
out = {}
for name, key in (("config2", "config2_envs4096"), ("config2_1M", "config2_envs1048576"), ("config3", "config3_envs65536"), ("config4", "config4_envs32768")):
    try:
        last = open(os.path.join(root, name, "pmc_summary.txt")).read().strip().splitlines()[-1]
        d = json.loads(last)
        if d:
            out[key] = d["traffic_bytes"]
    except Exception as e:
        print("no traffic for", name, e)
json.dump(out, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
print(out)
PY
# drop the bulky raw traces, keep the counter CSVs small enough to merge back
find $ROOT -name "*kernel_trace.csv" -size +2M -delete
