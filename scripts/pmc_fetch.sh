#!/bin/bash
# instruction / scalar cache counters of the step kernel.  Usage: scripts/pmc_fetch.sh <tag> [bench args...]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/ic_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -- python3 bench.py --no-cpu --no-sweep --steps 50 --warmup 10 "$@" > /dev/null 2> $OUT/p1.err || tail -3 $OUT/p1.err
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_STALL --output-format csv -d $OUT/p2 -- python3 bench.py --no-cpu --no-sweep --steps 50 --warmup 10 "$@" > /dev/null 2> $OUT/p2.err || tail -3 $OUT/p2.err
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "step_kernel" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:24s} avg/dispatch {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
