#!/bin/bash
# SQ counters of the step kernel (own pass, --pmc only).  Usage: scripts/pmc_sq.sh <tag> [bench args...]
# (--no-closed-loop --no-resident: the counter passes see single launches only -- no HIP-graph replay, no 64-step dispatches --
# as scripts/profile_r4.sh has always had it; the one unexplained fault of round 4 came in a pass of this script without them)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/p1 -- python3 bench.py --no-cpu --no-sweep --no-closed-loop --no-resident --steps 50 --warmup 10 "$@" > /dev/null 2> $OUT/p1.err || tail -3 $OUT/p1.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p2 -- python3 bench.py --no-cpu --no-sweep --no-closed-loop --no-resident --steps 50 --warmup 10 "$@" > /dev/null 2> $OUT/p2.err || tail -3 $OUT/p2.err
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
w = acc.get("SQ_WAVES")
if w:
    nw = sum(w) / len(w)
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
        if k in acc:
            print(f"per wave {k:22s} {sum(acc[k])/len(acc[k])/nw:12.1f}")
PY
