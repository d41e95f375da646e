#!/bin/bash
# HBM-side traffic per STEP of the resident sequence kernel (step_seq_kernel, 64 steps per dispatch) next to the per-launch
# kernel's, config 3 at 1 048 576 and 65 536 envs: separate --pmc passes, calibration factors of profiles/r4/calib
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r4/resident_traffic; rm -rf $OUT; mkdir -p $OUT
for spec in "3 1048576" "3 65536" "2 1048576"; do set -- $spec
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/c$1_$2_$c -- python3 bench.py --config $1 --envs $2 --no-cpu --no-sweep --no-configs --no-closed-loop --steps 128 --warmup 64 --min-seconds 0.001 --max-reps 2 > /dev/null 2> $OUT/err.txt || tail -3 $OUT/err.txt
done
python3 - $OUT $1 $2 <<'PY'
import csv, glob, os, sys
out, cid, envs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ALGO = {2: 356, 3: 420, 4: 460, 5: 482}
def avg(counter, kernel, per):
    v = []
    for f in glob.glob(os.path.join(out, f"c{cid}_{envs}_{counter}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and kernel in row["Kernel_Name"]:
                v.append(float(row["Counter_Value"]))
    v = v[len(v) // 4:] if len(v) > 4 else v
    return sum(v) / len(v) * 1024 / per if v else float("nan")
# calibration of this round (profiles/r4/calib): FETCH_SIZE reports 0.5 x the bytes read in this access shape, WRITE_SIZE 1.0 x
for name, kern, per in (("per launch (step_kernel)", "step_kernel<", 1), ("resident (step_seq_kernel, per step of 64)", "step_seq_kernel", 64)):
    rd, wr = avg("FETCH_SIZE", kern, per) / 0.5, avg("WRITE_SIZE", kern, per)
    alg = ALGO[cid] * envs
    print(f"config{cid} x {envs}: {name:44s} read {rd / 1e6:8.1f} MB + written {wr / 1e6:8.1f} MB = {(rd + wr) / 1e6:8.1f} MB per step = {(rd + wr) / alg:5.2f} x algorithmic ({alg / 1e6:.1f} MB)")
PY
done | tee gpurun_out/r4/resident_traffic.txt
rm -rf $OUT
