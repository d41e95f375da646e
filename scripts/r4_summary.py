#!/usr/bin/env python3
"""One table out of a profile directory of scripts/profile_r4.sh: python scripts/r4_summary.py gpurun_out/prof_r4 > profiles/r4/summary_table.txt"""
import csv
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

root = sys.argv[1]
print("round 4 profile summary (scripts/profile_r4.sh, kernel source %s; per-launch step_kernel; events us = HIP events around the timed regions / K of the "
      "traced bench line, rocprof us = rocprofv3 --kernel-trace --stats average; frac = algorithmic bytes / events us / 8 TB/s; traffic = "
      "calibrated FETCH_SIZE + WRITE_SIZE per launch; VALU busy = SQ_INSTS_VALU x 2.1 cycles / (1 024 SIMDs x kernel cycles at 2.4 GHz); "
      "waves/CU = SQ_WAVE_CYCLES x 4 / kernel cycles / 256; VALU/wave = SQ_INSTS_VALU / SQ_WAVES)" % bench.kernel_source_sha())
print("%-16s %8s %9s %10s %6s %10s %6s %7s %9s %9s %12s %8s %9s" % ("workload", "envs", "events us", "rocprof us", "frac", "traffic MB", "x alg",
                                                                    "VALU M", "SALU/VALU", "VALU busy", "wait/wavecyc", "waves/CU", "VALU/wave"))
names = ["config2", "config3", "config4", "config5", "config5_sorted", "config3_dense", "config4_dense", "config2_1M", "config3_1M",
         "config4_1M", "config5_1M", "config4_262k", "config5_524k"]
for name in names:
    d = os.path.join(root, name)
    try:
        line = json.loads(open(os.path.join(d, "bench_line.json")).read().strip().splitlines()[-1])
    except Exception:
        continue
    r = line.get("roofline", line)
    us, envs = r["kernel_us"], r["envs_per_launch"]
    cid = int(name[6])
    alg = bench.ALGO_BYTES[cid] * envs
    roc = float("nan")
    try:
        for x in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
            if "step_kernel" in x["Name"]:
                roc = float(x["AverageNs"]) / 1e3
                break
    except Exception:
        pass
    try:
        t = json.loads(open(os.path.join(d, "pmc_summary.txt")).read().strip().splitlines()[-1])["traffic_bytes"]
    except Exception:
        t = float("nan")
    try:
        q = json.loads(open(os.path.join(d, "sq_summary.txt")).read().strip().splitlines()[-1])
    except Exception:
        q = {}
    cyc = us * 2400.0
    valu = q.get("SQ_INSTS_VALU", float("nan"))
    print("%-16s %8d %9.2f %10.2f %6.3f %10.1f %6.2f %7.2f %9.2f %9.2f %12.2f %8.1f %9.0f" % (
        name, envs, us, roc, alg / (us * 1e-6) / 8e12, t / 1e6, t / alg, valu / 1e6, q.get("SQ_INSTS_SALU", float("nan")) / valu,
        valu * 2.1 / (1024 * cyc), q.get("SQ_WAIT_ANY", float("nan")) / q.get("SQ_WAVE_CYCLES", float("nan")),
        q.get("SQ_WAVE_CYCLES", float("nan")) * 4 / cyc / 256, valu / q.get("SQ_WAVES", float("nan"))))
