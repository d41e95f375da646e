#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: HBM-side bytes per step_kernel launch.

usage: summarize_pmc.py <prof dir with pmc_fetch/ pmc_write/> [<calibration dir with calib_fetch/ calib_write/ calib.txt>]

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): the counters come in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of a 16-B-per-lane streaming read and other access widths are uncalibrated, so the factor for the
step kernel's shape (one dword per lane, many row streams) is MEASURED with scripts/micro/fetch_calib.hip, a copy of
known size in the same shape, and applied here.  Prints a JSON object on the last line."""
import csv
import glob
import json
import os
import re
import sys


def avg_counter(root, subdir, name, kernel):
    vals = []
    for f in glob.glob(os.path.join(root, subdir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel in row.get("Kernel_Name", "") and row.get("Counter_Name") == name:
                vals.append(float(row["Counter_Value"]))
    if len(vals) > 8:
        vals = vals[len(vals) // 4:]          # drop the warm-up launches (cold caches)
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


out = sys.argv[1]
calib = sys.argv[2] if len(sys.argv) > 2 else None
f_fetch = f_write = None
if calib and os.path.exists(os.path.join(calib, "calib.txt")):
    m = re.search(r"read_bytes_per_launch=(\d+) write_bytes_per_launch=(\d+)", open(os.path.join(calib, "calib.txt")).read())
    rb, wb = int(m.group(1)), int(m.group(2))
    cf, _ = avg_counter(calib, "calib_fetch", "FETCH_SIZE", "soa_copy")
    cw, _ = avg_counter(calib, "calib_write", "WRITE_SIZE", "soa_copy")
    if cf and cw:
        f_fetch, f_write = cf * 1024 / rb, cw * 1024 / wb
        print(f"calibration (dword per lane, 32 row streams, {rb} B read + {wb} B written per launch): "
              f"FETCH_SIZE reports {f_fetch:.3f} x the bytes read, WRITE_SIZE {f_write:.3f} x the bytes written")
f, nf = avg_counter(out, "pmc_fetch", "FETCH_SIZE", "step_kernel")
w, nw = avg_counter(out, "pmc_write", "WRITE_SIZE", "step_kernel")
print(f"FETCH_SIZE avg per step_kernel dispatch: {f} KiB over {nf} dispatches (raw)")
print(f"WRITE_SIZE avg per step_kernel dispatch: {w} KiB over {nw} dispatches (raw)")
res = {}
if f is not None and w is not None:
    ff = f_fetch or 1.0
    fw = f_write or 1.0
    rd, wr = f * 1024 / ff, w * 1024 / fw
    print(f"corrected: read {rd:.0f} B + written {wr:.0f} B = {rd + wr:.0f} B per launch")
    res = {"read_bytes": rd, "write_bytes": wr, "traffic_bytes": rd + wr, "fetch_factor": ff, "write_factor": fw}
print(json.dumps(res))
