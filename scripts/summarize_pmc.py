#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py: average per step_kernel dispatch.
gfx950 corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; FETCH_SIZE reports half of the bytes of
a wide coalesced read stream (calibrate before trusting an absolute); WRITE_SIZE is exact for streaming stores."""
import csv
import glob
import os
import sys

out = sys.argv[1]


def avg_counter(subdir, name):
    files = glob.glob(os.path.join(out, subdir, "**", "*counter_collection.csv"), recursive=True)
    vals = []
    for f in files:
        for row in csv.DictReader(open(f)):
            if "step_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == name:
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


f, nf = avg_counter("pmc_fetch", "FETCH_SIZE")
w, nw = avg_counter("pmc_write", "WRITE_SIZE")
print(f"FETCH_SIZE avg per step_kernel dispatch: {f} KiB over {nf} dispatches (raw; x2 if the stream is wide-coalesced)")
print(f"WRITE_SIZE avg per step_kernel dispatch: {w} KiB over {nw} dispatches")
if f is not None and w is not None:
    print(f"raw bytes/launch = {(f + w) * 1024:.0f}; with FETCH x2 = {(2 * f + w) * 1024:.0f}")
