#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> duration of dockauv::step_kernel by kind of dispatch: queued back to back on the stream
(bench.py's timed regions: one dockauv_step_sequence call; under the profiler such a dispatch's start is the previous one's
end, i.e. its duration is the step period) and after an idle gap (the per-dispatch event-timing loop: isolated launches)."""
import csv, sys, statistics as st
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "step_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
s = [int(r["Start_Timestamp"]) for r in rows]
e = [int(r["End_Timestamp"]) for r in rows]
dur = [(b - a) / 1e3 for a, b in zip(s, e)]
gap = [(s[i + 1] - e[i]) / 1e3 for i in range(len(rows) - 1)]
print(f"step_kernel dispatches {len(rows)}, average duration {sum(dur) / len(dur):.2f} us (what --stats reports)")
q = [d for d, g in zip(dur[1:], gap) if g < 3.0]
iso = [d for d, g in zip(dur[1:], gap) if g >= 3.0]
if q:
    sq = sorted(q)
    print(f"queued back to back (gap to the previous end < 3 us): n={len(q)}  median {st.median(q):.2f}  mean {sum(q) / len(q):.2f}  "
          f"p10 {sq[len(q) // 10]:.2f}  p90 {sq[9 * len(q) // 10]:.2f} us")
if iso:
    print(f"after an idle gap (isolated launches):                 n={len(iso)}  median {st.median(iso):.2f}  mean {sum(iso) / len(iso):.2f} us")
