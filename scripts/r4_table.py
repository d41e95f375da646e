#!/usr/bin/env python3
"""Table of a bench.py JSON line (stdin or file): per workload the per-launch kernel time, fraction of 8 TB/s, the resident
sequence's time per step, the closed-loop time."""
import json
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
line = [l for l in txt.splitlines() if l.startswith("{")][-1]
d = json.loads(line)
rows = d.get("configs") or [dict(workload=d["config"]["workload"], envs=d["config"]["envs_per_gpu"], kernel_us=d["roofline"]["kernel_us"],
                                 roofline=d["roofline"], sequence_resident=d.get("sequence_resident"), closed_loop=d.get("closed_loop"))]
print(f"{'workload':58s} {'envs':>8s} {'launch us':>9s} {'frac':>6s} {'resident us':>11s} {'frac':>6s} {'ratio':>6s} {'closed us':>9s}")
for s in rows:
    name = s["workload"].split(":")[0] + (" " + s.get("layout", "") if s.get("layout") else "") + (" dense" if "ray-dense" in s["workload"] else "")
    r = s.get("sequence_resident") or {}
    cl = s.get("closed_loop") or {}
    print(f"{name:58s} {s['envs']:8d} {s['kernel_us']:9.2f} {s['roofline']['frac']:6.3f} "
          f"{r.get('us_per_step_events', float('nan')):11.2f} {r.get('frac_of_8TBps', float('nan')):6.3f} {r.get('vs_per_launch') or float('nan'):6.3f} "
          f"{cl.get('us_per_step', float('nan')):9.2f}")
for sw in d.get("sweep", []):
    print(f"sweep envs {sw['envs']:8d}  kernel_us {sw['kernel_us']:8.2f}  frac {sw['frac_of_8TBps']:.3f}  resident_us {sw.get('sequence_resident_us_per_step') or float('nan'):8.2f}  frac {sw.get('sequence_resident_frac_of_8TBps') or float('nan'):.3f}")
