#!/bin/bash
# Same-box A/B of product ray kernels compiled with ONE pass form only (record: profiles/r4/ab_same_box.txt).  The variant libraries are
# not built from the tree as it is: in dockauv_step.hip.inc's ray stage replace the dispatch
#     if (pad_log2 == 6) run_passes(IntC<6>{}); else if (!LOG || pad_log2 == 4) run_passes(IntC<4>{}); ...
# by `if (!LOG && ONLY_PL == 4) run_passes(IntC<4>{}); else if (!LOG && ONLY_PL == 6) run_passes(IntC<6>{}); else ...` and build
#     scripts/build_variant.py xpl4 -DONLY_PL=4      (run on config 3: 16-beam fan)
#     scripts/build_variant.py xpl6 -DONLY_PL=6      (run on configs 4 / 5: 63-ray fan)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
one() {  # lib config extra
  DOCKAUV_LIB=$GRAFT_REPO_ROOT/gym_dockauv_amd/lib/$1 timeout -k 10 200 python bench.py --config $2 $3 --no-configs --no-sweep --no-cpu --no-closed-loop --steps 1000 --warmup 500 --min-seconds 0.1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s config $2 $3 launch_us=%.2f resident_us=%.2f finite=%s' % ('$1', d['roofline']['kernel_us'], (d.get('sequence_resident') or {}).get('us_per_step_events', float('nan')), d['config']['obs_finite']))" || echo "$1 $2 FAILED"
}
for rep in 1 2; do
  one libdockauv.so 3 ""; one libdockauv_xpl4.so 3 ""
  one libdockauv.so 4 ""; one libdockauv_xpl6.so 4 ""
  one libdockauv.so 5 ""; one libdockauv_xpl6.so 5 ""
  one libdockauv.so 5 "--layout vehicle_sorted"; one libdockauv_xpl6.so 5 "--layout vehicle_sorted"
done | tee gpurun_out/r4/ab_onlypl.txt
