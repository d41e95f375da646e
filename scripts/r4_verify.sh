#!/bin/bash
# Re-entry check of a restored tree on the GPU box: smoke, the whole GPU suite, then the driver-shaped bench line
# (--steps 20 --warmup 5) under the host's wait modes (blocked wait = default, ROC_ACTIVE_WAIT_TIMEOUT = active wait for that
# many us before blocking): what a 20-step region pays around its launches -> gpurun_out/r4/verify_*.{log,txt}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/verify_smoke.log 2>&1 || { tail -20 gpurun_out/r4/verify_smoke.log; exit 1; }
tail -1 gpurun_out/r4/verify_smoke.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4/verify_tests.log 2>&1 || { tail -40 gpurun_out/r4/verify_tests.log; exit 1; }
tail -1 gpurun_out/r4/verify_tests.log
for rep in 1 2 3; do
for mode in blocked active100 active1000; do
  case $mode in
    blocked) unset ROC_ACTIVE_WAIT_TIMEOUT ;;
    active100) export ROC_ACTIVE_WAIT_TIMEOUT=100 ;;
    active1000) export ROC_ACTIVE_WAIT_TIMEOUT=1000 ;;
  esac
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu --no-configs --no-sweep --no-closed-loop --no-resident 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s value %.4e  ms_per_step %.5f  kernel_us %.3f' % ('$mode', d['value'], d['ms_per_step'], d['roofline']['kernel_us']))" || echo "$mode FAILED"
done; done | tee gpurun_out/r4/verify_wait_modes.txt
