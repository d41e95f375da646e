#!/bin/bash
# End-of-round run on the GPU box: smoke, whole GPU suite, the default bench line (un-profiled), then the profiles
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r1f}
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1 || { tail -20 gpurun_out/${TAG}_smoke.log; exit 1; }
tail -1 gpurun_out/${TAG}_smoke.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1 || { tail -40 gpurun_out/${TAG}_tests.log; exit 1; }
tail -1 gpurun_out/${TAG}_tests.log
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench.json')); print('value %.4e ms/step %.5f kernel_us %.2f frac %.4f cpu %.0f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_us'], d['roofline']['frac'], d['cpu_baseline']['value'])); print([ (s['envs'], round(s['kernel_us'],1), round(s['frac_of_8TBps'],3)) for s in d['sweep']])"
bash scripts/bench_all.sh ${TAG}_all > gpurun_out/${TAG}_all.txt 2>&1; cat gpurun_out/${TAG}_all.txt
timeout -k 10 1500 bash scripts/profile_round.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1 || { tail -30 gpurun_out/${TAG}_profile.log; exit 1; }
grep -A3 "^== " gpurun_out/${TAG}_profile.log | head -60
