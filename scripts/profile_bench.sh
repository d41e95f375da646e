#!/bin/bash
# Profile bench.py with rocprofv3 on the GPU box.  Usage: scripts/profile_bench.sh <round-tag> [bench args...]
# Writes summaries under gpurun_out/prof_<tag>/ ; copy the ones to keep into profiles/.
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu --no-sweep "$@" > $OUT/bench_trace.json 2> $OUT/bench_trace.err || { tail -5 $OUT/bench_trace.err; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -5 $OUT/kernel_stats.csv
# HBM traffic counters, separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu --no-sweep "$@" > /dev/null 2> $OUT/pmc_fetch.err || tail -3 $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu --no-sweep "$@" > /dev/null 2> $OUT/pmc_write.err || tail -3 $OUT/pmc_write.err
python3 scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1 || true
cat $OUT/pmc_summary.txt
