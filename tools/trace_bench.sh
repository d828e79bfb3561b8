#!/bin/bash
# Kernel trace of the eager headline step: bash tools/trace_bench.sh <tag> [bench args...]  (-> gpurun_out/tr_<tag>/kernel_stats.csv)
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tr_$tag
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace -o r -- python3 $R/bench.py --no-cpu-baseline --no-graph --steps 3 --warmup 1 "$@" > $O/trace.log 2>&1
cd $R
python tools/rocpd_stats.py $O/trace/r_results.db $O/kernel_stats.csv 4 skip-first-step > /dev/null
[ -n "$KEEP_TRACE" ] || rm -rf $O/trace
echo traced $tag
