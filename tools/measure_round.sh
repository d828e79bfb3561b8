#!/bin/bash
# One measurement pass on the GPU box: kernel trace of the eager bench, bench lines (hipGraph) of the five
# workloads, per-layer GEMM table.  usage: bash tools/measure_round.sh <tag>   (outputs under gpurun_out/m_<tag>/)
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/m_$tag
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace -o r -- python3 $R/bench.py --no-cpu-baseline --no-graph --steps 3 --warmup 1 > $O/trace.log 2>&1
cd $R
python tools/rocpd_stats.py $O/trace/r_results.db $O/kernel_stats.csv 4 > /dev/null
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for m in crnn conv3d swinrnn clusterswin; do
  timeout -k 10 300 python bench.py --model $m > $O/bench_$m.json 2> $O/bench_$m.err
done
timeout -k 10 300 python bench.py --arch resnet152 --lstm-layers 3 --no-cpu-baseline > $O/bench_resnet152.json 2> $O/bench_resnet152.err
timeout -k 10 300 python tools/bench_conv.py --reps 5 > $O/conv_layers.txt 2>&1
echo measured
