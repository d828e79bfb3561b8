#!/bin/bash
# Second half of the measurement pass (run tools/measure_round.sh <tag> first): bash tools/measure_round3.sh <tag>  (outputs under gpurun_out/m_<tag>/)
#  1. tools/measure_round.sh: kernel trace of the eager headline step, bench lines of the workloads, layer table
#  2. cold-cache layer table, Swin workloads at 32 clips, the fusion CNN
#  3. HBM traffic counters of the headline step (FETCH_SIZE / WRITE_SIZE, one pass each)
#  4. MFMA / instruction counters of one MFMA-bound layer, forward + data gradient + weight gradient
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/m_$tag
mkdir -p $O
cd $R
timeout -k 10 300 python tools/bench_conv.py --reps 4 --flush > $O/conv_layers_cold.txt 2>&1
VC_BENCH_DUMP_LAUNCHES=$O/launches.txt timeout -k 10 300 python bench.py --no-cpu-baseline > /dev/null 2> $O/launches.err
for m in swinrnn clusterswin; do
  timeout -k 10 300 python bench.py --model $m --batch 32 --no-cpu-baseline > $O/bench_${m}_b32.json 2> $O/bench_${m}_b32.err
done
timeout -k 10 300 python tools/bench_fusion.py 64 224 > $O/fusion.txt 2>&1
bash tools/pmc_bench.sh gpurun_out/m_$tag/pmc_hbm
python tools/hbm_traffic.py $O/pmc_hbm $O/hbm_traffic.json > /dev/null
for pass in fwd dgrad wgrad; do
  bash tools/pmc_conv.sh l3b1.c2 $pass gpurun_out/m_$tag/pmc_l3_$pass > /dev/null
  python tools/pmc_summary.py $O/pmc_l3_$pass igemm > $O/mfma_l3b1c2_$pass.txt
done
find $O -name "*.csv" -path "*pmc*" -delete
find $O -type d -empty -delete
echo round-measured
