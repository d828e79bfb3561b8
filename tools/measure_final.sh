#!/bin/bash
# Final pass on the GPU box: smoke, bench lines of every workload, HBM traffic counters of the headline step, then the
# headline bench line with the refreshed traffic file.  usage: bash tools/measure_final.sh  (outputs: gpurun_out/m_r3g/)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/m_r3g
mkdir -p $O
cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
# (the headline line is taken LAST, after the counter passes: its roofline.traffic reads the file they refresh)
for m in crnn conv3d swinrnn clusterswin; do
  timeout -k 10 300 python bench.py --model $m > $O/bench_$m.json 2> $O/bench_$m.err
done
timeout -k 10 300 python bench.py --arch resnet152 --lstm-layers 3 --no-cpu-baseline > $O/bench_resnet152.json 2> $O/bench_resnet152.err
for m in swinrnn clusterswin; do
  timeout -k 10 300 python bench.py --model $m --batch 32 --no-cpu-baseline > $O/bench_${m}_b32.json 2> $O/bench_${m}_b32.err
done
bash tools/pmc_bench.sh gpurun_out/m_r3g/pmc_hbm
python tools/hbm_traffic.py $O/pmc_hbm $O/hbm_traffic.json > /dev/null
find $O -name "*.csv" -path "*pmc*" -delete
find $O -type d -empty -delete
python - <<'PY'
import json, subprocess
d = json.load(open("gpurun_out/m_r3g/hbm_traffic.json"))
d["commit"] = None
json.dump(d, open("profiles/r03_hbm_traffic.json", "w"), indent=1)
PY
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo final-measured
