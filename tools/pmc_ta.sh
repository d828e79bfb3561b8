#!/bin/bash
# Texture-addresser (vector L1 front end) busy counters for one layer: bash tools/pmc_ta.sh <layer> <pass> <outdir>
set -e
layer=$1; pass=$2; out=$3
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
for grp in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_BUSY_avr TA_BUSY_max" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $grp -d $R/$out/$tag -o pmc --output-format csv -- python3 $R/tools/bench_conv.py --only $layer --passes $pass --reps 2 > $R/$out/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $R/$out/$tag.log; }
done
echo done
