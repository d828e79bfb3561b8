#!/bin/bash
# Counter passes for one conv layer of tools/bench_conv.py: usage tools/pmc_conv.sh <layer> <pass> <outdir>
set -e
layer=$1; pass=$2; out=$3
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAVES" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $grp -d $R/$out/$tag -o pmc --output-format csv -- python3 $R/tools/bench_conv.py --only $layer --passes $pass --reps 2 > $R/$out/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $R/$out/$tag.log; exit 1; }
done
echo done
