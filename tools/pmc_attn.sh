#!/bin/bash
# Instruction / MFMA counters of the window-attention kernels: usage tools/pmc_attn.sh <pass: fwd|bwd> <outdir>
set -e
pass=$1; out=$2
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAVES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $grp -d $R/$out/$tag -o pmc --output-format csv -- python3 $R/tools/bench_attn.py --passes $pass --reps 2 > $R/$out/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $R/$out/$tag.log; exit 1; }
done
echo done
