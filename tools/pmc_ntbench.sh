#!/bin/bash
# Counter passes over the stand-alone NT kernel bench: bash tools/pmc_ntbench.sh <layer> <outdir>   (run on the GPU box)
set -e
layer=$1; out=$2
R=$GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$R/video-classification_amd NTB_PASS=${NTB_PASS:-0} NTB_REPS=1
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/$out
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 90 rocprofv3 --pmc $grp -d $R/$out/$tag -o pmc --output-format csv -- $R/tools/probes/nt_bench.bin $layer > $R/$out/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $R/$out/$tag.log; exit 1; }
done
python3 $R/tools/pmc_summary.py $R/$out igemm
