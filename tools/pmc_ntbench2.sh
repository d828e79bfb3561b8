#!/bin/bash
# Memory-path counter passes over the stand-alone NT kernel bench:
#   bash tools/pmc_ntbench2.sh <layer> <outdir>      (run on the GPU box; VC_NT8_VAR selects the kernel form)
set -e
layer=$1; out=$2
R=$GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$R/video-classification_amd NTB_PASS=${NTB_PASS:-0} NTB_REPS=2
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/$out
i=0
for grp in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TD_TD_BUSY_sum" \
           "TCP_TCP_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_CACHE_MISS_sum TCP_UTCL1_REQUEST_sum"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $grp -d $R/$out/p$i -o pmc --output-format csv -- $R/tools/probes/nt_bench.bin $layer > $R/$out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/$out/p$i.log; exit 1; }
done
python3 - $R/$out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        k = 'nt8q' if 'nt8q' in k else 'nt8p' if 'nt8p' in k else 'nt8' if 'nt8' in k else 'old' if 'igemm' in k else None
        if k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print('  %-44s %14.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
