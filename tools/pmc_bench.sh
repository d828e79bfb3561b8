#!/bin/bash
# HBM traffic counters of the bench workload, one counter per pass (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2).
# usage (on the GPU box): bash tools/pmc_bench.sh <outdir-under-repo>
set -e
out=$1
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c -d $R/$out/$c -o pmc --output-format csv -- python3 $R/bench.py --no-graph --no-cpu-baseline --steps 2 --warmup 1 > $R/$out/$c.log 2>&1 || { echo "pass $c failed"; tail -3 $R/$out/$c.log; exit 1; }
done
echo done
