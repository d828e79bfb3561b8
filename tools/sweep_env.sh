#!/bin/bash
# Cold-cache layer-table totals under tuning variables: bash tools/sweep_env.sh "VAR=a VAR=b ..." passes
# (each setting is one run of tools/bench_conv.py --flush; prints the totals lines)
passes=${2:-fwd,dgrad,wgrad}
for kv in $1; do
  echo "== $kv"
  env $kv timeout -k 10 200 python tools/bench_conv.py --reps 4 --flush --passes $passes 2>/dev/null | grep "^total" || exit 1
done
