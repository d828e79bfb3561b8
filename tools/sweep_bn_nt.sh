#!/bin/bash
# Cache-policy sweep of the BatchNorm streaming passes (VC_BN_NT, two bits per kernel: 1 = non-temporal loads,
# 2 = non-temporal stores; bits 0-1 forward apply, 2-3 backward apply, 4-5 backward reduce; VC_BN_NT_MIN_MB = smallest
# tensor that gets the policy): the headline step (and optionally the kernel table) for each setting.
# usage: bash tools/sweep_bn_nt.sh <out-dir> <policy>[:<min MB>] ...     (TABLE=1 also runs tools/bench_bn.py)
set -e
O=$1; shift
mkdir -p $O
for s in "$@"; do
  v=${s%%:*}; mb=0; [[ $s == *:* ]] && mb=${s##*:}
  export VC_BN_NT=$v VC_BN_NT_MIN_MB=$mb
  [ -n "$TABLE" ] && timeout -k 10 120 python tools/bench_bn.py > $O/bn_nt${v}_$mb.txt 2>&1
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_nt${v}_$mb.json 2> $O/bench_nt${v}_$mb.err
  echo "VC_BN_NT=$v MIN_MB=$mb $(python -c "import json,sys; print(json.load(open('$O/bench_nt${v}_$mb.json'))['ms_per_step'])")"
done
