set -e
O=gpurun_out/aonce; mkdir -p $O
for lib in "" aonce1 aonce2 "" aonce1 aonce2; do
  P=; [ -n "$lib" ] && P=/root/repo/video-classification_amd/libvc_hip_$lib.so
  VC_LIB_PATH=$P timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_${lib:-base}.json 2> $O/bench.err
  echo "lib=${lib:-base} $(python -c "import json; d=json.load(open('$O/bench_${lib:-base}.json')); print(d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['roofline']['other_kernels'].items()})")"
done
