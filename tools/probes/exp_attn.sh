set -e
O=/root/repo/gpurun_out/attn_nt; mkdir -p $O
for lib in "" attnf attnfb ""; do
  P=; [ -n "$lib" ] && P=/root/repo/video-classification_amd/libvc_hip_$lib.so
  for b in 8 32; do
    VC_LIB_PATH=$P timeout -k 10 120 python bench.py --model swinrnn --batch $b --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_${lib:-base}_b$b.json 2> $O/bench.err
    echo "lib=${lib:-base} b=$b $(python -c "import json; print(json.load(open('$O/bench_${lib:-base}_b$b.json'))['ms_per_step'])")"
  done
done
