set -e
O=/root/repo/gpurun_out/swin_nt; mkdir -p $O
R=/root/repo
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace -o r -- python3 $R/bench.py --model swinrnn --batch 32 --no-cpu-baseline --no-graph --steps 3 --warmup 1 > $O/trace.log 2>&1
cd $R
python tools/rocpd_stats.py $O/trace/r_results.db $O/kernel_stats_swinrnn_b32.csv 4 > /dev/null
rm -rf $O/trace
for lib in "" epint1 epint3 lnnt3 lnnt12 lnnt15 ""; do
  P=; [ -n "$lib" ] && P=$R/video-classification_amd/libvc_hip_$lib.so
  for b in 8 32; do
    VC_LIB_PATH=$P timeout -k 10 120 python bench.py --model swinrnn --batch $b --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_${lib:-base}_b$b.json 2> $O/bench_${lib:-base}_b$b.err
    echo "lib=${lib:-base} b=$b $(python -c "import json; print(json.load(open('$O/bench_${lib:-base}_b$b.json'))['ms_per_step'])")"
  done
done
