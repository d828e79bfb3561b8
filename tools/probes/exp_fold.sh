set -e
O=gpurun_out/fold; mkdir -p $O
timeout -k 10 500 python -u -m pytest tests/test_gpu_igemm.py tests/test_gpu_benchsize.py tests/test_gpu_models.py -m gpu -x -q --timeout 180 --timeout-method thread > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for lib in "" packonly "" packonly; do
  P=; [ -n "$lib" ] && P=/root/repo/video-classification_amd/libvc_hip_$lib.so
  VC_LIB_PATH=$P timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_${lib:-halving}.json 2> $O/bench.err
  echo "lib=${lib:-halving} $(python -c "import json; print(json.load(open('$O/bench_${lib:-halving}.json'))['ms_per_step'])")"
done
for lib in "" packonly; do
  P=; [ -n "$lib" ] && P=/root/repo/video-classification_amd/libvc_hip_$lib.so
  VC_LIB_PATH=$P timeout -k 10 300 python tools/bench_conv.py --reps 4 --flush --passes fwd --stats --only l1b0.c2,l1b0.c3,l2b0.c3,l2b0.ds,l3b0.c3,l3b1.c2 > $O/cold_stats_${lib:-halving}.txt 2>&1
done
paste <(awk '{print $1, $6, $9, $10}' $O/cold_stats_halving.txt) <(awk '{print $9, $10}' $O/cold_stats_packonly.txt)
