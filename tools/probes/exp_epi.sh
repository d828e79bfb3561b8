set -e
O=gpurun_out/epi_nt; mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q --timeout 180 --timeout-method thread > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for v in 0 1 2 3 0; do
  P=/root/repo/video-classification_amd/libvc_hip_epint$v.so; [ $v = 0 ] && P=
  VC_LIB_PATH=$P timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_epi$v.json 2> $O/bench_epi$v.err
  echo "EPI_NT=$v $(python -c "import json; print(json.load(open('$O/bench_epi$v.json'))['ms_per_step'])")"
done
for m in crnn conv3d; do for v in 0 15; do
  VC_BN_NT=$v timeout -k 10 120 python bench.py --model $m --no-cpu-baseline --steps 30 --warmup 5 > $O/bench_${m}_bn$v.json 2> $O/bench_${m}_bn$v.err
  echo "$m VC_BN_NT=$v $(python -c "import json; print(json.load(open('$O/bench_${m}_bn$v.json'))['ms_per_step'])")"
done; done
