set -e
O=gpurun_out/stats_epi; mkdir -p $O
timeout -k 10 500 python -u -m pytest tests/test_gpu_igemm.py tests/test_gpu_benchsize.py tests/test_gpu_kernels.py tests/test_gpu_swin.py -m gpu -x -q --timeout 180 --timeout-method thread > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -1 $O/tests.log
timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python -c "import json; print(json.load(open('$O/bench.json'))['ms_per_step'])"
timeout -k 10 300 python tools/bench_conv.py --reps 4 --flush --passes fwd > $O/cold_plain.txt 2>&1
timeout -k 10 300 python tools/bench_conv.py --reps 4 --flush --passes fwd --stats > $O/cold_stats.txt 2>&1
paste <(awk '{print $1, $6, $9, $10}' $O/cold_plain.txt) <(awk '{print $9, $10}' $O/cold_stats.txt)
