O=gpurun_out/r152; mkdir -p $O
for i in 1 2; do
timeout -k 10 300 python -u -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "resnet152 and f32" --timeout 420 --timeout-method thread > $O/alone$i.log 2>&1
grep -E "worst gradient error|passed|failed" $O/alone$i.log | cut -c1-400
done
timeout -k 10 400 python -u -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "(resnet50 or resnet152) and f32" --timeout 420 --timeout-method thread > $O/both.log 2>&1
grep -E "worst gradient error|passed|failed" $O/both.log | cut -c1-400
timeout -k 10 400 python -u -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "resnet50 or (resnet152 and f32)" --timeout 420 --timeout-method thread > $O/both2.log 2>&1
grep -E "worst gradient error|passed|failed" $O/both2.log | cut -c1-400
