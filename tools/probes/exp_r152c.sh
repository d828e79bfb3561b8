O=gpurun_out/r152; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 300 python -u -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "resnet152 and f32" --timeout 420 --timeout-method thread > $O/$name.log 2>&1
  echo "$name: $(grep -E "worst gradient error" $O/$name.log | grep -v print | cut -c30-130)"; }
for i in 1 2 3; do
run new$i X=1
run old$i VC_LIB_PATH=/root/repo/video-classification_amd/libvc_hip_old.so
done
run new_bn0 VC_BN_NT=0
run new_y Y=1
