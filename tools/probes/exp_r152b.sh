O=gpurun_out/r152; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 300 python -u -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "resnet152 and f32" --timeout 420 --timeout-method thread > $O/$name.log 2>&1
  echo "$name: $(grep -E "worst gradient error" $O/$name.log | grep -v print | cut -c30-200)"; }
run oldlib VC_LIB_PATH=/root/repo/video-classification_amd/libvc_hip_old.so
run base X=1
run nostats VC_FUSE_BN_STATS=0
run notail VC_FUSE_BN_TAIL=0
run noepi VC_FUSE_EPILOGUE=0
run nomask VC_BN_INPLACE_MASK=0
run nolstm VC_PERSISTENT_LSTM=0 VC_PERSISTENT_LSTM_BWD=0
run poison VC_POISON=1
