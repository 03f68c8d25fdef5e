cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_c1gexp.so
W=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_c1gnostag.so
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or conv1x1_kernels_agree" 2>&1 | tail -2
for r in 1 2; do
echo "== igemm      : $(IMGS=2048 NO_WGRAD=1 python tools/conv_microbench.py b3b 2>&1 | grep 'fwd')"
echo "== stagger    : $(FB_C1G=1 IMGS=2048 NO_WGRAD=1 python tools/conv_microbench.py b3b 2>&1 | grep 'fwd')"
echo "== no stagger : $(FB_LIB_PATH=$W FB_C1G=1 IMGS=2048 NO_WGRAD=1 python tools/conv_microbench.py b3b 2>&1 | grep 'fwd')"
done
for e in 1 2 3 11 16 18; do
  echo "== EXP=$e: $(FB_LIB_PATH=$V FB_C1G=1 FB_C1G_EXP=$e IMGS=2048 NO_WGRAD=1 python tools/conv_microbench.py b3b 2>&1 | grep 'fwd')"
done
