cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or conv1x1_kernels_agree" 2>&1 | tail -2
D=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants
for r in 1 2; do
echo "== igemm : $(IMGS=2048 NO_WGRAD=1 python tools/conv_microbench.py b3b 2>&1 | grep 'fwd')"
echo "== IQ=2  : $(FB_C1G=1 IMGS=2048 NO_WGRAD=1 python tools/conv_microbench.py b3b 2>&1 | grep 'fwd')"
for q in 1 3 4; do
echo "== IQ=$q  : $(FB_LIB_PATH=$D/libfbengine_g1iq$q.so FB_C1G=1 IMGS=2048 NO_WGRAD=1 python tools/conv_microbench.py b3b 2>&1 | grep 'fwd')"
done
done
bash tools/scratch/g1t.sh 2>&1 | grep -v "^$"
