cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "masked or conv_dgrad or fused_bn_backward" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_training.py tests/test_gpu_bf16_structural.py tests/test_gpu_bf16_parity.py -m gpu -x -q 2>&1 | tail -3
run() { timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-side-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
echo "masked addend in halo4 : $(run)"
echo "materialised           : $(FB_H4_NO_MASK=1 run)"
done
