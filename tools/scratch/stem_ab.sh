cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "masked_addend or conv_dgrad" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_training.py tests/test_gpu_bf16_structural.py -m gpu -x -q 2>&1 | tail -3
run() { timeout 900 python bench.py --model resnet152 --stem standard --pixels 224 --images 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], 'streams', d['config'].get('streams'))"; }
for r in 1 2; do
echo "masked addend in the 1x1 dgrad : $(run)"
echo "materialised                   : $(FB_C1P_NO_MASK=1 run)"
done
