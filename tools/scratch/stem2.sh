cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "wgrad" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_training.py -m gpu -x -q 2>&1 | tail -2
run() { timeout 900 python bench.py --model resnet152 --stem standard --pixels 224 --images 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], 'G', d['config']['chunk_group'], 'streams', d['config'].get('streams'))"; }
for r in 1 2 3; do
echo "r152: $(run)"
done
