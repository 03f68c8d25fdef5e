cd $GRAFT_REPO_ROOT
run() { timeout 900 python bench.py "$@" --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], 'streams', d['config'].get('streams'))"; }
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_ops.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do
echo "r152 bf16 auto      : $(run --model resnet152 --stem standard --pixels 224 --images 2048 --steps 3 --warmup 1)"
echo "r152 bf16 2 streams : $(FB_WGRAD_STREAM=1 run --model resnet152 --stem standard --pixels 224 --images 2048 --steps 3 --warmup 1)"
done
echo "r152 gradreg auto: $(run --model resnet152 --stem standard --pixels 224 --images 1024 --grad-reg 0.5 --steps 2 --warmup 1)"
echo "r50 cifar auto   : $(run --model resnet50 --pixels 32 --images 12544 --steps 3 --warmup 1)"
echo "r18 headline     : $(run --steps 5 --warmup 2 --no-side-configs)"
