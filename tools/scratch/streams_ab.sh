cd $GRAFT_REPO_ROOT
run() { timeout 900 python bench.py "$@" --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for r in 1 2; do
echo "r152 gradreg 2 streams: $(run --model resnet152 --stem standard --pixels 224 --images 1024 --grad-reg 0.5 --steps 2 --warmup 1)"
echo "r152 gradreg 1 stream : $(FB_WGRAD_STREAM=0 run --model resnet152 --stem standard --pixels 224 --images 1024 --grad-reg 0.5 --steps 2 --warmup 1)"
echo "r50 224 2 streams: $(run --model resnet50 --stem standard --pixels 224 --images 2048 --steps 3 --warmup 1)"
echo "r50 224 1 stream : $(FB_WGRAD_STREAM=0 run --model resnet50 --stem standard --pixels 224 --images 2048 --steps 3 --warmup 1)"
echo "r50 cifar 32px 2 streams: $(run --model resnet50 --pixels 32 --images 12544 --steps 3 --warmup 1)"
echo "r50 cifar 32px 1 stream : $(FB_WGRAD_STREAM=0 run --model resnet50 --pixels 32 --images 12544 --steps 3 --warmup 1)"
done
echo "r18 2 streams: $(run --steps 5 --warmup 2 --no-side-configs)"
echo "r18 1 stream : $(FB_WGRAD_STREAM=0 run --steps 5 --warmup 2 --no-side-configs)"
