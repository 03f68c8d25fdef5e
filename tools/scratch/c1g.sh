cd $GRAFT_REPO_ROOT
run() { timeout 900 python bench.py --model resnet152 --stem standard --pixels 224 --images 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], 'G', d['config']['chunk_group'], 'streams', d['config'].get('streams'))"; }
for r in 1 2; do
echo "default  : $(run)"
echo "FB_C1G=2 : $(FB_C1G=2 run)"
echo "FB_C1G=1 : $(FB_C1G=1 run)"
echo "2 streams: $(FB_WGRAD_STREAM=1 run)"
done
