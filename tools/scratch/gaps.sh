cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="--model resnet152 --stem standard --pixels 224 --images 2048 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --serialize"
for v in 0 1; do
rm -rf /tmp/kt$v; mkdir -p /tmp/kt$v
FB_C1G=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$v -o bench -- python3 bench.py $args > /tmp/kt$v/bench.log 2>&1
echo "=== FB_C1G=$v"; grep -o '"ms_per_step": [0-9.]*' /tmp/kt$v/bench.log | head -1
python3 tools/kernel_gaps.py /tmp/kt$v 12
done
