cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="--model resnet152 --stem standard --pixels 224 --images 2048 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --serialize"
for v in 0 2; do
rm -rf /tmp/kt$v; mkdir -p /tmp/kt$v
FB_C1G=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$v -o bench -- python3 bench.py $args > /tmp/kt$v/bench.log 2>&1
echo "=== FB_C1G=$v"
python3 tools/kernel_gaps.py /tmp/kt$v 3 | head -3
python3 tools/kernel_gaps.py /tmp/kt$v 0 "wgrad1x1_kernel<8, 8>"
python3 tools/kernel_gaps.py /tmp/kt$v 0 "wgrad3x3_v2_kernel<14"
python3 tools/kernel_gaps.py /tmp/kt$v 0 "bn_bwd_reduce_kernel<bf16_tag, false" | head -8
done
