#!/bin/bash
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r5h; mkdir -p $out
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv_fwd_and_stats or conv_dgrad" 2>&1 | grep -v amdgpu.ids | tail -n 4 ) > $out/tests.log; tail -n 2 $out/tests.log
r152() { timeout 900 python bench.py --model resnet152 --stem standard --pixels 224 --images 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
echo "r152 default (gemm fwd): $(r152)"
echo "r152 FB_C1G=0: $(FB_C1G=0 r152)"
echo "r152 FB_C1G=2 (gemm fwd + dgrad): $(FB_C1G=2 r152)"
echo "r152 default again: $(r152)"
echo "r152 FB_C1G=0 again: $(FB_C1G=0 r152)"
