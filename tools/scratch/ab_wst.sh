#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_ops.py -q -x -k "wgrad" 2>&1 | tail -2
SPLITS=1 python tools/conv_microbench.py l2g l3g l4g 2>&1 | grep -E "wgrad"
SPLITS=5 python tools/conv_microbench.py l1g 2>&1 | grep -E "wgrad"
for i in 1 2; do echo "step: $(python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["ms_per_step"])')"; done
