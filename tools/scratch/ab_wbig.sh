#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_ops.py -q -x -k "wgrad" 2>&1 | tail -3
echo "== big"; SPLITS=5,1 python tools/conv_microbench.py l1g l2g 2>&1 | grep -E "wgrad"
echo "== 64-pixel steps"; FB_WGRAD3_BIG=0 SPLITS=5,1 python tools/conv_microbench.py l1g l2g 2>&1 | grep -E "wgrad"
bash tools/scratch/ab_step.sh FB_WGRAD3_BIG=0
