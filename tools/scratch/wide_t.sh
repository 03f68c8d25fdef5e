#!/bin/bash
cd "$GRAFT_REPO_ROOT"
FB_WGRAD3_WIDE=1 python -m pytest tests/test_gpu_ops.py -q -x -k "test_conv_wgrad" 2>&1 | tail -2
echo "== 64-channel tiles"; SPLITS=1 python tools/conv_microbench.py l4g 2>&1 | grep wgrad
echo "== 128-channel tiles, 8 waves"; FB_WGRAD3_WIDE=1 SPLITS=1 python tools/conv_microbench.py l4g 2>&1 | grep wgrad
bash tools/scratch/ab_step.sh FB_WGRAD3_WIDE=1
