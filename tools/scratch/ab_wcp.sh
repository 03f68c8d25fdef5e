#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_ops.py -q -x -k "wgrad or conv_fwd or dgrad" 2>&1 | tail -3
echo "== compact"; SPLITS=1 python tools/conv_microbench.py l4g 2>&1 | grep -E "wgrad"
echo "== padded"; FB_WGRAD3_COMPACT=0 SPLITS=1 python tools/conv_microbench.py l4g 2>&1 | grep -E "wgrad"
bash tools/scratch/ab_step.sh FB_WGRAD3_COMPACT=0
