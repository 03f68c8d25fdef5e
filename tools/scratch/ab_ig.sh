#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for st in 2 1; do echo "== FB_IGEMM_STAGES=$st"; FB_IGEMM_STAGES=$st NO_WGRAD=1 python tools/conv_microbench.py d2 d3 d4 2>&1 | grep -E "fwd"; done
FB_IGEMM_STAGES=1 python -m pytest tests/test_gpu_ops.py -q -x -k "conv_fwd or dgrad" 2>&1 | tail -2
bash tools/scratch/ab_step.sh FB_IGEMM_STAGES=1
