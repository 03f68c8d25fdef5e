#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_ops.py -q -x -k "conv_fwd or dgrad or bwd_stat or fused" 2>&1 | tail -3
echo "== compact"; ADD=1 NO_WGRAD=1 python tools/conv_microbench.py l4g 2>&1 | grep -E "fwd|dgrad"
echo "== padded, 128-channel tiles"; FB_H4_COMPACT=0 ADD=1 NO_WGRAD=1 python tools/conv_microbench.py l4g 2>&1 | grep -E "fwd|dgrad"
bash tools/scratch/ab_step.sh FB_H4_COMPACT=0
