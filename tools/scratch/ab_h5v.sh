#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_ops.py -q -x -k "conv_fwd or dgrad" 2>&1 | tail -2
echo "== vert"; ADD=1 NO_WGRAD=1 python tools/conv_microbench.py l1g 2>&1 | grep -E "fwd|dgrad"
echo "== strided"; FB_H5_VERT=0 ADD=1 NO_WGRAD=1 python tools/conv_microbench.py l1g 2>&1 | grep -E "fwd|dgrad"
bash tools/scratch/ab_step.sh FB_H5_VERT=0
