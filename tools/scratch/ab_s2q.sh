#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for st in 2 1; do
echo "== FB_S2Q_STAGES=$st"; FB_S2Q_STAGES=$st FB_S2_QUAD_ALL=1 ADD=1 NO_WGRAD=1 python tools/conv_microbench.py d2 d3 d4 2>&1 | grep -E "dgrad"
done
echo "== d4 via implicit GEMM"; ADD=1 NO_WGRAD=1 python tools/conv_microbench.py d4 2>&1 | grep -E "dgrad"
python -m pytest tests/test_gpu_ops.py -q -x -k "dgrad" 2>&1 | tail -3
FB_S2Q_STAGES=1 FB_S2_QUAD_ALL=1 python -m pytest tests/test_gpu_ops.py -q -x -k "dgrad" 2>&1 | tail -3
