#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_ops.py -q -x -k "dgrad" 2>&1 | tail -2
NO_WGRAD=1 python tools/conv_microbench.py d2 d3 d4 2>&1 | grep -E "dgrad"
