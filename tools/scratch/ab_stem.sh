#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_ops.py -q -x -k "conv_fwd" 2>&1 | tail -2
NO_WGRAD=1 python tools/conv_microbench.py stemg 2>&1 | grep fwd
python -m pytest tests/test_gpu_engine.py tests/test_gpu_bf16_structural.py -q -x 2>&1 | tail -2
