#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_engine.py tests/test_gpu_bf16_parity.py -q -x 2>&1 | tail -4
bash tools/scratch/ab_step.sh FB_WGRAD_CHAIN=0
