#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -q -x -k "accumulate or mt_ or multi or engine or groups" 2>&1 | tail -2
bash tools/scratch/ab_lib.sh ng8
