#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2k
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_training.py -m gpu -q -x -k "sharded or rccl or lmdb or two_rank" > gpurun_out/r2k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2k/pytest.log
grep -n "passed\|failed\|^FAILED\|Error" gpurun_out/r2k/pytest.log | tail -12
