#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_ops.py -q -x -k "chunk_chain" 2>&1 | tail -1
echo "== 5 stages"; python tools/scratch/chain_bench.py 2>&1 | grep -E "per-chunk|chained, (4|8|16)"
echo "== 3 stages"; FB_LIB_PATH=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_st3.so python tools/scratch/chain_bench.py 2>&1 | grep -E "chained, (4|8|16)"
echo "== 8 stages"; FB_LIB_PATH=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_st8.so python tools/scratch/chain_bench.py 2>&1 | grep -E "chained, (4|8|16)"
