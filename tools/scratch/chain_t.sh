#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_ops.py -q -x -k "chunk_chain" 2>&1 | tail -1
python tools/scratch/chain_bench.py 2>&1 | grep -E "per-chunk|chained"
bash tools/scratch/ab_step.sh FB_WGRAD_CHAIN=0
