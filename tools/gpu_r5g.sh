#!/bin/bash
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r5g; mkdir -p $out
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv_fwd_and_stats or conv_dgrad" 2>&1 | grep -v amdgpu.ids | tail -n 12 ) > $out/tests.log; tail -n 6 $out/tests.log
for v in "FB_C1G=1" "FB_C1G=0"; do
  echo "== $v"
  ( env $v IMGS=1024 NO_WGRAD=1 timeout 300 python tools/conv_microbench.py b2a b2b b3a b3b b4a b4b 2>&1 | grep -v amdgpu.ids )
done
