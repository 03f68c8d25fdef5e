#!/bin/bash
# round 5: the pipelined streaming 1x1 kernel -- parity tests, then microbenchmarks against the round-3 form (same box)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5b
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv_fwd_and_stats or conv_dgrad" 2>&1 | grep -v amdgpu.ids | tail -n 25 ) > gpurun_out/r5b/tests.log
tail -n 5 gpurun_out/r5b/tests.log
for v in "FB_C1S_PIPE=0 FB_C1S_ADD_ASM=0" "FB_C1S_PIPE=0" "FB_C1S_PIPE=1" "FB_C1S_PIPE=1 FB_C1P_NW=4"; do
  echo "== $v"
  ( env $v IMGS=1024 ADD=1 NO_WGRAD=1 timeout 300 python tools/conv_microbench.py b1a b1b b2a b2b b3a b3b 2>&1 | grep -v amdgpu.ids )
  ( env $v ADD=1 NO_WGRAD=1 timeout 300 python tools/conv_microbench.py s2 s3 s4 2>&1 | grep -v amdgpu.ids )
done > gpurun_out/r5b/micro.log 2>&1
cat gpurun_out/r5b/micro.log
