#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5d
python tools/scratch/pipe_debug.py 2>&1 | grep -v amdgpu.ids
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv_fwd_and_stats or conv_dgrad" 2>&1 | grep -v amdgpu.ids | tail -n 8 ) > gpurun_out/r5d/tests.log
tail -n 3 gpurun_out/r5d/tests.log
V=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants
for v in "X=default" "FB_LIB_PATH=$V/libfbengine_l0s1.so" "FB_LIB_PATH=$V/libfbengine_l1s0.so" "FB_LIB_PATH=$V/libfbengine_l0s0.so" "FB_C1S_PIPE=0"; do
  echo "== $v"
  ( env $v IMGS=1024 ADD=1 NO_WGRAD=1 timeout 300 python tools/conv_microbench.py b1a b1b b2a b2b b3a b3b 2>&1 | grep -v amdgpu.ids )
  ( env $v ADD=1 NO_WGRAD=1 timeout 300 python tools/conv_microbench.py s2 s3 s4 2>&1 | grep -v amdgpu.ids )
done > gpurun_out/r5d/micro.log 2>&1
cat gpurun_out/r5d/micro.log
