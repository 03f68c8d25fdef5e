#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5c
python tools/scratch/bw_probe.py 2>&1 | grep -v amdgpu.ids
for v in "FB_C1P_EXP=0" "FB_C1P_EXP=1" "FB_C1P_EXP=2" "FB_C1P_EXP=3" "FB_C1P_EXP=0 FB_C1P_NW=4" "FB_C1P_EXP=1 FB_C1P_NW=4" "FB_C1P_EXP=2 FB_C1P_NW=4"; do
  echo "== $v"
  ( env $v IMGS=1024 ADD=1 NO_WGRAD=1 timeout 300 python tools/conv_microbench.py b1a b2a b3a b3b 2>&1 | grep -v amdgpu.ids )
done
