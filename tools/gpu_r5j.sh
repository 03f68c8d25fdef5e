#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in "X=0" "FB_H4_PHASE=1,1" "FB_H4_PHASE=1,2" "FB_H4_PHASE=1,3" "FB_H4_PHASE=2,2" "FB_H4_PHASE=2,3" "X=0"; do
  echo "== $v"; ( env $v NO_WGRAD=1 timeout 300 python tools/conv_microbench.py l2g l3g l4g 2>&1 | grep -v amdgpu.ids )
done
