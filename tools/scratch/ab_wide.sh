#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== default"; NO_WGRAD=1 python tools/conv_microbench.py l2g l3g l4g 2>&1 | grep -v amdgpu
echo "== FB_H4_WIDE=8,16"; FB_H4_WIDE=8,16 NO_WGRAD=1 python tools/conv_microbench.py l2g l3g 2>&1 | grep -v amdgpu
