#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for poll in 1 2 4 8; do echo "== FB_BNF_POLL=$poll"; FB_BNF_POLL=$poll TRACE=1 python tools/bn_bwd_microbench.py 2>&1 | grep -v amdgpu | grep -E "^\| (64|128|256|512)|C=64"; done
