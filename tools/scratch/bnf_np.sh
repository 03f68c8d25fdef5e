#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for tag in np8 np4; do
 for poll in -1 2; do echo "== $tag FB_BNF_POLL=$poll"; FB_LIB_PATH=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_$tag.so FB_BNF_POLL=$poll TRACE=1 python tools/bn_bwd_microbench.py 2>&1 | grep -v amdgpu | grep -E "^\| (64|128|256|512)|C=64|Error|error"; done
done
