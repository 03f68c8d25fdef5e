#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== default"; SPLITS=1 python tools/conv_microbench.py l4g 2>&1 | grep -E "wgrad"
echo "== one workgroup per CU"; FB_LIB_PATH=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_onewg.so SPLITS=1 python tools/conv_microbench.py l4g 2>&1 | grep -E "wgrad"
