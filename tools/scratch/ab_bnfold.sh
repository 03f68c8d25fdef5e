#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== default"; NO_WGRAD=1 python tools/conv_microbench.py l2g l3g 2>&1 | grep fwd
echo "== BN + ReLU of the producer applied to the staged halo slice (timing only)"; FB_LIB_PATH=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_bnfold.so NO_WGRAD=1 python tools/conv_microbench.py l2g l3g 2>&1 | grep fwd
echo "== default again"; NO_WGRAD=1 python tools/conv_microbench.py l2g l3g 2>&1 | grep fwd
