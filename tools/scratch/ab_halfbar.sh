#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== default"; NO_WGRAD=1 python tools/conv_microbench.py l2g l3g l4g 2>&1 | grep -E "fwd|dgrad"
echo "== one barrier per two taps (timing only)"; FB_LIB_PATH=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_halfbar.so NO_WGRAD=1 python tools/conv_microbench.py l2g l3g l4g 2>&1 | grep -E "fwd|dgrad"
