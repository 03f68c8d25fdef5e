#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== default"; ADD=1 NO_WGRAD=1 python tools/conv_microbench.py l1g 2>&1 | grep -E "fwd|dgrad"
echo "== whole-line stores (timing only)"; FB_LIB_PATH=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_h5line.so ADD=1 NO_WGRAD=1 python tools/conv_microbench.py l1g 2>&1 | grep -E "fwd|dgrad"
echo "== default"; ADD=1 NO_WGRAD=1 python tools/conv_microbench.py l1g 2>&1 | grep -E "fwd|dgrad"
