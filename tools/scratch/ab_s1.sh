#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "nontemporal stores"; NO_WGRAD=1 python tools/conv_microbench.py s2 s3 s4 2>&1 | grep -E "fwd|dgrad"
echo "plain stores"; FB_LIB_PATH=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_s1plain.so NO_WGRAD=1 python tools/conv_microbench.py s2 s3 s4 2>&1 | grep -E "fwd|dgrad"
