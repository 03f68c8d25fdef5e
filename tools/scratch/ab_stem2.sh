#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "nontemporal stores"; NO_WGRAD=1 python tools/conv_microbench.py stemg 2>&1 | grep fwd
echo "plain stores"; FB_LIB_PATH=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_k32plain.so NO_WGRAD=1 python tools/conv_microbench.py stemg 2>&1 | grep fwd
