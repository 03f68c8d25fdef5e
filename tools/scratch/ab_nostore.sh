#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== default"; SPLITS=1 python tools/conv_microbench.py l2g l3g l4g 2>&1 | grep -E "wgrad"
echo "== no slab stores (timing only)"; FB_LIB_PATH=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_nostore.so SPLITS=1 python tools/conv_microbench.py l2g l3g l4g 2>&1 | grep -E "wgrad"
bash tools/scratch/ab_lib.sh nostore
