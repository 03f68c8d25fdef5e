#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for g in 2 4 8 16 32 98; do echo "G=$g"; python tools/bn_bwd_microbench.py $g 2>&1 | grep "^| 64\|^| 128\|^| 256\|^| 512" | cut -d'|' -f2-4; done
