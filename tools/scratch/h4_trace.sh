#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for args in "16 128 128 12544 0" "16 128 128 12544 1" "8 256 256 12544 0" "4 512 512 12544 0"; do tools/scratch/h4_trace.bin $args; done
