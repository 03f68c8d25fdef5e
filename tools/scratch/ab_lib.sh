#!/bin/bash
# usage: ab_lib.sh <variant tag>   -- A = default library, B = csrc/variants/libfbengine_<tag>.so; two runs each, interleaved
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
echo "A: $(python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["ms_per_step"])')"
echo "B: $(FB_LIB_PATH=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_$1.so python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["ms_per_step"])')"
done
