#!/bin/bash
# usage: ab_step.sh "<ENV=.. for B>"   -- A = default, B = with the given environment; two runs each, interleaved
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
echo "A: $(python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["ms_per_step"])')"
echo "B: $(env $1 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["ms_per_step"])')"
done
