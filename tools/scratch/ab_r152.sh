#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for e in "" "FB_C1S_NW=8" "FB_C1S_NW=4"; do
echo "[$e] $(env $e python bench.py --model resnet152 --stem standard --pixels 224 --images 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])')"
done
