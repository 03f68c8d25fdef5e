#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/bst
for rep in 1 2; do
for w in none 4 8 16 32 "4,8" "4,8,16"; do
  if [ "$w" = none ]; then export FB_FUSED_BWD_STAT=0; unset FB_FUSED_BWD_STAT_W; else export FB_FUSED_BWD_STAT=1 FB_FUSED_BWD_STAT_W="$w"; fi
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | python -c "import sys,json; print('w=$w', json.loads(sys.stdin.read())['ms_per_step'])"
done; done | tee gpurun_out/bst/ab_w.log
