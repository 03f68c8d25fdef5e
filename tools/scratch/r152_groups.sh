#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for g in 8 12 16; do
  echo "== chunk_group $g"
  timeout 600 python bench.py --model resnet152 --stem standard --pixels 224 --images 6144 --chunk-group $g --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['config'])"
done
FB_WGRAD_STREAM=0 python tools/step_breakdown.py bf16 8 resnet152 standard 224 128 2>/dev/null | tail -60
