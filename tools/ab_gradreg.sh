#!/bin/bash
# config 3 (regulariser, f16x2): schedule switches and chunk-group sizes, ms/step on one box
cd "$GRAFT_REPO_ROOT"
run() { timeout 900 python bench.py --grad-reg 0.5 --steps 2 --warmup 1 --no-cpu-baseline --no-side-configs --no-kernel-timing "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config'].get('chunk_group', ''))"; }
echo "default: $(run)"
echo "one stream: $(FB_WGRAD_STREAM=0 run)"
echo "replay off: $(FB_REPLAY=0 run)"
for g in 14 20 28 36 49; do echo "chunk-group $g: $(run --chunk-group $g)"; done
echo "default again: $(run)"
