#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side-configs --no-kernel-timing "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config'].get('chunk_group',''))"; }
for g in 98 65 49 33 98; do echo "chunk-group $g: $(run --chunk-group $g)"; done
