#!/bin/bash
# same-box A/B of an environment switch: bash tools/ab_env.sh "FB_X=1" [rounds]  -> ms/step of the headline workload without / with it
cd "$GRAFT_REPO_ROOT"
run() { timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for round in $(seq 1 ${2:-2}); do
  echo "round $round default: $(run)"
  echo "round $round $1: $(env $1 bash -c "$(declare -f run); run")"
done
