#!/bin/bash
# HBM traffic per kernel of the bench workload (one counter per pass, see profiles/r1_pmc_notes.md):
#   bash tools/pmc_bench.sh <tag> [extra bench.py flags]  -> gpurun_out/pmcbench_<tag>/{summary.md,hbm_traffic.json}
tag=${1:-run}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcbench_$tag; mkdir -p $out
cmd="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-side-configs --no-kernel-timing --serialize $*"
i=0
for c in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/p$i -o p$i -- $cmd > $out/p$i.txt 2>&1
done
# (the warm-up step is profiled too: 2 steps in the trace)
python3 tools/pmc_summary.py $out "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (one counter per pass) -- $cmd" 2 | cut -c1-220
