#!/bin/bash
# Kernel-trace profile of the default bench workload; run on the GPU box:  bash tools/profile_bench.sh <tag> [bench args]
# Writes gpurun_out/prof_<tag>/ (raw csv) and gpurun_out/prof_<tag>.md (summary to copy into profiles/).
tag=${1:-run}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-configs --serialize "$@" > gpurun_out/prof_$tag/bench.log 2>&1
python3 tools/kernel_stats.py gpurun_out/prof_$tag 3 "rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-configs --serialize $*" > gpurun_out/prof_$tag.md
tail -2 gpurun_out/prof_$tag/bench.log | cut -c1-400
head -45 gpurun_out/prof_$tag.md | cut -c1-200
