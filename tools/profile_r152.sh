#!/bin/bash
# rocprofv3 kernel statistics of the ResNet-152 @224 step (BASELINE config 5's shape, one GPU):  bash tools/profile_r152.sh <tag>  -> gpurun_out/prof_<tag>.md
tag=${1:-r152}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_$tag
args="--model resnet152 --stem standard --pixels 224 --images 2048 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --serialize"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python3 bench.py $args > gpurun_out/prof_$tag/bench.log 2>&1
python3 tools/kernel_stats.py gpurun_out/prof_$tag 2 "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py $args" > gpurun_out/prof_$tag.md
head -40 gpurun_out/prof_$tag.md | cut -c1-220
rm -rf gpurun_out/prof_$tag
