#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_exch
FB_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_exch -o ex -- python3 bench.py --images 6272 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_exch/log.txt 2>&1
python3 tools/kernel_stats.py gpurun_out/prof_exch 13 "exchange" 2>/dev/null | grep -v -E "conv|bn_|wgrad|head_|stem|mt_accumulate|maxpool|avgpool" | head -40
