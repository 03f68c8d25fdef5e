#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/profile_bench.sh chain --no-kernel-timing > /dev/null 2>&1
FB_WGRAD_CHAIN=0 bash tools/profile_bench.sh nochain --no-kernel-timing > /dev/null 2>&1
for t in chain nochain; do echo "== $t"; grep -E "total kernel time|mt_accumulate|wgrad3x3_v2_kernel<4, 1|wgrad_reduce|reduce_kernel<at|elementwise|mt_finalize|CatArray|copy" gpurun_out/prof_$t.md | cut -c1-170; done
rm -rf gpurun_out/prof_chain gpurun_out/prof_nochain
