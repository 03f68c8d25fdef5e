#!/bin/bash
# tools/race_probe.py under the schedule / kernel switches that matter (run on the GPU box: gpurun -- bash tools/race_probe.sh)
cd "$GRAFT_REPO_ROOT"
for cfg in "FB_NONE=1" "FB_ACC_OVERLAP=0" "FB_REPLAY=0" "FB_DISABLE_HALO5=1" "FB_DISABLE_HALO5=1 FB_DISABLE_HALO4=1"; do
  echo "== $cfg"
  env $cfg timeout 900 python tools/race_probe.py ${1:-32} 8 3 bf16 2>&1 | grep -v amdgpu.ids | cut -c1-400 | grep -E "step . avg|runs differ|reference|Error|error" | sort | uniq -c | sort -rn | head -12
done
echo "== 16 chunks, groups of 5"
timeout 900 python tools/race_probe.py ${1:-32} 16 5 bf16 2>&1 | grep -E "runs differ|reference"
echo "== f32"
timeout 900 python tools/race_probe.py 16 8 3 f32 2>&1 | grep -E "runs differ|reference"
