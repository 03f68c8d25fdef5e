#!/bin/bash
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r5f; mkdir -p $out
FB_WGRAD_STREAM=0 python tools/step_breakdown.py bf16 8 resnet152 standard 224 128 > $out/breakdown_r152.md 2>&1; tail -n 22 $out/breakdown_r152.md
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -n 8 ) > $out/pytest_gpu.log; tail -n 4 $out/pytest_gpu.log
