#!/bin/bash
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r5l; mkdir -p $out
FB_WGRAD_STREAM=0 python tools/step_breakdown.py f32 4 resnet152 standard 224 128 > $out/breakdown_r152_f32.md 2>&1; grep -v amdgpu $out/breakdown_r152_f32.md | sed -n 1,3p; sed -n '/call shape/,$p' $out/breakdown_r152_f32.md | head -45
