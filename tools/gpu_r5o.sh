#!/bin/bash
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r5o; mkdir -p $out
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_gradreg.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -n 4 ) > $out/tests.log; tail -n 2 $out/tests.log
FB_WGRAD_STREAM=0 python tools/step_breakdown.py f32 4 resnet152 standard 224 128 > $out/breakdown_r152_f32.md 2>&1; grep -v amdgpu $out/breakdown_r152_f32.md | sed -n 1,2p; grep "wgrad 128->128 k3\|wgrad 64->64 k3\|wgrad 256->256 k3" $out/breakdown_r152_f32.md
timeout 900 python bench.py --model resnet152 --stem standard --pixels 224 --images 1024 --grad-reg 0.5 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | head -c 300; echo
