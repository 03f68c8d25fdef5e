#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/f16
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "batchnorm or fp16x2 or pool" > gpurun_out/f16/ops.log 2>&1; tail -4 gpurun_out/f16/ops.log
for m in f16x2; do FB_F32_SPLIT=$m timeout 900 python bench.py --grad-reg 0.5 --steps 1 --warmup 1 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | head -c 300 | sed "s/^/$m /"; echo; done | tee gpurun_out/f16/ab.log
FB_F32_SPLIT=f16x2 FB_WGRAD_STREAM=0 timeout 600 python tools/step_breakdown.py f32 49 > gpurun_out/f16/breakdown_f32b.md 2>&1; tail -14 gpurun_out/f16/breakdown_f32b.md
timeout 2400 python -m pytest tests/test_gpu_gradreg.py tests/test_gpu_engine.py tests/test_gpu_sharded.py -q > gpurun_out/f16/engine.log 2>&1; grep -n "passed\|failed\|^FAILED" gpurun_out/f16/engine.log | tail -12
