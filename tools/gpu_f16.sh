#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/f16
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "fp16x2 or weight_prep" > gpurun_out/f16/ops.log 2>&1; tail -15 gpurun_out/f16/ops.log
timeout 1500 python -m pytest tests/test_gpu_gradreg.py tests/test_gpu_engine.py -q -x > gpurun_out/f16/engine.log 2>&1; tail -8 gpurun_out/f16/engine.log
for m in f16x2 bf16x6; do FB_F32_SPLIT=$m timeout 900 python bench.py --grad-reg 0.5 --steps 1 --warmup 1 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | head -c 300 | sed "s/^/$m /"; echo; done | tee gpurun_out/f16/ab.log
