#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/mfma_accum_probe.hip && /tmp/mfma_probe > gpurun_out/r2d/mfma_probe.txt 2>&1; cat gpurun_out/r2d/mfma_probe.txt
FB_WGRAD_STREAM=0 python tools/step_breakdown.py bf16 98 > gpurun_out/r2d/breakdown_bf16.md 2>&1; tail -12 gpurun_out/r2d/breakdown_bf16.md
FB_WGRAD_STREAM=0 python tools/step_breakdown.py f32 49 > gpurun_out/r2d/breakdown_f32.md 2>&1; tail -12 gpurun_out/r2d/breakdown_f32.md
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r2d/bench_default.json 2> gpurun_out/r2d/bench_default.err; tail -c 2500 gpurun_out/r2d/bench_default.json; tail -3 gpurun_out/r2d/bench_default.err
