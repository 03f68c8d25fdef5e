#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2c
python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_gradreg.py tests/test_gpu_training.py tests/test_gpu_sharded.py -m gpu -q -s > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/pytest.log
grep -n "passed\|failed\|^FAILED" gpurun_out/r2c/pytest.log | tail -12
timeout 600 python bench.py --grad-reg 0.5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2c/bench_gradreg_split.json 2> gpurun_out/r2c/bench_gradreg_split.err; tail -c 900 gpurun_out/r2c/bench_gradreg_split.json
FB_F32_EXACT=1 timeout 600 python bench.py --grad-reg 0.5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2c/bench_gradreg_exact.json 2> gpurun_out/r2c/bench_gradreg_exact.err; tail -c 300 gpurun_out/r2c/bench_gradreg_exact.json
timeout 600 python bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2c/bench_f32_split.json 2> gpurun_out/r2c/bench_f32_split.err; tail -c 300 gpurun_out/r2c/bench_f32_split.json
