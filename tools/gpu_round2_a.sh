#!/bin/bash
# first GPU pass of round 2: whole GPU test suite, bf16 parity probe, default bench, the K=400 and grad_reg variants
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout 900 python tools/bf16_parity_probe.py > gpurun_out/r2a/bf16_probe.log 2>&1; tail -3 gpurun_out/r2a/bf16_probe.log
timeout 600 python bench.py > gpurun_out/r2a/bench_default.json 2> gpurun_out/r2a/bench_default.err; tail -c 1500 gpurun_out/r2a/bench_default.json
timeout 600 python bench.py --chunk 125 --no-cpu-baseline > gpurun_out/r2a/bench_k400.json 2> gpurun_out/r2a/bench_k400.err; tail -c 600 gpurun_out/r2a/bench_k400.json
timeout 600 python bench.py --grad-reg 0.5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2a/bench_gradreg.json 2> gpurun_out/r2a/bench_gradreg.err; tail -c 600 gpurun_out/r2a/bench_gradreg.json
FB_BENCH_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2a/bench_selflaunch.json 2> gpurun_out/r2a/bench_selflaunch.err; echo "selflaunch rc=$?"; tail -c 400 gpurun_out/r2a/bench_selflaunch.json; tail -5 gpurun_out/r2a/bench_selflaunch.err
