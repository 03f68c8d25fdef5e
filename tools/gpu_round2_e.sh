#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2e
python -m pytest tests -m gpu -q -s > gpurun_out/r2e/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2e/pytest.log
grep -n "passed\|failed\|^FAILED" gpurun_out/r2e/pytest.log | tail -12
grep -n "\[torch.float32\] chunk" gpurun_out/r2e/pytest.log | head -4
timeout 600 python bench.py --grad-reg 0.5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2e/bench_gradreg_split.json 2> gpurun_out/r2e/bench_gradreg_split.err; tail -c 300 gpurun_out/r2e/bench_gradreg_split.json
FB_BENCH_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2e/bench_selflaunch.json 2> gpurun_out/r2e/bench_selflaunch.err; echo "selflaunch rc=$?"; tail -c 300 gpurun_out/r2e/bench_selflaunch.json
