#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2l
python -m pytest tests -m gpu -q > gpurun_out/r2l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2l/pytest.log
grep -n "passed\|failed\|^FAILED" gpurun_out/r2l/pytest.log | tail -8
timeout 900 python bench.py --no-cpu-baseline --no-side-configs > gpurun_out/r2l/bench_default.json 2> gpurun_out/r2l/bench_default.err; head -c 430 gpurun_out/r2l/bench_default.json; echo
FB_BENCH_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2l/bench_selflaunch.json 2> gpurun_out/r2l/bench_selflaunch.err; echo "selflaunch rc=$?"; tail -c 600 gpurun_out/r2l/bench_selflaunch.json | head -c 300; echo
