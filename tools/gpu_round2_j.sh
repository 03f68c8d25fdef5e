#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2j
python -m pytest tests -m gpu -q > gpurun_out/r2j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2j/pytest.log
grep -n "passed\|failed\|^FAILED" gpurun_out/r2j/pytest.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2j/smoke.log 2>&1; tail -3 gpurun_out/r2j/smoke.log
timeout 900 python bench.py > gpurun_out/r2j/bench_default.json 2> gpurun_out/r2j/bench_default.err; head -c 500 gpurun_out/r2j/bench_default.json; echo
timeout 600 python bench.py --chunk 125 --no-cpu-baseline --no-side-configs > gpurun_out/r2j/bench_k400.json 2> gpurun_out/r2j/bench_k400.err; head -c 400 gpurun_out/r2j/bench_k400.json; echo
FB_BENCH_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2j/bench_selflaunch.json 2> gpurun_out/r2j/bench_selflaunch.err; echo "selflaunch rc=$?"; head -c 300 gpurun_out/r2j/bench_selflaunch.json; echo
