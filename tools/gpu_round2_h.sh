#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2h
python -m pytest tests -m gpu -q > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2h/pytest.log
grep -n "passed\|failed\|^FAILED" gpurun_out/r2h/pytest.log | tail -8
timeout 900 python bench.py > gpurun_out/r2h/bench_default.json 2> gpurun_out/r2h/bench_default.err; head -c 600 gpurun_out/r2h/bench_default.json; echo
bash tools/profile_bench.sh r2a > gpurun_out/r2h/profile.log 2>&1; tail -3 gpurun_out/r2h/profile.log | cut -c1-300
bash tools/pmc_bench.sh r2a > gpurun_out/r2h/pmc.log 2>&1; tail -12 gpurun_out/r2h/pmc.log | cut -c1-200
python tools/roofline_table.py gpurun_out/prof_r2a gpurun_out/pmcbench_r2a 3 > gpurun_out/r2h/roofline_per_kernel.md 2>&1; head -30 gpurun_out/r2h/roofline_per_kernel.md | cut -c1-200
find gpurun_out/prof_r2a gpurun_out/pmcbench_r2a -name "*kernel_trace*" -size +20M -delete
FB_WGRAD_STREAM=0 python tools/step_breakdown.py bf16 98 > gpurun_out/r2h/breakdown_bf16.md 2>&1
