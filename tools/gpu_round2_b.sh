#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_training.py tests/test_gpu_bf16_parity.py -m gpu -q -s --durations=10 > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
grep -n "passed\|failed\|bf16 vs f32\|momentum buffers" gpurun_out/r2b/pytest.log | tail -12
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2b/prof_f32 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --grad-reg 0.5 --steps 1 --warmup 1 --serialize --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2b/bench_f32_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2b/bench_f32_prof.err
cd $GRAFT_REPO_ROOT; python tools/kernel_stats.py gpurun_out/r2b/prof_f32 2 "rocprofv3 --kernel-trace --stats -- python3 bench.py --grad-reg 0.5 --steps 1 --warmup 1 --serialize --no-cpu-baseline" > gpurun_out/r2b/prof_f32.md 2>&1; head -40 gpurun_out/r2b/prof_f32.md
find gpurun_out/r2b/prof_f32 -name "*kernel_trace*" -delete
