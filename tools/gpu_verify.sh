#!/bin/bash
# verification run: whole GPU suite, smoke, default bench, config-5-shaped measurements
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/verify
python -m pytest tests -m gpu -q > gpurun_out/verify/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/verify/pytest.log
grep -n "passed\|failed\|^FAILED" gpurun_out/verify/pytest.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/verify/smoke.log 2>&1; tail -2 gpurun_out/verify/smoke.log
timeout 900 python bench.py --no-cpu-baseline --no-side-configs > gpurun_out/verify/bench_default.json 2> gpurun_out/verify/bench_default.err; head -c 420 gpurun_out/verify/bench_default.json; echo
timeout 900 python bench.py --model resnet152 --stem standard --pixels 224 --images 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/verify/bench_r152_bf16.json 2> gpurun_out/verify/bench_r152_bf16.err; head -c 600 gpurun_out/verify/bench_r152_bf16.json; echo; tail -2 gpurun_out/verify/bench_r152_bf16.err | cut -c1-300
timeout 1200 python bench.py --model resnet152 --stem standard --pixels 224 --images 512 --grad-reg 0.5 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/verify/bench_r152_gradreg.json 2> gpurun_out/verify/bench_r152_gradreg.err; head -c 600 gpurun_out/verify/bench_r152_gradreg.json; echo; tail -2 gpurun_out/verify/bench_r152_gradreg.err | cut -c1-300
