#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2i
python -m pytest tests/test_gpu_ops.py -m gpu -q -k "conv_fwd or conv_dgrad" > gpurun_out/r2i/pytest_ops.log 2>&1; tail -4 gpurun_out/r2i/pytest_ops.log | cut -c1-300
NO_WGRAD=1 python tools/conv_microbench.py d2 d3 d4 2>&1 | grep -v amdgpu.ids > gpurun_out/r2i/micro_new.txt; cat gpurun_out/r2i/micro_new.txt
FB_DISABLE_S2_QUAD=1 FB_DISABLE_S2_FWD=1 NO_WGRAD=1 python tools/conv_microbench.py d2 d3 d4 2>&1 | grep -v amdgpu.ids > gpurun_out/r2i/micro_old.txt; cat gpurun_out/r2i/micro_old.txt
python -m pytest tests/test_gpu_engine.py tests/test_gpu_training.py tests/test_gpu_bf16_parity.py -m gpu -q > gpurun_out/r2i/pytest.log 2>&1; tail -4 gpurun_out/r2i/pytest.log | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline --no-side-configs > gpurun_out/r2i/bench_default.json 2> gpurun_out/r2i/bench_default.err; head -c 420 gpurun_out/r2i/bench_default.json; echo
