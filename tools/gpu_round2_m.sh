#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2m
python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_gradreg.py -m gpu -q > gpurun_out/r2m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2m/pytest.log
grep -n "passed\|failed\|^FAILED" gpurun_out/r2m/pytest.log | tail -8
timeout 600 python bench.py --grad-reg 0.5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2m/bench_gradreg.json 2> gpurun_out/r2m/bench_gradreg.err; python -c "
import json;d=json.loads(open('gpurun_out/r2m/bench_gradreg.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],{k:(v['ms_total'],v['tflops']) for k,v in d['roofline']['isolated'].items()})"
python -m pytest tests/test_gpu_training.py -m gpu -q -k "gradreg or central or legacy or acc" > gpurun_out/r2m/pytest_train.log 2>&1; tail -3 gpurun_out/r2m/pytest_train.log | cut -c1-200
