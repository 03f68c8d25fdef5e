#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2f
python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_sharded.py tests/test_gpu_training.py -m gpu -q -x > gpurun_out/r2f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f/pytest.log
grep -n "passed\|failed\|^FAILED" gpurun_out/r2f/pytest.log | tail -6
timeout 600 python bench.py --grad-reg 0.5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2f/bench_gradreg_split.json 2> gpurun_out/r2f/bench_gradreg_split.err; tail -c 400 gpurun_out/r2f/bench_gradreg_split.json; echo
timeout 600 python bench.py --no-cpu-baseline --no-side-configs > gpurun_out/r2f/bench_default.json 2> gpurun_out/r2f/bench_default.err; head -c 500 gpurun_out/r2f/bench_default.json; echo
FB_FUSED_POOL=0 timeout 600 python bench.py --no-cpu-baseline --no-side-configs > gpurun_out/r2f/bench_nofusedpool.json 2> gpurun_out/r2f/bench_nofusedpool.err; head -c 400 gpurun_out/r2f/bench_nofusedpool.json; echo
python tools/bn_mall_experiment.py > gpurun_out/r2f/bn_mall_nt.txt 2>&1; cat gpurun_out/r2f/bn_mall_nt.txt
FB_WGRAD_STREAM=0 python tools/step_breakdown.py bf16 1 resnet152 standard 224 128 > gpurun_out/r2f/breakdown_r152.md 2>&1; head -5 gpurun_out/r2f/breakdown_r152.md; tail -14 gpurun_out/r2f/breakdown_r152.md
FB_EXTRA_HIPCC_FLAGS="-DFB_BN_REDUCE_PLAIN" python -m fullbatchtraining_amd.build --force > gpurun_out/r2f/rebuild.log 2>&1
python tools/bn_mall_experiment.py > gpurun_out/r2f/bn_mall_plain.txt 2>&1; cat gpurun_out/r2f/bn_mall_plain.txt
