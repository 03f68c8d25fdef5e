#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2g
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv_fwd or conv_dgrad" > gpurun_out/r2g/pytest_ops.log 2>&1; tail -3 gpurun_out/r2g/pytest_ops.log
for cfg in default 64x3 128x3; do
  echo "== FB_IGEMM_CFG=$cfg"
  if [ $cfg = default ]; then unset FB_IGEMM_CFG; else export FB_IGEMM_CFG=$cfg; fi
  NO_WGRAD=1 python tools/conv_microbench.py d2 d3 d4 s2 s3 s4 l4g stemg 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r2g/igemm_cfg.txt 2>&1
cat gpurun_out/r2g/igemm_cfg.txt
export FB_IGEMM_CFG=64x3
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv_fwd or conv_dgrad" > gpurun_out/r2g/pytest_ops_64x3.log 2>&1; tail -3 gpurun_out/r2g/pytest_ops_64x3.log
export FB_IGEMM_CFG=128x3
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv_fwd or conv_dgrad" > gpurun_out/r2g/pytest_ops_128x3.log 2>&1; tail -3 gpurun_out/r2g/pytest_ops_128x3.log
unset FB_IGEMM_CFG
python -m pytest tests/test_gpu_engine.py tests/test_gpu_gradreg.py tests/test_gpu_training.py -m gpu -q -s > gpurun_out/r2g/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2g/pytest.log
grep -n "passed\|failed\|^FAILED" gpurun_out/r2g/pytest.log | tail -8; grep -n "\[torch.float32\] chunk" gpurun_out/r2g/pytest.log | head -4
timeout 600 python bench.py --grad-reg 0.5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2g/bench_gradreg.json 2> gpurun_out/r2g/bench_gradreg.err; python -c "
import json;d=json.loads(open('gpurun_out/r2g/bench_gradreg.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],{k:(v['ms_total'],v['tflops']) for k,v in d['roofline']['isolated'].items()})"
timeout 900 python bench.py --no-cpu-baseline --no-side-configs > gpurun_out/r2g/bench_default.json 2> gpurun_out/r2g/bench_default.err; head -c 420 gpurun_out/r2g/bench_default.json; echo
FB_WGRAD_STREAM=0 python tools/step_breakdown.py bf16 1 resnet152 standard 224 128 > gpurun_out/r2g/breakdown_r152.md 2>&1; head -5 gpurun_out/r2g/breakdown_r152.md; tail -14 gpurun_out/r2g/breakdown_r152.md
