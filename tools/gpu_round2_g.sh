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
