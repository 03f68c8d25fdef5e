#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2n
python -m pytest tests/test_gpu_ops.py -m gpu -q -k "conv_fwd or conv_dgrad" > gpurun_out/r2n/pytest_ops.log 2>&1; tail -4 gpurun_out/r2n/pytest_ops.log | cut -c1-300
FB_H4_WIDE=8,16 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "conv_fwd or conv_dgrad" > gpurun_out/r2n/pytest_ops_wide.log 2>&1; tail -4 gpurun_out/r2n/pytest_ops_wide.log | cut -c1-300
NO_WGRAD=1 python tools/conv_microbench.py l2g l3g l4g 2>&1 | grep -v amdgpu.ids > gpurun_out/r2n/micro_default.txt; cat gpurun_out/r2n/micro_default.txt
FB_H4_WIDE=8,16 NO_WGRAD=1 python tools/conv_microbench.py l2g l3g 2>&1 | grep -v amdgpu.ids > gpurun_out/r2n/micro_wide.txt; cat gpurun_out/r2n/micro_wide.txt
FB_DISABLE_HALO4=1 NO_WGRAD=1 python tools/conv_microbench.py l4g 2>&1 | grep -v amdgpu.ids > gpurun_out/r2n/micro_igemm.txt; cat gpurun_out/r2n/micro_igemm.txt
