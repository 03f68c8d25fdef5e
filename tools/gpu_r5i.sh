#!/bin/bash
# config 3 (regulariser, bf16x6): split of the next weight fragment threaded through the MFMAs -- parity + step time A/B
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r5i; mkdir -p $out
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gradreg.py tests/test_gpu_engine.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -n 5 ) > $out/tests.log; tail -n 3 $out/tests.log
gr() { timeout 900 python bench.py --grad-reg 0.5 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['train_loss_last'])"; }
echo "gradreg bf16x6 default: $(gr)"
V=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants
[ -f $V/libfbengine_nopipe.so ] && echo "gradreg bf16x6 round-4 split order: $(FB_LIB_PATH=$V/libfbengine_nopipe.so gr)"
echo "gradreg bf16x6 default: $(gr)"
DT=f32 NO_WGRAD=1 python tools/conv_microbench.py l2g l3g l4g d3 2>&1 | grep -v amdgpu.ids
[ -f $V/libfbengine_nopipe.so ] && FB_LIB_PATH=$V/libfbengine_nopipe.so DT=f32 NO_WGRAD=1 python tools/conv_microbench.py l2g l3g l4g d3 2>&1 | grep -v amdgpu.ids
