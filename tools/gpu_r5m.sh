#!/bin/bash
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r5m; mkdir -p $out
V=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_sp1.so
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gradreg.py tests/test_gpu_engine.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -n 5 ) > $out/tests.log; tail -n 3 $out/tests.log
for v in "X=0" "FB_LIB_PATH=$V"; do echo "== $v"; ( env $v DT=f32 NO_WGRAD=1 timeout 300 python tools/conv_microbench.py l1g l2g l3g 2>&1 | grep -v amdgpu.ids ); done
gr() { timeout 900 python bench.py --grad-reg 0.5 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['train_loss_last'])"; }
echo "gradreg bf16x6 cross-tap: $(gr)"; echo "gradreg bf16x6 weight-only: $(FB_LIB_PATH=$V gr)"; echo "gradreg bf16x6 cross-tap: $(gr)"
