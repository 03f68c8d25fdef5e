#!/bin/bash
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r5n; mkdir -p $out
V=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_pf0.so
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_structural.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -n 4 ) > $out/tests.log; tail -n 2 $out/tests.log
for v in "X=0" "FB_LIB_PATH=$V" "X=0" "FB_LIB_PATH=$V"; do echo "== $v"; ( env $v NO_WGRAD=1 timeout 300 python tools/conv_microbench.py l2g l3g l4g 2>&1 | grep -v amdgpu.ids ); done
r18() { timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['train_loss_last'])"; }
echo "r18 prefetch: $(r18)"; echo "r18 no prefetch: $(FB_LIB_PATH=$V r18)"; echo "r18 prefetch: $(r18)"; echo "r18 no prefetch: $(FB_LIB_PATH=$V r18)"
