#!/bin/bash
# the pipelined 1x1 kernel in the models: ResNet-152 @224 A/B (same box), the engine / structural tests that run these shapes, the headline quick
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r5e; mkdir -p $out
( timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_structural.py tests/test_gpu_engine.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -n 6 ) > $out/tests.log; tail -n 3 $out/tests.log
r152() { timeout 900 python bench.py --model resnet152 --stem standard --pixels 224 --images 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
echo "r152 default: $(r152)"
echo "r152 FB_C1S_PIPE=0: $(FB_C1S_PIPE=0 r152)"
echo "r152 FB_C1S_PIPE=0 FB_C1S_ADD_ASM=0 (round 4): $(FB_C1S_PIPE=0 FB_C1S_ADD_ASM=0 r152)"
echo "r152 default again: $(r152)"
r18() { timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])"; }
echo "r18 default: $(r18)"; echo "r18 FB_C1S_PIPE=0: $(FB_C1S_PIPE=0 r18)"; echo "r18 default: $(r18)"
