#!/bin/bash
# same-box A/B of a variant build (tools/build_variant.py <tag> ...): per-launch microbenchmarks, parity tests of the variant, ms/step
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_$tag.so
echo "== microbench default"; python tools/conv_microbench.py l2g l3g l4g 2>&1 | grep -v amdgpu
echo "== microbench $tag"; FB_LIB_PATH=$V python tools/conv_microbench.py l2g l3g l4g 2>&1 | grep -v amdgpu
echo "== parity tests with $tag"; FB_LIB_PATH=$V python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_structural.py -m gpu -x -q -k "conv or structural" 2>&1 | tail -n 3
run() { timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for round in 1 2; do
  echo "round $round default: $(run)"
  echo "round $round $tag: $(FB_LIB_PATH=$V run)"
done
