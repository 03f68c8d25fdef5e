#!/bin/bash
# A/B of the epilogue prefetch depth of the resident-filter 64-channel convolution (FB_H5_ADD_DEPTH 3 / 5 / 7): the +add input gradients
cd "$GRAFT_REPO_ROOT"
for tag in default ad5 ad7; do
  V=""; [ $tag != default ] && V=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_$tag.so
  echo "== $tag"
  FB_LIB_PATH=$V FB_WGRAD_STREAM=0 python tools/step_breakdown.py bf16 98 2>/dev/null | grep -E "64->64 k3" | head -12
done
run() { timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for round in 1 2; do
  echo "round $round default: $(run)"
  echo "round $round ad5: $(FB_LIB_PATH=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_ad5.so run)"
  echo "round $round ad7: $(FB_LIB_PATH=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants/libfbengine_ad7.so run)"
done
