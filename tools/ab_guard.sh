#!/bin/bash
# same-box A/B of the store-guard variants (tools/build_variant.py): ms/step of the headline workload, two rounds
cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/fullbatchtraining_amd/csrc/variants
for round in 1 2; do
  for tag in default noguard g1nomem g3nomem; do
    if [ $tag = default ]; then unset FB_LIB_PATH; else export FB_LIB_PATH=$V/libfbengine_$tag.so; fi
    ms=$(timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "round $round $tag: $ms ms/step"
  done
done
