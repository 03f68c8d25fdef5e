#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "two_batchnorms or batchnorm_apply_in_its_loader" -p no:cacheprovider 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_bf16_structural.py tests/test_gpu_training.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
run() { timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['train_loss_last'])"; }
for round in 1 2; do
  echo "round $round dual: $(run)"
  echo "round $round separate (FB_BN_BWD_DUAL=0): $(FB_BN_BWD_DUAL=0 run)"
done
