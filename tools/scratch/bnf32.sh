#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "two_batchnorms or batchnorm_apply_in_its_loader" -p no:cacheprovider 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_training.py tests/test_gpu_gradreg.py tests/test_gpu_sharded.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
run() { timeout 900 python bench.py --grad-reg 0.5 --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-side-configs --no-kernel-timing 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['train_loss_last'])"; }
echo "config 3 (bf16x6), both fusions: $(run)"
echo "config 3 (bf16x6), FB_WGRAD_BNF=0 FB_BN_BWD_DUAL=0: $(FB_WGRAD_BNF=0 FB_BN_BWD_DUAL=0 run)"
