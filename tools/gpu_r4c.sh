#!/bin/bash
# round 4: sharded tests, the default bench line, and the 1-rank RCCL exchange under the last backward pass at several CU reserves
mkdir -p gpurun_out/r4c
timeout 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_ops.py -m gpu -q -x -k "rccl or checkpoint or chunk_clip or eval_coeffs or head_tta or shuffle" -p no:cacheprovider > gpurun_out/r4c/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r4c/pytest.log
( time python bench.py ) > gpurun_out/r4c/bench_default.json 2> gpurun_out/r4c/bench_default.err
cp gpurun_out/bench_detail.json gpurun_out/r4c/bench_detail_default.json
for r in 0 8 16 32; do
  FB_FORCE_DIST=1 FB_CU_RESERVE=$r python bench.py --images 6272 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/r4c/exchange_r$r.json 2> gpurun_out/r4c/exchange_r$r.err
done
python bench.py --images 6272 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-side-configs > gpurun_out/r4c/rank_share_plain.json 2> gpurun_out/r4c/rank_share_plain.err
