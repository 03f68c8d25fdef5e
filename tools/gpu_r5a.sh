#!/bin/bash
# round 5, first GPU call: the new tests + where the Bottleneck 1x1 shapes stand (microbench at the in-step group size, HBM bytes, SQ counters)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5a
( timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_bf16_parity.py -m gpu -x -q -s -k "chunk_gradients_vs_oracle or command_list or bench_mean_gradient" 2>&1 | grep -v amdgpu.ids | tail -n 40 ) > gpurun_out/r5a/tests1.log
( timeout 600 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q -k "chained or rccl_collectives_one_rank or two_ranks" 2>&1 | grep -v amdgpu.ids | tail -n 15 ) > gpurun_out/r5a/tests2.log
( IMGS=1024 ADD=1 NO_WGRAD=1 timeout 300 python tools/conv_microbench.py b1a b1b b2a b2b b3a b3b b4a b4b c1 c2 c3 c4 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5a/micro.log
for c in b3a b3b; do
  ( IMGS=1024 ADD=1 NO_WGRAD=1 bash tools/pmc_mem.sh $c 2>&1 | grep -v amdgpu.ids | tail -n 12 ) > gpurun_out/r5a/mem_$c.log
  ( IMGS=1024 ADD=1 NO_WGRAD=1 bash tools/pmc_case.sh $c 2>&1 | grep -v amdgpu.ids | tail -n 12 ) > gpurun_out/r5a/sq_$c.log
done
tail -n 5 gpurun_out/r5a/tests1.log gpurun_out/r5a/tests2.log; cat gpurun_out/r5a/micro.log
