#!/bin/bash
# copies the outputs of tools/gpu_round6_final.sh (gpurun_out/r6final, scratch) into profiles/ (tracked)
set -e
cd "$(dirname "$0")/.."
s=gpurun_out/r6final; d=profiles
cp $s/kernel_stats.md $d/r6_final_kernel_stats.md
cp $s/gradreg_kernel_stats.md $d/r6_final_gradreg_kernel_stats.md
cp $s/hbm_traffic_per_kernel.md $d/r6_final_hbm_traffic_per_kernel.md
cp $s/mfma_util.md $d/r6_final_mfma_util.md
cp $s/roofline_per_kernel.md $d/r6_final_roofline_per_kernel.md
cp $s/breakdown_bf16.md $d/r6_final_step_breakdown.md
cp $s/breakdown_r152.md $d/r6_r152_breakdown.md
cp $s/bench_default.json $d/r6_final_bench_line.json
cp $s/bench_r152_bf16.json $d/r6_final_bench_r152_bf16.json
cp $s/bench_r152_gradreg.json $d/r6_final_bench_r152_gradreg.json
cp $s/bench_r152_gradreg_f16x2.json $d/r6_final_bench_r152_gradreg_f16x2.json
cp $s/bench_exchange.json $d/r6_final_bench_exchange.json
cp $s/bench_detail_last.json $d/r6_final_bench_exchange_detail.json
cp $s/hbm_traffic.json $d/hbm_traffic.json
cp $s/mfma_util.json $d/mfma_util.json
cp $s/dispatch_r152_bf16.md $d/r6_dispatch_r152_bf16.md
cp $s/dispatch_r152_f32.md $d/r6_dispatch_r152_f32.md
cp $s/r152_gradreg_kernel_stats.md $d/r6_r152_gradreg_kernel_stats.md
cp $s/kernel_power.md $d/r6_kernel_power.md
