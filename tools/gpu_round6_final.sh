#!/bin/bash
# final evidence run of round 6: driver-form bench line, rocprofv3 kernel stats (bf16 step and the regularised fp32 step), PMC HBM traffic,
# per-kernel roofline table, per-launch breakdowns (ResNet-18 headline group, ResNet-152 @224 group), config-5-shaped bench lines
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
out=gpurun_out/r6final; mkdir -p $out
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; head -c 400 $out/bench_default.json; echo
bash tools/profile_bench.sh r6final --no-kernel-timing > $out/profile.log 2>&1; tail -3 $out/profile.log | cut -c1-300
bash tools/pmc_bench.sh r6final > $out/pmc.log 2>&1; tail -6 $out/pmc.log | cut -c1-200
bash tools/pmc_mfma.sh r6final > $out/pmc_mfma.log 2>&1; tail -3 $out/pmc_mfma.log | cut -c1-300; cp gpurun_out/pmcmfma_r6final/summary.md $out/mfma_util.md; cp gpurun_out/pmcmfma_r6final/mfma_util.json $out/mfma_util.json
python tools/roofline_table.py gpurun_out/prof_r6final gpurun_out/pmcbench_r6final 3 > $out/roofline_per_kernel.md 2>&1; head -14 $out/roofline_per_kernel.md | cut -c1-200
FB_WGRAD_STREAM=0 python tools/step_breakdown.py bf16 98 > $out/breakdown_bf16.md 2>&1; tail -14 $out/breakdown_bf16.md
FB_WGRAD_STREAM=0 python tools/step_breakdown.py bf16 16 resnet152 standard 224 128 > $out/breakdown_r152.md 2>&1; tail -16 $out/breakdown_r152.md
mkdir -p gpurun_out/prof_r6final_gradreg
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r6final_gradreg -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --grad-reg 0.5 --steps 1 --warmup 1 --serialize --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof_r6final_gradreg/bench.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/kernel_stats.py gpurun_out/prof_r6final_gradreg 2 "rocprofv3 --kernel-trace --stats -- python3 bench.py --grad-reg 0.5 --steps 1 --warmup 1 --serialize --no-cpu-baseline --no-kernel-timing" > $out/gradreg_kernel_stats.md 2>&1; head -20 $out/gradreg_kernel_stats.md | cut -c1-160
cp gpurun_out/prof_r6final.md $out/kernel_stats.md; cp gpurun_out/pmcbench_r6final/summary.md $out/hbm_traffic_per_kernel.md; cp gpurun_out/pmcbench_r6final/hbm_traffic.json $out/hbm_traffic.json
rm -rf gpurun_out/prof_r6final gpurun_out/pmcbench_r6final/p1 gpurun_out/pmcbench_r6final/p2 gpurun_out/prof_r6final_gradreg
timeout 900 python bench.py --model resnet152 --stem standard --pixels 224 --images 2048 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_r152_bf16.json 2> $out/bench_r152_bf16.err; head -c 420 $out/bench_r152_bf16.json; echo
timeout 900 python bench.py --model resnet152 --stem standard --pixels 224 --images 1024 --grad-reg 0.5 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/bench_r152_gradreg.json 2> $out/bench_r152_gradreg.err; head -c 420 $out/bench_r152_gradreg.json; echo
FB_F32_SPLIT=f16x2 timeout 900 python bench.py --model resnet152 --stem standard --pixels 224 --images 1024 --grad-reg 0.5 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/bench_r152_gradreg_f16x2.json 2> $out/bench_r152_gradreg_f16x2.err; head -c 420 $out/bench_r152_gradreg_f16x2.json; echo
# the sharded step through RCCL with one rank (a rank's share of an 8-GPU job): where the late bucket's reduce-scatter runs relative to the backward pass
FB_FORCE_DIST=1 python bench.py --images 6272 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2> $out/bench_exchange.err | grep "^{" > $out/bench_exchange.json; head -c 300 $out/bench_exchange.json; echo
cp gpurun_out/bench_detail.json $out/bench_detail_last.json
# which kernel serves which launch of a rank's share of config 5 (2 chunks on 8 GPUs) against the one-GPU group
python tools/dispatch_table.py resnet152 standard 224 bf16 2 16 > $out/dispatch_r152_bf16.md 2>/dev/null
python tools/dispatch_table.py resnet152 standard 224 f32 1 8 > $out/dispatch_r152_f32.md 2>/dev/null
# kernel statistics of config 5 as BASELINE states it (with the regulariser, fp32 storage, bf16x6)
mkdir -p gpurun_out/prof_r6final_r152gr
A="--model resnet152 --stem standard --pixels 224 --images 1024 --grad-reg 0.5 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --serialize"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r6final_r152gr -o bench -- python3 $GRAFT_REPO_ROOT/bench.py $A > $GRAFT_REPO_ROOT/gpurun_out/prof_r6final_r152gr/bench.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/kernel_stats.py gpurun_out/prof_r6final_r152gr 2 "rocprofv3 --kernel-trace --stats -- python3 bench.py $A" > $out/r152_gradreg_kernel_stats.md 2>&1; head -12 $out/r152_gradreg_kernel_stats.md | cut -c1-160
rm -rf gpurun_out/prof_r6final_r152gr
python tools/kernel_power.py 1.0 > $out/kernel_power.md 2>/dev/null; head -8 $out/kernel_power.md
