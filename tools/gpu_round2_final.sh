#!/bin/bash
# final evidence run of round 2: bench lines, rocprofv3 kernel stats (bf16 step and the regularised fp32 step), PMC HBM traffic, tables
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r2final
timeout 900 python bench.py > gpurun_out/r2final/bench_default.json 2> gpurun_out/r2final/bench_default.err; head -c 500 gpurun_out/r2final/bench_default.json; echo
bash tools/profile_bench.sh r2final > gpurun_out/r2final/profile.log 2>&1; tail -3 gpurun_out/r2final/profile.log | cut -c1-300
bash tools/pmc_bench.sh r2final > gpurun_out/r2final/pmc.log 2>&1; tail -6 gpurun_out/r2final/pmc.log | cut -c1-200
python tools/roofline_table.py gpurun_out/prof_r2final gpurun_out/pmcbench_r2final 3 > gpurun_out/r2final/roofline_per_kernel.md 2>&1; head -12 gpurun_out/r2final/roofline_per_kernel.md | cut -c1-200
FB_WGRAD_STREAM=0 python tools/step_breakdown.py bf16 98 > gpurun_out/r2final/breakdown_bf16.md 2>&1; tail -14 gpurun_out/r2final/breakdown_bf16.md
mkdir -p gpurun_out/prof_r2final_gradreg
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2final_gradreg -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --grad-reg 0.5 --steps 1 --warmup 1 --serialize --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r2final_gradreg/bench.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/kernel_stats.py gpurun_out/prof_r2final_gradreg 2 "rocprofv3 --kernel-trace --stats -- python3 bench.py --grad-reg 0.5 --steps 1 --warmup 1 --serialize --no-cpu-baseline" > gpurun_out/r2final/gradreg_kernel_stats.md 2>&1; head -24 gpurun_out/r2final/gradreg_kernel_stats.md | cut -c1-160
timeout 600 python bench.py --chunk 125 --no-cpu-baseline --no-side-configs > gpurun_out/r2final/bench_k400.json 2> gpurun_out/r2final/bench_k400.err; head -c 400 gpurun_out/r2final/bench_k400.json; echo
find gpurun_out/prof_r2final gpurun_out/pmcbench_r2final gpurun_out/prof_r2final_gradreg -name "*kernel_trace*" -size +8M -delete
timeout 900 python bench.py --model resnet152 --stem standard --pixels 224 --images 512 --grad-reg 0.5 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/r2final/bench_r152_gradreg.json 2> gpurun_out/r2final/bench_r152_gradreg.err; head -c 420 gpurun_out/r2final/bench_r152_gradreg.json; echo
FB_F32_SPLIT=f16x2 FB_WGRAD_STREAM=0 python tools/step_breakdown.py f32 49 > gpurun_out/r2final/breakdown_f32_f16x2.md 2>&1; tail -14 gpurun_out/r2final/breakdown_f32_f16x2.md
