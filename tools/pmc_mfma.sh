#!/bin/bash
# MFMA utilisation per kernel of the bench workload from hardware counters (one pass: SQ + GRBM blocks are independent):
#   bash tools/pmc_mfma.sh <tag> [extra bench.py flags]  -> gpurun_out/pmcmfma_<tag>/summary.md
# MfmaUtil (gfx94x formula, MI355X_MICROARCH.md) = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x CUs x 4 SIMDs)
tag=${1:-run}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcmfma_$tag; mkdir -p $out
cmd="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-side-configs --no-kernel-timing --serialize $*"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -o p1 -- $cmd > $out/p1.txt 2>&1
python3 tools/pmc_mfma_summary.py $out "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $cmd" 2 | cut -c1-220
