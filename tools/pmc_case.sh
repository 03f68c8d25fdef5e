# usage: bash tools/pmc_case.sh <case> [ENV=VAL ...]   -> gpurun_out/pmc_<case>/
case=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$case; mkdir -p $out
for e in "$@"; do export $e; done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/p1 -o p1 -- python3 tools/conv_microbench.py $case > $out/p1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $out/p2 -o p2 -- python3 tools/conv_microbench.py $case > $out/p2.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out/p3 -o p3 -- python3 tools/conv_microbench.py $case > $out/p3.txt 2>&1
python3 - <<PY
import csv, collections, glob
for p in ("p1","p2","p3"):
    try: rows=list(csv.DictReader(open(f"$out/{p}/{p}_counter_collection.csv")))
    except Exception as e: print(p, "missing", e); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows: agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if "conv" in k: print(p, k, {c: f"{sum(x)/len(x):.3g}" for c,x in v.items()})
PY
