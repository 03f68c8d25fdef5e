# HBM traffic of the conv kernels of one microbench case:  bash tools/pmc_mem.sh <case>   -> gpurun_out/pmcmem_<case>/
# (one counter per pass: larger sets exceed the TCC counter capacity and rocprofv3 aborts)
case=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmcmem_$case; mkdir -p $out
for e in "$@"; do export $e; done
i=0
for c in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/p$i -o p$i -- python3 tools/conv_microbench.py $case > $out/p$i.txt 2>&1
done
python3 - <<PY
import csv, collections
for p in ("p1","p2"):
    try: rows=list(csv.DictReader(open(f"$out/{p}/{p}_counter_collection.csv")))
    except Exception as e: print(p, "missing", e); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows: agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if "conv" in k: print(p, k, {c: f"{sum(x)/len(x):.4g}" for c,x in v.items()}, "launches", len(next(iter(v.values()))))
PY
grep -v amdgpu.ids $out/p1.txt | grep "TF/s" | head -8
