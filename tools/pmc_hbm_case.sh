#!/bin/bash
# HBM bytes per launch (FETCH_SIZE x2, WRITE_SIZE; one counter per pass) of single convolution launches:
#   bash tools/pmc_hbm_case.sh "<cases>" [ENV=VAL ...]   -> gpurun_out/pmchbm_<first case>/
cases=$1; shift
tag=$(echo $cases | cut -d' ' -f1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmchbm_$tag; mkdir -p $out
for e in "$@"; do export $e; done
i=0
for c in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/p$i -o p$i -- python3 tools/conv_microbench.py $cases > $out/p$i.txt 2>&1
done
cat $out/p1.txt | grep -v amdgpu.ids
python3 - <<PY
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1","p2"):
    for r in csv.DictReader(open(f"$out/{p}/{p}_counter_collection.csv")):
        g = r.get("Grid_Size", "")
        agg[(r["Kernel_Name"][:70], g)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), v in agg.items():
    if "conv" not in k and "wgrad" not in k: continue
    f = v.get("FETCH_SIZE", [0]); w = v.get("WRITE_SIZE", [0])
    print(f"{k:70s} grid {g:>8s} n={len(f):3d} fetch {2e3*sum(f)/len(f)/1e6:8.1f} MB  write {1e3*sum(w)/len(w)/1e6:8.1f} MB")
PY
