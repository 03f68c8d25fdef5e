#!/bin/bash
# HBM traffic per kernel of the bench workload (one counter per pass, see profiles/r1_pmc_notes.md):
#   bash tools/pmc_bench.sh <tag>   -> gpurun_out/pmcbench_<tag>/summary.md
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcbench_$tag; mkdir -p $out
i=0
for c in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/p$i -o p$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --serialize > $out/p$i.txt 2>&1
done
python3 - <<PY > $out/summary.md
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2"):
    try: rows = list(csv.DictReader(open(f"$out/{p}/{p}_counter_collection.csv")))
    except Exception as e: print(p, "missing", e); continue
    for r in rows: agg[r["Kernel_Name"][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| kernel | launches | FETCH_SIZE x2 (MB/launch) | WRITE_SIZE (MB/launch) | total GB over the run |\\n|---|---|---|---|---|")
tot = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -(2 * sum(kv[1].get("FETCH_SIZE", [0])) + sum(kv[1].get("WRITE_SIZE", [0])))):
    f, w = v.get("FETCH_SIZE", [0]), v.get("WRITE_SIZE", [0])
    n = max(len(f), len(w))
    gb = (2 * sum(f) + sum(w)) / 1e6
    tot += gb
    if gb > 0.5: print(f"| \`{k}\` | {n} | {2 * sum(f) / max(len(f), 1) / 1e3:.1f} | {sum(w) / max(len(w), 1) / 1e3:.1f} | {gb:.1f} |")
print(f"\\ntotal {tot:.1f} GB (FETCH_SIZE doubled per MI355X_MICROARCH.md: 128-byte requests are tallied as 64 B on gfx950; KB units)")
PY
cat $out/summary.md | cut -c1-200
