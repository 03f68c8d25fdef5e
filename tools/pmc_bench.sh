#!/bin/bash
# HBM traffic per kernel of the bench workload (one counter per pass, see profiles/r1_pmc_notes.md):
#   bash tools/pmc_bench.sh <tag>   -> gpurun_out/pmcbench_<tag>/summary.md
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcbench_$tag; mkdir -p $out
i=0
for c in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/p$i -o p$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side-configs --serialize > $out/p$i.txt 2>&1
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
# per-launch HBM bytes of the three MFMA kernel classes bench.py reports (copy to profiles/hbm_traffic.json: bench.py reads it for
# roofline.traffic).  Forward / input-gradient launches share kernels; one chunk group issues its 20 forward convolutions before
# its 19 input-gradient ones, so the class follows from the position in dispatch order.
python3 - <<PY > $out/hbm_traffic.json
import csv, json, collections
per = collections.defaultdict(lambda: [0.0, 0.0])          # dispatch id -> [fetch KB, write KB]
name = {}
for p, col in (("p1", 0), ("p2", 1)):
    for r in csv.DictReader(open(f"$out/{p}/{p}_counter_collection.csv")):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            per[int(r["Dispatch_Id"])][col] += float(r["Counter_Value"])
            name[int(r["Dispatch_Id"])] = r["Kernel_Name"]
cls = collections.defaultdict(list)
k = 0
for d in sorted(per):
    n = name[d]
    b = (2 * per[d][0] + per[d][1]) * 1e3                      # KB -> bytes, FETCH_SIZE doubled (MI355X_MICROARCH.md, gfx950)
    if n.startswith("void conv_wgrad") :
        cls["wgrad"].append(b)
    elif any(t in n for t in ("conv3x3s1_halo4", "conv3x3s1_c64_halo5", "conv_igemm_v3", "conv_igemm_kernel", "conv1x1_k32")):
        cls["igemm_fwd" if k % 39 < 20 else "igemm_dgrad"].append(b)
        if "halo5_kernel<0" in n: assert k % 39 < 20, (k, n)
        if "halo5_kernel<1" in n: assert k % 39 >= 20, (k, n)
        k += 1
print(json.dumps({"command": "tools/pmc_bench.sh $tag (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass, bench.py --steps 1 --warmup 0 --no-side-configs --serialize)",
                  "unit": "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)",
                  "classes": {c: {"launches": len(v), "bytes_per_launch": sum(v) / len(v)} for c, v in cls.items()}}, indent=1))
PY
cat $out/summary.md | cut -c1-200; cat $out/hbm_traffic.json
