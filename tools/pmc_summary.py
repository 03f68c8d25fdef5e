"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass as MI355X_MICROARCH.md prescribes):

    python tools/pmc_summary.py <dir with p1/p1_counter_collection.csv and p2/...> "<command description>" [steps]

writes <dir>/summary.md (table) and <dir>/hbm_traffic.json -- copy the latter to profiles/hbm_traffic.json: bench.py reads it for
`roofline.traffic` and the `hbm` object and refuses to run the headline configuration without it.
Corrections (MI355X_MICROARCH.md, HBM section): counter values are KB; FETCH_SIZE counts a 128-byte request as 64 B on gfx950 -> doubled.
"""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)                     # drop the argument list, keep template arguments
    name = name.replace("HIP_vector_type<unsigned int, 4u>", "uint4")
    return name[:96]


def main():
    out, command = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    agg = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    for p in ("p1", "p2"):
        path = os.path.join(out, p, f"{p}_counter_collection.csv")
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels, total = {}, 0.0
    for k, v in agg.items():
        f, w = v["FETCH_SIZE"], v["WRITE_SIZE"]
        n = max(len(f), len(w))
        if n == 0:
            continue
        fetch = 2e3 * sum(f) / max(len(f), 1)              # KB -> bytes, x2 (gfx950: 128-byte requests tallied as 64 B)
        write = 1e3 * sum(w) / max(len(w), 1)
        kernels[k] = {"launches": n / steps, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "bytes_per_launch": fetch + write}
        total += (fetch + write) * n / steps
    doc = {"command": command, "unit": "bytes (FETCH_SIZE x2 + WRITE_SIZE; per launch, and per step for the total)", "steps_profiled": steps,
           "total_bytes_per_step": total, "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"]))}
    if not kernels or total <= 0:
        sys.exit("pmc_summary: no FETCH_SIZE / WRITE_SIZE rows found -- not writing an empty hbm_traffic.json")
    with open(os.path.join(out, "hbm_traffic.json"), "w") as handle:
        json.dump(doc, handle, indent=1)
    with open(os.path.join(out, "summary.md"), "w") as handle:
        handle.write(f"`{command}`\n\n| kernel | launches / step | FETCH_SIZE x2 (MB / launch) | WRITE_SIZE (MB / launch) | GB / step |\n|---|---|---|---|---|\n")
        for k, v in doc["kernels"].items():
            gb = v["bytes_per_launch"] * v["launches"] / 1e9
            if gb > 0.05:
                handle.write(f"| `{k}` | {v['launches']:g} | {v['fetch_bytes_per_launch'] / 1e6:.1f} | {v['write_bytes_per_launch'] / 1e6:.1f} | {gb:.2f} |\n")
        handle.write(f"\ntotal {total / 1e9:.1f} GB per step\n")
    print(open(os.path.join(out, "summary.md")).read())


if __name__ == "__main__":
    main()
