"""Aggregate a rocprofv3 --kernel-trace csv into a markdown table:  python tools/kernel_stats.py <dir> <n_steps> "<command>"."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, n_steps, cmd = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit(f"no *kernel_trace.csv under {d}")
    agg = defaultdict(list)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                agg[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in agg.values()) / 1e3
    print(f"Command: `{cmd}`\n")
    log = os.path.join(d, "bench.log")
    if os.path.isfile(log):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            print("bench line of the same run: " + lines[-1].strip() + "\n")
    print(f"total kernel time {total:.1f} ms over {n_steps} steps (warm-up included)\n")
    print("| kernel | calls | total ms | ms/step | % | avg us | min us | max us |\n|---|---|---|---|---|---|---|---|")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        t = sum(v) / 1e3
        if t / total < 0.0005:
            continue
        print(f"| `{name[:100]}` | {len(v)} | {t:.1f} | {t / n_steps:.1f} | {100 * t / total:.1f} | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} |")


if __name__ == "__main__":
    main()
