"""Gaps between consecutive kernels of a rocprofv3 kernel trace (one stream): python tools/kernel_gaps.py <dir with *_kernel_trace.csv> [top]
gap before a launch = its start - the previous kernel's end (negative: the two overlapped).  Summed per kernel name, as "gap before" and "gap after"."""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:60]


def main():
    path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(path))), key=lambda r: r[0])
    before, after, dur, cnt = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
    end_max = rows[0][1]
    total_gap = 0
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        gap = s1 - max(e0, end_max)
        end_max = max(end_max, e0)
        if abs(gap) < 2_000_000:                  # (host-side pauses between steps are not kernel gaps)
            before[n1] += gap; after[n0] += gap; total_gap += gap
        dur[n0] += e0 - s0; cnt[n0] += 1
    span = rows[-1][1] - rows[0][0]
    print(f"{len(rows)} launches, span {span / 1e6:.1f} ms, kernel time {sum(dur.values()) / 1e6:.1f} ms, sum of gaps {total_gap / 1e6:.1f} ms")
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    print("| kernel | launches | time ms | gap before ms | gap after ms | (gap before + after) / launch us |\n|---|---|---|---|---|---|")
    for n, _ in sorted(dur.items(), key=lambda kv: -kv[1])[:top]:
        print(f"| `{n}` | {cnt[n]} | {dur[n] / 1e6:.1f} | {before[n] / 1e6:.2f} | {after[n] / 1e6:.2f} | {(before[n] + after[n]) / 1e3 / max(cnt[n], 1):.1f} |")


def by_predecessor(path, focus):
    """Average duration of the launches of kernels whose name contains ``focus``, split by the kernel that ran before them."""
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Grid_Size", "")) for r in csv.DictReader(open(path))), key=lambda r: r[0])
    acc = collections.defaultdict(list)
    for (s0, e0, n0, g0), (s1, e1, n1, g1) in zip(rows, rows[1:]):
        if focus in n1:
            acc[(n1, g1, n0)].append((e1 - s1) / 1e3)
    print(f"| kernel (grid) | after | launches | avg us |\n|---|---|---|---|")
    for (n1, g1, n0), v in sorted(acc.items(), key=lambda kv: -len(kv[1]))[:16]:
        print(f"| `{n1}` ({g1}) | `{n0}` | {len(v)} | {sum(v) / len(v):.1f} |")


if __name__ == "__main__":
    if len(sys.argv) > 3:
        by_predecessor(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0], sys.argv[3])
        sys.exit(0)
    main()
