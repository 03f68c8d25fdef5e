import re, collections, sys
rows = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for ln in open(sys.argv[1]):
    m = re.match(r"\| \d+ \| (.+?) \| (\d+) \| (\d+) \| (\d+) \|", ln)
    if m:
        r = rows[m.group(1)]; r[0] += 1; r[1] += int(m.group(2)); r[2] = int(m.group(3)); r[3] = int(m.group(4))
tot = sum(r[1] for r in rows.values())
print("total us", tot)
for k, r in sorted(rows.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print(f"{k:60s} x{r[0]:3d} {r[1] / 1000:7.1f} ms {100 * r[1] / tot:5.1f}%  {r[2]:5d} TF/s {r[3]:5d} GB/s")
