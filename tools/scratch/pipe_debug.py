"""Where does the pipelined 1x1 kernel differ from the round-3 streaming kernel?  (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fullbatchtraining_amd import lib
lib.load()
for (cin, cout, hw, n, mode) in [(64, 128, 16, 3, 0), (128, 256, 8, 5, 0), (256, 512, 4, 16, 0), (256, 1024, 14, 2, 0), (64, 128, 16, 700, 0), (256, 128, 8, 5, 1), (64, 128, 16, 300, 1)]:
    torch.manual_seed(0)
    x = torch.randn(n, hw, hw, cin, device="cuda").bfloat16()
    w = (torch.randn(cout, 1, cin, device="cuda") * 0.1).bfloat16()
    outs = []
    for pipe in ("0", "1"):
        os.environ["FB_C1S_PIPE"] = pipe
        out = torch.zeros(n, hw, hw, cout, device="cuda", dtype=torch.bfloat16)
        stat = torch.zeros(2, (n * hw * hw + 127) // 128, cout, device="cuda")
        lib.conv2d(x, w, out, 1, 1, 1, 0, mode, stat_partial=stat if mode == 0 else None)
        torch.cuda.synchronize()
        outs.append((out.float().reshape(-1, cout), stat))
    (a, sa), (b, sb) = outs
    bad = (a != b)
    print(f"{cin}->{cout} {hw}x{hw} n={n} mode={mode}: M={a.shape[0]} differing elements {int(bad.sum())}, stat diff {float((sa - sb).abs().max()):.3g}")
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print("   rows:", rows[:12].tolist(), "...", rows[-4:].tolist(), "n_rows", len(rows), " cols:", cols[:8].tolist(), "...", cols[-4:].tolist(), "n_cols", len(cols))
        r = int(rows[0]); c = int(cols[0])
        print("   sample old/new:", a[r, c:c + 8].tolist(), b[r, c:c + 8].tolist())
