import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fullbatchtraining_amd import lib
def bench(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
cin = cout = 512; ipg = 128; chunks = 98; n = ipg * chunks
x = torch.randn(n, 4, 4, cin, device="cuda").relu().to(torch.bfloat16)
dy = torch.randn(n, 4, 4, cout, device="cuda").to(torch.bfloat16)
per = torch.empty(chunks, cout, 9, cin, device="cuda")
t0 = bench(lambda: lib.conv2d_wgrad(x, dy, per, 3, 3, 1, 1, ipg, 1))
print(f"per-chunk: {t0:.1f} us")
a = lib.WgradArgs(x.data_ptr(), dy.data_ptr(), None, n, 4, 4, cin, 4, 4, cout, 3, 3, 1, 1, ipg, 1, lib.dtype_code(torch.bfloat16), 0)
tiles = 64
for chains in (4, 7, 8, 14, 16):
    slabs = torch.empty(chains, cout, 9, cin, device="cuda"); sqp = torch.empty(chunks, tiles, 8, device="cuda"); tot = torch.empty(cout, 9, cin, device="cuda")
    def run():
        lib.call("fb_conv2d_wgrad_chain", lib.C.byref(a), chains, slabs.data_ptr(), sqp.data_ptr())
        lib.wgrad_reduce(slabs, tot, 0, 1, chains, cout, 9, cin, cin)
    print(f"chained, {chains} chains ({tiles * chains} workgroups) + reduce: {bench(run):.1f} us")
