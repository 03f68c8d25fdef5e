"""What the chip sustains for pure writes / reads / copies (torch kernels), for comparison with the output-dominated 1x1 convolutions."""
import torch
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (411, 1644):
    x = torch.empty(mb * 1000 * 1000 // 2, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(x)
    s = t(lambda: x.zero_()); print(f"{mb} MB zero_: {mb / 1e3 / s:.0f} GB/s")
    s = t(lambda: x.fill_(1.5)); print(f"{mb} MB fill_: {mb / 1e3 / s:.0f} GB/s")
    s = t(lambda: y.copy_(x)); print(f"{mb} MB copy: {2 * mb / 1e3 / s:.0f} GB/s (read + write)")
    s = t(lambda: x.sum()); print(f"{mb} MB sum (read): {mb / 1e3 / s:.0f} GB/s")
