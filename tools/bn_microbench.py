"""BN kernel microbenchmark (GPU box): achieved HBM bandwidth of fb_bn_apply / fb_bn_bwd_reduce / fb_bn_bwd_apply on the four
ResNet-18 map sizes, one chunk group (39 x 128 images, bf16).   python tools/bn_microbench.py [n_chunks]"""
import sys

import torch

sys.path.insert(0, ".")
from fullbatchtraining_amd import lib  # noqa: E402


def bench(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 39
    n = G * 128
    h = lib.load()
    for C, hw in ((64, 32), (128, 16), (256, 8), (512, 4)):
        px, ppg = n * hw * hw, 128 * hw * hw
        x = torch.randn(px, C, device="cuda").bfloat16()
        dout = torch.randn(px, C, device="cuda").bfloat16()
        y, dx, res = torch.empty_like(x), torch.empty_like(x), torch.randn_like(x)
        mask = torch.empty(px * C // 8, dtype=torch.uint8, device="cuda")
        scale, shift = torch.rand(G, C, device="cuda") + 0.5, torch.randn(G, C, device="cuda")
        mean, invstd = torch.randn(G, C, device="cuda") * 0.1, torch.rand(G, C, device="cuda") + 0.5
        coef = torch.randn(G, C, 3, device="cuda")
        part = torch.empty(h.fb_ws_bn_partial_floats(px, C), device="cuda")
        nbytes = px * C * 2
        dt = lib.dtype_code(torch.bfloat16)
        t = bench(lambda: lib.call("fb_bn_apply", x.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, None, px, C, ppg, 0, 1,
                                   mask.data_ptr(), None, 0, dt, None, None))
        print(f"C={C:4d} {hw:2d}x{hw:<2d} bn_apply        {t*1e6:8.1f} us  {2.0625*nbytes/t/1e12:5.2f} TB/s")
        t = bench(lambda: lib.call("fb_bn_apply", x.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), res.data_ptr(), None, None, px, C, ppg,
                                   0, 1, mask.data_ptr(), None, 0, dt, None, None))
        print(f"C={C:4d} {hw:2d}x{hw:<2d} bn_apply+res    {t*1e6:8.1f} us  {3.0625*nbytes/t/1e12:5.2f} TB/s")
        t = bench(lambda: lib.call("fb_bn_bwd_reduce", dout.data_ptr(), None, mask.data_ptr(), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), C, 0,
                                   part.data_ptr(), px, C, ppg, dt, None, None))
        print(f"C={C:4d} {hw:2d}x{hw:<2d} bn_bwd_reduce   {t*1e6:8.1f} us  {2.0625*nbytes/t/1e12:5.2f} TB/s")
        t = bench(lambda: lib.call("fb_bn_bwd_apply", dout.data_ptr(), None, mask.data_ptr(), x.data_ptr(), coef.data_ptr(), dx.data_ptr(), None, px, C,
                                   ppg, dt, None, None))
        print(f"C={C:4d} {hw:2d}x{hw:<2d} bn_bwd_apply    {t*1e6:8.1f} us  {3.0625*nbytes/t/1e12:5.2f} TB/s")
        t = bench(lambda: y.copy_(x))
        print(f"C={C:4d} {hw:2d}x{hw:<2d} torch copy      {t*1e6:8.1f} us  {2*nbytes/t/1e12:5.2f} TB/s")


if __name__ == "__main__":
    main()
