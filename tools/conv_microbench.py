"""Micro-benchmark of single conv launches (HIP events) for kernel tuning.  Usage: python tools/conv_microbench.py [case ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fullbatchtraining_amd import lib

CASES = {  # name: (cin, cout, k, stride, hw, n_img)
    "l1": (64, 64, 3, 1, 32, 1664), "l2": (128, 128, 3, 1, 16, 1664), "l3": (256, 256, 3, 1, 8, 1664), "l4": (512, 512, 3, 1, 4, 1664),
    "l2s": (64, 128, 3, 2, 32, 1664), "stem": (32, 64, 1, 1, 32, 1664),
    # the downsampling blocks at the benchmark's group size (98 chunks): stride-2 3x3 and the 1x1 shortcut (on the pooled input)
    "d2": (64, 128, 3, 2, 32, 12544), "d3": (128, 256, 3, 2, 16, 12544), "d4": (256, 512, 3, 2, 8, 12544),
    "s2": (64, 128, 1, 1, 16, 12544), "s3": (128, 256, 1, 1, 8, 12544), "s4": (256, 512, 1, 1, 4, 12544),
    "l1g": (64, 64, 3, 1, 32, 12544), "l2g": (128, 128, 3, 1, 16, 12544), "l3g": (256, 256, 3, 1, 8, 12544), "l4g": (512, 512, 3, 1, 4, 12544), "stemg": (32, 64, 1, 1, 32, 12544),
    # ResNet-152 @224, one chunk group of 4 chunks (512 images): the 1x1 convolutions of the Bottleneck stages and their 3x3s
    "b1a": (64, 256, 1, 1, 56, 512), "b1b": (256, 64, 1, 1, 56, 512), "b2a": (128, 512, 1, 1, 28, 512), "b2b": (512, 128, 1, 1, 28, 512),
    "b3a": (256, 1024, 1, 1, 14, 512), "b3b": (1024, 256, 1, 1, 14, 512), "b4a": (512, 2048, 1, 1, 7, 512), "b4b": (2048, 512, 1, 1, 7, 512),
    "c1": (64, 64, 3, 1, 56, 512), "c2": (128, 128, 3, 1, 28, 512), "c3": (256, 256, 3, 1, 14, 512), "c4": (512, 512, 3, 1, 7, 512),
    "stemI": (160, 64, 1, 1, 112, 512),              # the ImageNet stem on its pre-gathered 7x7x3 patches (147 -> 160 channels), half a ResNet-152 chunk group
    "l1big": (64, 64, 3, 1, 32, 3840), "l2big": (128, 128, 3, 1, 16, 3840), "l3big": (256, 256, 3, 1, 8, 3840), "l4big": (512, 512, 3, 1, 4, 3840),
}


def bench(fn, iters=10):
    # SUSTAIN=<seconds>: loop that long before timing -- a 10-launch loop runs at boost clock; the step does not (the board sits at its power cap:
    # profiles/r5_kernel_power.md), and a kernel's sustained time is what it costs there
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    sustain = float(os.environ.get("SUSTAIN", "0"))
    if sustain > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < sustain:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        iters = 200
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    names = sys.argv[1:] or list(CASES)
    dtype = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32
    for name in names:
        cin, cout, k, stride, hw, n = CASES[name]
        n = int(os.environ.get("IMGS", n))                           # (IMGS=1024: the chunk group of the ResNet-152 step)
        pad = k // 2
        ho = (hw + 2 * pad - k) // stride + 1
        x = torch.randn(n, hw, hw, cin, device="cuda").to(dtype)
        w = (torch.randn(cout, k * k, cin, device="cuda") * 0.05).to(dtype)
        wt = (torch.randn(cin, k * k, cout, device="cuda") * 0.05).to(dtype)
        y = torch.empty(n, ho, ho, cout, device="cuda", dtype=dtype)
        dy = torch.randn(n, ho, ho, cout, device="cuda").to(dtype)
        dx = torch.empty(n, hw, hw, cin, device="cuda", dtype=dtype)
        stat = torch.zeros(2, (n * ho * ho + 127) // 128, cout, device="cuda")
        flops = 2 * n * ho * ho * cout * k * k * cin
        eb = 2 if dtype == torch.bfloat16 else 4
        byts = (x.numel() + y.numel()) * eb
        t = bench(lambda: lib.conv2d(x, w, y, k, k, stride, pad, 0, stat_partial=stat))
        print(f"{name:5s} fwd   {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s  {byts / t / 1e3:7.1f} GB/s", flush=True)
        t = bench(lambda: lib.conv2d(dy, wt, dx, k, k, stride, pad, 1))
        print(f"{name:5s} dgrad {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s  {byts / t / 1e3:7.1f} GB/s", flush=True)
        if os.environ.get("ADD"):                                  # the residual form: dst = dgrad + addend (same shape)
            add = torch.randn(n, hw, hw, cin, device="cuda").to(dtype)
            t = bench(lambda: lib.conv2d(dy, wt, dx, k, k, stride, pad, 1, addend=add, addend_mode=1))
            print(f"{name:5s} dgrad+add {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s  {(byts + add.numel() * eb) / t / 1e3:7.1f} GB/s", flush=True)
        ipg = 128
        if os.environ.get("NO_WGRAD"):
            continue
        for split in (tuple(int(v) for v in os.environ["SPLITS"].split(",")) if os.environ.get("SPLITS") else ((32, 8, 1) if k == 3 and stride == 1 else (8, 1))):
            slab = torch.empty(n // ipg * split * cout * k * k * cin, device="cuda")
            try:
                t = bench(lambda: lib.conv2d_wgrad(x, dy, slab, k, k, stride, pad, ipg, split))
                print(f"{name:5s} wgrad split={split:3d} {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s")
            except Exception as e:
                print(name, "wgrad", split, "failed", e)


if __name__ == "__main__":
    main()
