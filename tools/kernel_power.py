"""Board power, shader clock and energy per launch of single kernels (GPU box):  python tools/kernel_power.py [seconds per case]

Every case loops ONE kernel of the ResNet-18 step at the benchmark's group size (98 chunks of 128 images, bf16) for ~1.5 s while a host thread reads the
amdgpu hwmon files of this GPU (bench.PowerSampler).  Columns: us per launch, average board power, shader clock, joules per launch -- and what the step's
launches of that kernel add up to.  The board idles at ~250 W; the cap is 1400 W.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PowerSampler  # noqa: E402
from fullbatchtraining_amd import lib  # noqa: E402


def run_case(name, fn, seconds, per_step):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 5 * 1e3
    reps = max(10, int(seconds * 1e6 / us))
    ps = PowerSampler(0)
    time.sleep(0.3)
    ps.start()
    t0 = time.perf_counter()
    done = 0
    while done < reps:                       # (queued in slices so that the host never runs seconds ahead of the device)
        for _ in range(min(50, reps - done)):
            fn()
        done += min(50, reps - done)
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    r = ps.result()
    us_loaded = el / reps * 1e6
    if r is None:
        print(f"| {name} | {us:.0f} | {us_loaded:.0f} | -- | -- | -- | -- |")
        return
    j = r["avg_w"] * us_loaded * 1e-6
    print(f"| {name} | {us:.0f} | {us_loaded:.0f} | {r['avg_w']} | {r['sclk_mhz']} | {j:.3f} | {per_step} x = {j * per_step:.1f} J |", flush=True)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
    G = 98
    n = G * 128
    h = lib.load()
    dt = lib.dtype_code(torch.bfloat16)
    print("| kernel (98 chunks) | us alone | us in the loop | W | sclk MHz | J / launch | launches per step (4 groups) |\n|---|---|---|---|---|---|---|")
    ps = PowerSampler(0); ps.start(); time.sleep(1.5); r = ps.result()
    print(f"| (idle) | | | {r['avg_w'] if r else '--'} | {r['sclk_mhz'] if r else '--'} | | |")
    for C, hw, n33 in ((64, 32, 4), (128, 16, 3), (256, 8, 3), (512, 4, 3)):
        px, ppg = n * hw * hw, 128 * hw * hw
        x = torch.randn(n, hw, hw, C, device="cuda").bfloat16()
        dy = torch.randn(n, hw, hw, C, device="cuda").bfloat16()
        y, dx = torch.empty_like(x), torch.empty_like(x)
        w = (torch.randn(C, 9, C, device="cuda") * 0.05).bfloat16()
        stat = torch.zeros(2, (px + 127) // 128, C, device="cuda")
        run_case(f"3x3 forward {C}->{C} @{hw}x{hw} (+stats)", lambda: lib.conv2d(x, w, y, 3, 3, 1, 1, 0, stat_partial=stat), seconds, 4 * n33)
        run_case(f"3x3 input gradient {C}->{C} @{hw}x{hw}", lambda: lib.conv2d(dy, w, dx, 3, 3, 1, 1, 1), seconds, 4 * n33)
        split = {64: 32, 128: 8, 256: 1, 512: 1}[C]
        slab = torch.empty(G * split * C * 9 * C, device="cuda")
        run_case(f"3x3 weight gradient {C}->{C} @{hw}x{hw}", lambda: lib.conv2d_wgrad(x, dy, slab, 3, 3, 1, 1, 128, split), seconds, 4 * n33)
        del slab
        mask = torch.empty(px * C // 8, dtype=torch.uint8, device="cuda")
        scale, shift = torch.rand(G, C, device="cuda") + 0.5, torch.randn(G, C, device="cuda")
        mean, invstd = torch.randn(G, C, device="cuda") * 0.1, torch.rand(G, C, device="cuda") + 0.5
        coef = torch.randn(G, C, 3, device="cuda")
        part = torch.empty(h.fb_ws_bn_partial_floats(px, C), device="cuda")
        run_case(f"bn_apply C{C} @{hw}x{hw}", lambda: lib.call("fb_bn_apply", x.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, None, px, C, ppg, 0, 1,
                                                               mask.data_ptr(), None, 0, dt, None, None), seconds, 4 * (n33 + 1))
        run_case(f"bn_bwd_reduce C{C} @{hw}x{hw}", lambda: lib.call("fb_bn_bwd_reduce", dy.data_ptr(), None, mask.data_ptr(), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), C, 0,
                                                                    part.data_ptr(), px, C, ppg, dt, None, None), seconds, 4 * (n33 + 1))
        run_case(f"bn_bwd_apply C{C} @{hw}x{hw}", lambda: lib.call("fb_bn_bwd_apply", dy.data_ptr(), None, mask.data_ptr(), x.data_ptr(), coef.data_ptr(), dx.data_ptr(), None, px, C,
                                                                   ppg, dt, None, None), seconds, 4 * (n33 + 1))
        run_case(f"torch copy_ of that tensor", lambda: y.copy_(x), seconds, 0)
        del x, dy, y, dx, mask, part
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
