"""Do an MFMA-bound convolution and an HBM-bound BatchNorm pass overlap when they run on two streams?  Times N launches of each alone and
both together (HIP events around the whole region).  GPU box: python tools/overlap_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fullbatchtraining_amd import lib

N_IMG = 12544
dt = torch.bfloat16


def conv_case(cin, cout, hw, mode=0):
    x = torch.randn(N_IMG, hw, hw, cin, device="cuda").to(dt)
    w = (torch.randn(cout, 9, cin, device="cuda") * 0.05).to(dt)
    y = torch.empty(N_IMG, hw, hw, cout, device="cuda", dtype=dt)
    stat = torch.zeros(2, N_IMG * hw * hw // 128, cout, device="cuda")
    return lambda: lib.conv2d(x, w, y, 3, 3, 1, 1, mode, stat_partial=stat if mode == 0 else None)


def wgrad_case(c, hw):
    x = torch.randn(N_IMG, hw, hw, c, device="cuda").to(dt)
    dy = torch.randn(N_IMG, hw, hw, c, device="cuda").to(dt)
    slab = torch.empty(N_IMG // 128 * c * 9 * c, device="cuda")
    return lambda: lib.conv2d_wgrad(x, dy, slab, 3, 3, 1, 1, 128, 1)


def bn_case(C, hw):
    px, ppg = N_IMG * hw * hw, 128 * hw * hw
    x = torch.randn(N_IMG, hw, hw, C, device="cuda").to(dt)
    y = torch.empty_like(x)
    G = N_IMG // 128
    scale, shift = torch.rand(G, C, device="cuda") + 0.5, torch.randn(G, C, device="cuda")
    mask = torch.empty(x.numel() // 8, device="cuda", dtype=torch.uint8)
    return lambda: lib.call("fb_bn_apply", x.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, None, px, C, ppg, 0, 1,
                            mask.data_ptr(), None, hw, lib.dtype_code(dt), None, None)


def region(fns, streams, iters):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main = torch.cuda.current_stream()
    a.record()
    for fn, st in zip(fns, streams):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            for _ in range(iters):
                fn()
    for st in streams:
        main.wait_stream(st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cases = {"conv 256->256 8x8 fwd": conv_case(256, 256, 8), "conv 128->128 16x16 fwd": conv_case(128, 128, 16), "conv 512->512 4x4 fwd": conv_case(512, 512, 4),
             "conv 64->64 32x32 fwd": conv_case(64, 64, 32), "wgrad 256 8x8": wgrad_case(256, 8), "wgrad 128 16x16": wgrad_case(128, 16)}
    bns = {"bn_apply C64 32x32": bn_case(64, 32), "bn_apply C128 16x16": bn_case(128, 16)}
    if os.environ.get("PROBE_CONVS"):
        cases = {k: v for k, v in cases.items() if k.startswith("conv") and "64->64" not in k}
        # two convolutions on two streams (what two half-group lanes would run when both are in an MFMA-bound kernel)
        c1, c2 = conv_case(128, 128, 16), conv_case(128, 128, 16)
        for f in (c1, c2):
            f()
        ta = region([c1], [s1], 10)
        tab = region([c1, c2], [s1, s2], 10)
        print(f"two conv 128->128 16x16 fwd on two streams: alone {ta:.0f} us, pair {tab:.0f} us (2 x alone = {2 * ta:.0f})")
    print("| MFMA-side kernel | alone us | HBM-side kernel | alone us | both, per pair us | sum | hidden |\n|---|---|---|---|---|---|---|")
    for cn, cf in cases.items():
        for bn, bf in bns.items():
            for f in (cf, bf):
                f()
            ta = region([cf], [s1], 10)
            tb = region([bf], [s2], 10)
            k = max(1, round(ta / tb))                      # launches of the short kernel per launch of the long one
            tab = region([cf, lambda: [bf() for _ in range(k)]], [s1, s2], 10)
            print(f"| {cn} | {ta:.0f} | {k} x {bn} | {k * tb:.0f} | {tab:.0f} | {ta + k * tb:.0f} | {100 * (ta + k * tb - tab) / min(ta, k * tb):.0f} % of the shorter |")


if __name__ == "__main__":
    main()
