"""Per-launch breakdown of one chunk group (forward + backward) of the benchmark workload: every library call of the schedule with its
shape, duration (HIP events, one stream), algorithmic TFLOP/s and GB/s.  GPU box:

    FB_WGRAD_STREAM=0 python tools/step_breakdown.py [bf16|f32] [G] [model stem pixels chunk]
"""
import os
import sys
from collections import defaultdict

os.environ.setdefault("FB_WGRAD_STREAM", "0")
os.environ["FB_REPLAY"] = "0"          # every launch through the (timed) Python call wrapper
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fullbatchtraining_amd import engine as E
from fullbatchtraining_amd import lib
from fullbatchtraining_amd.cfg import compose
from fullbatchtraining_amd.models import construct_model


def main():
    dtype = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.bfloat16
    G = int(sys.argv[2]) if len(sys.argv) > 2 else (98 if dtype == torch.bfloat16 else 49)
    eb = 2 if dtype == torch.bfloat16 else 4
    model_name, stem, pixels, chunk = (sys.argv[3:7] + ["resnet18", "CIFAR", "32", "128"][len(sys.argv[3:7]):])
    pixels, chunk = int(pixels), int(chunk)
    torch.manual_seed(1)
    model = construct_model(compose([f"model={model_name}", f"model.stem={stem}"]).model, 3, 10)
    eng = E.Engine(model, pixels, chunk, G, compute_dtype=dtype)
    gen = torch.Generator().manual_seed(1234)
    x = torch.randn(G * chunk, 3, pixels, pixels, generator=gen)
    y = torch.randint(0, 10, (G * chunk,), generator=gen).cuda()
    patches = E.stem_patches(x.cuda(), eng.plan.stem, dtype)
    eng.prep_weights(eng.theta, 1)
    records = []
    real_call = E.call

    def describe(name, args):
        if name == "fb_conv2d":
            a = args[0]._obj
            macs = a.n_img * a.Hd * a.Wd * a.Cd * a.R * a.S * a.Cs if a.mode == 0 else a.n_img * a.Hs * a.Ws * a.Cs * a.R * a.S * a.Cd
            byt = (a.n_img * a.Hs * a.Ws * a.Cs + a.n_img * a.Hd * a.Wd * a.Cd) * eb + (a.n_img * a.Hd * a.Wd * a.Cd * eb if a.addend else 0)
            return f"{'fwd' if a.mode == 0 else 'dgrad'} {a.Cs}->{a.Cd} k{a.R} s{a.stride} {a.Hs}x{a.Ws}->{a.Hd}x{a.Wd}{' +add' if a.addend else ''}", 2 * macs, byt
        if name == "fb_conv2d_wgrad":
            a = args[0]._obj
            macs = a.n_img * a.Hd * a.Wd * a.Cd * a.R * a.S * a.Cs
            return f"wgrad {a.Cs}->{a.Cd} k{a.R} s{a.stride} {a.Hs}x{a.Ws} split{a.split_k}", 2 * macs, (a.n_img * a.Hs * a.Ws * a.Cs + a.n_img * a.Hd * a.Wd * a.Cd) * eb
        if name in ("fb_bn_apply", "fb_bn_bwd_reduce", "fb_bn_bwd_apply"):
            if name == "fb_bn_apply":
                px, C, passes = args[7], args[8], 2 + (1 if args[4] else 0)
            elif name == "fb_bn_bwd_reduce":
                px, C, passes = args[9], args[10], 2
            else:
                px, C, passes = args[7], args[8], 3 + (1 if args[6] else 0)
            return f"{name[3:]} C{C} px/img {px // (G * chunk)}", 0, px * C * eb * passes
        return name[3:], 0, 0

    def timed_call(name, *args):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        real_call(name, *args)
        b.record()
        records.append((name, args, a, b))

    for it in range(2):                                        # first round warms up
        records.clear()
        E.call = timed_call
        eng.group_gradient(patches, y, G, eng.g)
        E.call = real_call
        torch.cuda.synchronize()
    rows, classes = [], defaultdict(float)
    for name, args, a, b in records:
        desc, flop, byt = describe(name, args)
        us = a.elapsed_time(b) * 1e3
        rows.append((us, desc, flop, byt))
        classes[name] += us
    total = sum(r[0] for r in rows)
    print(f"{model_name} {stem} {pixels}px: one chunk group of {G} chunks ({G * chunk} images), {str(dtype)}: {total / 1e3:.2f} ms of launches"
          + (f"; x {390 / G:.2f} groups per step = {total / 1e3 * 390 / G:.1f} ms" if model_name == "resnet18" else f" = {G * chunk / total * 1e6:.0f} images/s") + "\n")
    print("| # | call | us | TFLOP/s | GB/s (algorithmic) |\n|---|---|---|---|---|")
    for i, (us, desc, flop, byt) in enumerate(rows):
        if us < 15:
            continue
        if model_name != 'resnet18' and us < total / 400:
            continue
        print(f"| {i} | {desc} | {us:.0f} | {flop / us / 1e6:.0f} | {byt / us / 1e3:.0f} |")
    if model_name != "resnet18":                               # hundreds of launches: aggregate by shape
        agg = defaultdict(lambda: [0, 0.0, 0, 0])
        for us, desc, flop, byt in rows:
            a = agg[desc]
            a[0] += 1; a[1] += us; a[2] += flop; a[3] += byt
        print("\n| call shape | launches | us total | share | TFLOP/s | GB/s (algorithmic) |\n|---|---|---|---|---|---|")
        for desc, (n, us, flop, byt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
            print(f"| {desc} | {n} | {us:.0f} | {100 * us / total:.1f} % | {flop / us / 1e6:.0f} | {byt / us / 1e3:.0f} |")
    print("\n| entry point | ms per group | share |\n|---|---|---|")
    for k, v in sorted(classes.items(), key=lambda kv: -kv[1]):
        print(f"| {k} | {v / 1e3:.2f} | {100 * v / total:.1f} % |")


if __name__ == "__main__":
    main()
