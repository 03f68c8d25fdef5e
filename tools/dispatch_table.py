"""Which kernel fb_conv2d / fb_conv2d_wgrad select for every convolution launch of one chunk group, for several group sizes side by side (the library's own launch
records: fb_profile_read_launches).  GPU box:

    python tools/dispatch_table.py resnet152 standard 224 bf16 2 16        (a rank's share of BASELINE config 5: 2 chunks of an 8-GPU job, 16 on one GPU)
"""
import gc
import os
import sys

os.environ.setdefault("FB_WGRAD_STREAM", "0")
os.environ["FB_REPLAY"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fullbatchtraining_amd import engine as E
from fullbatchtraining_amd import lib
from fullbatchtraining_amd.cfg import compose
from fullbatchtraining_amd.models import construct_model


def launches_of(model_name, stem, pixels, dtype, G, chunk=128):
    torch.manual_seed(1)
    model = construct_model(compose([f"model={model_name}", f"model.stem={stem}"]).model, 3, 10)
    eng = E.Engine(model, pixels, chunk, G, compute_dtype=dtype)
    gen = torch.Generator().manual_seed(1234)
    x = torch.randn(G * chunk, 3, pixels, pixels, generator=gen)
    y = torch.randint(0, 10, (G * chunk,), generator=gen).cuda()
    patches = E.stem_patches(x.cuda(), eng.plan.stem, dtype)
    eng.prep_weights(eng.theta, 1)
    eng.group_gradient(patches, y, G, eng.g)
    torch.cuda.synchronize()
    lib.profile_enable(True, 1 << 16)
    eng.group_gradient(patches, y, G, eng.g)
    torch.cuda.synchronize()
    rec = lib.profile_read_launches()
    lib.profile_read()
    lib.profile_enable(False)
    out = []
    for cls, w, ms in rec:
        if cls not in ("igemm_fwd", "igemm_dgrad", "wgrad"):
            continue
        n, hs, ws, cs, hd, wd, cd, r, stride, flags, kernel = w
        kind = {"igemm_fwd": "forward", "igemm_dgrad": "input gradient", "wgrad": "weight gradient"}[cls]
        extra = ""
        if cls == "igemm_dgrad":
            extra = {0: "", 1: " + addend", 2: " + pooled addend"}[flags & 3] + (" (through the ReLU bitmask)" if flags & 4 else "")
        out.append((f"{kind} {cs}->{cd} k{r} s{stride} @{hs}x{ws}{extra}", lib.PROF_KERNELS.get(kernel, "?"), 1000 * ms))
    del eng, patches
    gc.collect(), torch.cuda.empty_cache()
    return out


def main():
    model_name, stem, pixels, dt = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    groups = [int(g) for g in sys.argv[5:]]
    dtype = torch.bfloat16 if dt == "bf16" else torch.float32
    tabs = {G: launches_of(model_name, stem, pixels, dtype, G) for G in groups}
    shapes = []
    for G in groups:
        for s, k, us in tabs[G]:
            if s not in shapes:
                shapes.append(s)
    print(f"{model_name} / {stem} stem / {pixels} px / {dt}: kernel per convolution launch of one chunk group (launches of that shape per group; us per launch)\n")
    print("| launch | " + " | ".join(f"{G} chunks ({G * 128} images)" for G in groups) + " |\n|---|" + "---|" * len(groups))
    for s in shapes:
        cells = []
        for G in groups:
            hits = [(k, us) for ss, k, us in tabs[G] if ss == s]
            names = sorted({k for k, _ in hits})
            cells.append(" / ".join(names) + f" ({len(hits)} x {sum(u for _, u in hits) / len(hits):.0f} us)")
        mark = " **" if len({c.split(" (")[0] for c in cells}) > 1 else ""
        print(f"| {s}{mark} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main()
