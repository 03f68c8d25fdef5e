"""Premise probe for two half-group lanes: does the device finish two chunk groups of G/2 chunks, free-running on two streams (each with its own
weight-gradient stream), sooner than one group of G chunks?  Two Engine objects, two host threads, replayed command lists.  GPU box:

    python tools/lanes_probe.py [G] [iters] [offset_ms]
"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fullbatchtraining_amd import engine as E
from fullbatchtraining_amd.cfg import compose
from fullbatchtraining_amd.models import construct_model


def make(G, seed):
    torch.manual_seed(1)
    model = construct_model(compose(["model=resnet18", "model.stem=CIFAR"]).model, 3, 10)
    eng = E.Engine(model, 32, 128, G, compute_dtype=torch.bfloat16)
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(G * 128, 3, 32, 32, generator=gen)
    y = torch.randint(0, 10, (G * 128,), generator=gen).cuda()
    patches = E.stem_patches(x.cuda(), eng.plan.stem, torch.bfloat16)
    eng.prep_weights(eng.theta, 1)
    return eng, patches, y


def run(eng, patches, y, G, iters, stream, delay=0.0):
    time.sleep(delay)
    with torch.cuda.stream(stream):
        for _ in range(iters):
            eng.group_gradient(patches, y, G, eng.g)


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 98
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    offset = float(sys.argv[3]) / 1e3 if len(sys.argv) > 3 else 0.0
    one = make(G, 1)
    s0 = torch.cuda.Stream()
    run(*one, G, 2, s0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(*one, G, iters, s0)
    torch.cuda.synchronize()
    t_one = (time.perf_counter() - t0) / iters * 1e3
    print(f"one lane, groups of {G} chunks: {t_one:.2f} ms per group", flush=True)
    del one
    torch.cuda.empty_cache()
    a, b = make(G // 2, 2), make(G // 2, 3)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    run(*a, G // 2, 2, sa)
    run(*b, G // 2, 2, sb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(*a, G // 2, iters, sa)
    torch.cuda.synchronize()
    t_half = (time.perf_counter() - t0) / iters * 1e3
    print(f"one lane, groups of {G // 2} chunks: {t_half:.2f} ms per group ({2 * t_half:.2f} per {G} chunks)", flush=True)
    ta = threading.Thread(target=run, args=(*a, G // 2, iters, sa))
    tb = threading.Thread(target=run, args=(*b, G // 2, iters, sb, offset))
    t0 = time.perf_counter()
    ta.start(); tb.start(); ta.join(); tb.join()
    torch.cuda.synchronize()
    t_two = (time.perf_counter() - t0) / iters * 1e3
    print(f"two lanes of {G // 2} chunks (second lane {offset * 1e3:.0f} ms late): {t_two:.2f} ms per {G} chunks "
          f"({100 * (t_two / t_one - 1):+.1f} % against one lane of {G})", flush=True)


if __name__ == "__main__":
    main()
