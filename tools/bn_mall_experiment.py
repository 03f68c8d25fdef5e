"""Can the BN-backward apply pass read (dout, x) out of the 256 MB Infinity Cache if it runs right behind the reduce pass of the SAME
chunks?  Whole group (98 chunks: reduce all, finalize, apply all -- 10 bytes per element from HBM) vs sub-batches of S chunks
(reduce S, finalize S, apply S, next S ...).  GPU box:   python tools/bn_mall_experiment.py"""
import sys

import torch

sys.path.insert(0, ".")
from fullbatchtraining_amd import lib  # noqa: E402


def bench(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    G = 98
    h = lib.load()
    dt = lib.dtype_code(torch.bfloat16)
    for C, hw in ((64, 32), (128, 16), (256, 8), (512, 4)):
        ppg = 128 * hw * hw
        px = G * ppg
        x = torch.randn(px, C, device="cuda").bfloat16()
        dout = torch.randn(px, C, device="cuda").bfloat16()
        dx = torch.empty_like(x)
        mask = torch.randint(0, 255, (px * C // 8,), dtype=torch.uint8, device="cuda")
        scale = torch.rand(G, C, device="cuda") + 0.5
        mean, invstd = torch.randn(G, C, device="cuda") * 0.1, torch.rand(G, C, device="cuda") + 0.5
        coef = torch.zeros(G, C, 3, device="cuda")
        part = torch.empty(h.fb_ws_bn_partial_floats(px, C), device="cuda")
        gout = torch.zeros(G, 2 * C, device="cuda")
        eb = 2

        def run(S):
            for g0 in range(0, G, S):
                g = min(S, G - g0)
                p, o = g * ppg, g0 * ppg * C * eb
                rows = h.fb_bn_bwd_reduce_rows(p, ppg)
                lib.call("fb_bn_bwd_reduce", dout.data_ptr() + o, None, mask.data_ptr() + o // 16, x.data_ptr() + o, mean.data_ptr() + 4 * g0 * C,
                         invstd.data_ptr() + 4 * g0 * C, C, 0, part.data_ptr(), p, C, ppg, dt)
                lib.call("fb_bn_bwd_finalize", part.data_ptr(), rows, g, C, float(ppg), scale.data_ptr() + 4 * g0 * C, mean.data_ptr() + 4 * g0 * C,
                         invstd.data_ptr() + 4 * g0 * C, C, 0, gout.data_ptr() + 4 * g0 * 2 * C, gout.data_ptr() + 4 * (g0 * 2 * C + C), 2 * C,
                         coef.data_ptr() + 4 * g0 * C * 3, 0)
                lib.call("fb_bn_bwd_apply", dout.data_ptr() + o, None, mask.data_ptr() + o // 16, x.data_ptr() + o, coef.data_ptr() + 4 * g0 * C * 3,
                         dx.data_ptr() + o, None, p, C, ppg, dt, None, None)

        base = bench(lambda: run(G))
        line = f"C={C:4d} {hw:2d}x{hw:<2d}  whole group {base:7.0f} us"
        for S in (14, 7, 4, 2):
            t = bench(lambda: run(S))
            line += f" | S={S}: {t:7.0f} us ({t / base:.2f}x, {2 * S * ppg * C * eb / 1e6:.0f} MB live)"
        print(line, flush=True)


if __name__ == "__main__":
    main()
