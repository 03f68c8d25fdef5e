"""Out-of-bounds read hunting: every operand of the hot kernels is carved out of a slab whose surroundings are poisoned (NaN / 0xFF), the
launch is repeated with zeroed surroundings, and the two results must be bit-identical and finite.  A kernel that reads past an operand
and lets the value reach its result (e.g. masks it by a multiplication) shows up here deterministically.

    python tools/poison_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fullbatchtraining_amd import lib  # noqa: E402

GUARD = 1 << 20      # bytes on either side


def carve(t, poison):
    """A copy of ``t`` in the middle of a slab whose other bytes are NaN-bf16 / 0xFF (poison) or zero."""
    nbytes = t.numel() * t.element_size()
    slab = torch.empty(GUARD + nbytes + GUARD, dtype=torch.uint8, device="cuda")
    if poison:
        if t.dtype == torch.uint8:
            slab.fill_(0xFF)
        else:
            slab.view(torch.bfloat16).fill_(float("nan"))
    else:
        slab.zero_()
    v = slab[GUARD:GUARD + nbytes].view(t.dtype).view(t.shape)
    v.copy_(t)
    return v


def check(name, fn, tensors):
    """fn(*operands) -> tuple of results; operands carved with poisoned and with zeroed surroundings."""
    outs = []
    for poison in (False, True, True):
        ops = [carve(t, poison) if t is not None else None for t in tensors]
        res = fn(*ops)
        torch.cuda.synchronize()
        outs.append([r.clone() for r in res])
    ok = all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(outs[0], outs[1]))
    fin = all(bool(torch.isfinite(r.float()).all()) for r in outs[1])
    print(f"{name:70s} {'ok' if ok and fin else 'DEPENDS ON MEMORY OUTSIDE ITS OPERANDS' if not ok else 'non-finite'}", flush=True)
    return ok and fin


def main():
    torch.manual_seed(0)
    bf = torch.bfloat16
    bad = 0
    for n in (256, 384, 136):
        for (C, W) in ((64, 32), (128, 16), (256, 8), (512, 4)):
            x = (torch.randn(n, W, W, C, device="cuda")).to(bf)
            dy = (torch.randn(n, W, W, C, device="cuda") * 0.1).to(bf)
            d = torch.randn(n, W, W, C, device="cuda").to(bf)
            w = (torch.randn(C, 9, C, device="cuda") * 0.05).to(bf)
            act = torch.randn(n, W, W, C, device="cuda")
            bits = ((act.reshape(-1, 8) > 0).to(torch.int32) << torch.arange(8, device="cuda")).sum(1).to(torch.uint8)
            nblk = (n * W * W + 127) // 128

            def fwd(x_, w_):
                out = torch.empty(n, W, W, C, dtype=bf, device="cuda")
                stat = torch.zeros(2, nblk, C, device="cuda")
                lib.conv2d(x_, w_, out, 3, 3, 1, 1, 0, stat_partial=stat)
                return out, stat
            bad += not check(f"conv fwd 3x3 {C}->{C} W={W} n={n}", fwd, [x, w])

            for amode, mask in ((0, False), (1, False), (1, True)):
                if mask and C != 64:
                    continue

                def dgrad(dy_, w_, d_, bits_):
                    out = torch.empty(n, W, W, C, dtype=bf, device="cuda")
                    lib.conv2d(dy_, w_, out, 3, 3, 1, 1, 1, addend=d_ if amode else None, addend_mode=amode, addend_mask=bits_ if mask else None)
                    return (out,)
                bad += not check(f"conv dgrad 3x3 {C}->{C} W={W} n={n} addend={amode} mask={mask}", dgrad, [dy, w, d, bits])

            ipg = 128 if n % 128 == 0 else n
            for split in (1, 5, 64):
                if split > ipg // 2:
                    continue

                def wgrad(x_, dy_):
                    slab = torch.zeros(n // ipg, split, C, 9, C, device="cuda")
                    lib.conv2d_wgrad(x_, dy_, slab, 3, 3, 1, 1, ipg, split)
                    return (slab,)
                bad += not check(f"conv wgrad 3x3 {C}->{C} W={W} n={n} split={split}", wgrad, [x, dy])

            # BatchNorm passes
            px, ppg = n * W * W, ipg * W * W
            groups = n // ipg
            scale, shift = torch.rand(groups, C, device="cuda") + 0.5, torch.randn(groups, C, device="cuda") * 0.1
            mean_tab, invstd = torch.randn(groups, C, device="cuda") * 0.1, torch.rand(groups, C, device="cuda") + 0.5
            coef = torch.randn(groups, C, 3, device="cuda") * 0.1

            def bn_apply(x_, res_):
                y = torch.empty_like(x)
                b = torch.zeros(x.numel() * 2 // 16, dtype=torch.uint8, device="cuda")
                lib.call("fb_bn_apply", x_.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), res_.data_ptr(), None, None, px, C, ppg, 0, 1,
                         b.data_ptr(), None, 0, lib.dtype_code(bf), None, None)
                return y, b
            bad += not check(f"bn_apply C={C} W={W} n={n}", bn_apply, [x, d])

            def bn_reduce(dout_, bits_, x_):
                rows = lib.load().fb_bn_bwd_reduce_rows(px, ppg)
                part = torch.zeros(2, rows, C, device="cuda")
                lib.call("fb_bn_bwd_reduce", dout_.data_ptr(), None, bits_.data_ptr(), x_.data_ptr(), mean_tab.data_ptr(), invstd.data_ptr(), C, 0,
                         part.data_ptr(), px, C, ppg, lib.dtype_code(bf))
                return (part,)
            bad += not check(f"bn_bwd_reduce C={C} W={W} n={n}", bn_reduce, [dy, bits, x])

            def bn_bwd_apply(dout_, bits_, x_):
                dx, dyo = torch.empty_like(x), torch.empty_like(x)
                lib.call("fb_bn_bwd_apply", dout_.data_ptr(), None, bits_.data_ptr(), x_.data_ptr(), coef.data_ptr(), dx.data_ptr(), dyo.data_ptr(), px, C, ppg,
                         lib.dtype_code(bf), None, None)
                return dx, dyo
            bad += not check(f"bn_bwd_apply C={C} W={W} n={n}", bn_bwd_apply, [dy, bits, x])

        # stem on pre-gathered patches (1x1, 32 -> 64) and its weight gradient
        xs = torch.randn(n, 32, 32, 32, device="cuda").to(bf)
        ws = (torch.randn(64, 1, 32, device="cuda") * 0.1).to(bf)
        dys = (torch.randn(n, 32, 32, 64, device="cuda") * 0.1).to(bf)

        def stem(x_, w_):
            out = torch.empty(n, 32, 32, 64, dtype=bf, device="cuda")
            stat = torch.zeros(2, n * 8, 64, device="cuda")
            lib.conv2d(x_, w_, out, 1, 1, 1, 0, 0, stat_partial=stat)
            return out, stat
        bad += not check(f"stem conv 1x1 32->64 n={n}", stem, [xs, ws])
        ipg = 128 if n % 128 == 0 else n
        for split in (1, 10, 256):
            def stem_wgrad(x_, dy_):
                slab = torch.zeros(n // ipg, split, 64, 1, 32, device="cuda")
                lib.conv2d_wgrad(x_, dy_, slab, 1, 1, 1, 0, ipg, split)
                return (slab,)
            bad += not check(f"stem wgrad 1x1 32->64 n={n} split={split}", stem_wgrad, [xs, dys])
    print(f"{bad} launches depend on memory outside their operands")


if __name__ == "__main__":
    main()
