"""Race hunting at the benchmark's real shape: ResNet-18, 32 px, chunks of 128, several chunk groups per step.

Runs three full-gradient evaluations + updates in ONE stream (FB_WGRAD_STREAM=0, FB_ACC_OVERLAP=0: the reference trace -- the engine is
deterministic, so everything must be bit-identical) and then repeatedly with the schedule under test (the environment as given), and reports
the first tensors that differ: per-chunk losses, per-chunk squared norms, the averaged gradient per parameter tensor, parameters.

    python tools/race_probe.py [repeats] [n_chunks] [G] [bf16|f32] [steps]      (soak: `3 390 98 bf16 40` = the benchmark's full step, 40 times)

FB_PROBE_INSTRUMENT=1 records checkpoints into the backward pass of the 32 x 32 stage (per-chunk sums of squares of every input gradient /
BatchNorm-backward result, read as fp32 words) and prints the first non-finite ones; FB_PROBE_TWICE=1 additionally repeats every BatchNorm
backward reduction into a private buffer and prints the rows / channels where the two differ.  That is how the store-data hazard behind
csrc/common.h store_b128_guard was found: ONE bf16 of the resident-filter convolution's output (dword 1 of a 16-byte store) held the low
half of a byte offset (~1e38) -- the value the register allocator had put into that register in the instruction after the store.
tools/race_probe.sh runs the switches that narrowed it down (only with the resident-filter kernel, only with two busy streams)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import make_data  # noqa: E402


FD = float(os.environ.get("FB_PROBE_FD", "0"))       # block_strength of the finite-difference regulariser (fp32 passes, per-chunk weight sets)


def build(pixels, chunk, G, dtype):
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.engine import Engine, stem_patches
    from fullbatchtraining_amd.models import construct_model

    cfg = compose(["model=resnet18", "model.stem=CIFAR"])
    torch.manual_seed(0)
    model = construct_model(cfg.model, 3, 10)
    return Engine(model, pixels, chunk, G, compute_dtype=dtype, fd_sets=1 if FD else 0), stem_patches


TWICE = []
CHECKS = []          # (label, per-chunk squared norms of a tensor of the backward chain read as fp32 words: non-finite = corrupted)


def instrument(eng):
    """Checkpoints inside the backward pass (recorded into the command lists like any other launch): after every input-gradient convolution
    and every BatchNorm backward of the 32x32 stage, the per-chunk sum of squares of the result."""
    from fullbatchtraining_amd import lib

    def check(label, t, G):
        words = t.numel() * t.element_size() // 4 // G
        out = torch.zeros(G, device="cuda")
        ws = torch.zeros(G * lib.MT_BLOCKS, device="cuda")
        lib.call("fb_mt_sqnorm", t.data_ptr(), words, G, words, 1.0, None, 0.0, out.data_ptr(), ws.data_ptr())
        CHECKS.append((label, out, ws))

    dgrad0, bn0 = eng._dgrad, eng._bn_bwd

    def dgrad(L, dx, G, wsets, **kw):
        out = dgrad0(L, dx, G, wsets, **kw)
        if L.wout == 32 and os.environ.get("FB_PROBE_CHECKS", "1") != "0":
            check(f"G{G} dgrad {L.conv_name} in", dx, G)
            check(f"G{G} dgrad {L.conv_name} out", out, G)
        return out

    def bn_bwd(L, dout, mask, G, gout, pidx, want_dy, reduced=False):
        if L.wout == 32 and os.environ.get("FB_PROBE_CHECKS", "1") != "0":
            check(f"G{G} bn_bwd {L.bn_name} in-before", dout, G)
            check(f"G{G} bn_bwd {L.bn_name} x-before", L.x[:G * eng.chunk], G)
        twice = L.wout == 32 and os.environ.get("FB_PROBE_TWICE", "0") == "1"
        if twice:            # private partial-sum buffers: the pass's own and those of a second, identical reduction right after it
            px, ppg = G * eng.chunk * L.hout * L.wout, eng.chunk * L.hout * L.wout
            rows = lib.load().fb_bn_bwd_reduce_rows(px, ppg)
            p1, p2 = torch.zeros(2 * rows * L.cout, device="cuda"), torch.zeros(2 * rows * L.cout, device="cuda")
            keep, eng.stat_ws = eng.stat_ws, p1
        dx, dy = bn0(L, dout, mask, G, gout, pidx, want_dy, reduced)
        if twice:
            eng.stat_ws = keep
            bits = eng.masks.get(mask.data_ptr()) if mask is not None else None
            lib.call("fb_bn_bwd_reduce", dout.data_ptr(), None if bits is not None else (mask.data_ptr() if mask is not None else None),
                     bits.data_ptr() if bits is not None else None, L.x.data_ptr(), eng.mean_tab[pidx].data_ptr(), L.invstd.data_ptr(),
                     eng.plan.ch_total, L.ch_off, p2.data_ptr(), px, L.cout, ppg, eng.dtc)
            TWICE.append((f"G{G} {L.bn_name}", p1, p2, rows, L.cout))
        if L.wout == 32 and os.environ.get("FB_PROBE_CHECKS", "1") != "0":
            check(f"G{G} bn_bwd {L.bn_name} stat_ws", eng.stat_ws[:2 * G * eng.chunk * 8 * 64], 1)
            check(f"G{G} bn_bwd {L.bn_name} in", dout, G)
            check(f"G{G} bn_bwd {L.bn_name} coef", L.coef, 1)
            check(f"G{G} bn_bwd {L.bn_name} dx", dx, G)
        return dx, dy

    eng._dgrad, eng._bn_bwd = dgrad, bn_bwd


def run(x, y, pixels, chunk, G, dtype, steps, lrs):
    eng, stem_patches = build(pixels, chunk, G, dtype)
    del CHECKS[:]
    del TWICE[:]
    if os.environ.get("FB_PROBE_INSTRUMENT", "0") == "1":
        instrument(eng)
    patches, yd = stem_patches(x.cuda(), eng.plan.stem, dtype), y.cuda()
    trace = []
    for step in range(steps):
        loss, correct, sq = eng.full_gradient(patches, yd, lrs[step], block_strength=FD)
        avg = eng.avg.clone()
        eng.grad_and_param_sqnorm()
        eng.sgd_step(lrs[step], 5e-4, 0.9, 0.0, True, grad_clip=0.25)
        trace.append(dict(loss=loss.clone(), correct=correct.clone(), sq=sq.clone(), avg=avg, theta=eng.theta.clone(),
                          rm=eng.running_mean.clone()))
        if TWICE and step == 0:
            torch.cuda.synchronize()
            for label, p1, p2, rows, C in TWICE:
                a, b = p1.view(2, rows, C), p2.view(2, rows, C)
                if not torch.equal(a, b):
                    diff = ((a != b) | torch.isnan(a)).any(2).any(0).nonzero().flatten().tolist()
                    r0 = diff[0]
                    print(f"    reduction differs from its repetition: {label}: {len(diff)} of {rows} rows, rows {diff[:6]}..{diff[-3:]}  "
                          f"first row pass {a[:, r0, :4].flatten().tolist()} again {b[:, r0, :4].flatten().tolist()} "
                          f"columns {((a[:, r0] != b[:, r0]) | torch.isnan(a[:, r0])).any(0).nonzero().flatten().tolist()[:20]}")
        if CHECKS and step == 0:
            torch.cuda.synchronize()
            bad = [(label, out.tolist()) for label, out, _ in CHECKS if not bool(torch.isfinite(out).all())]
            if bad:
                print(f"    first corrupted checkpoints (of {len(CHECKS)}): " + "; ".join(f"{l} {v}" for l, v in bad[:4]))
    torch.cuda.synchronize()
    return eng, trace


def report(eng, ref, got):
    bad = False
    for step, (a, b) in enumerate(zip(ref, got)):
        for key in ("loss", "correct", "sq", "avg", "theta", "rm"):
            if torch.equal(a[key], b[key]):
                continue
            bad = True
            if key in ("loss", "correct", "sq"):
                idx = (a[key] != b[key]).nonzero().flatten().tolist()
                print(f"  step {step} {key}: chunks {idx} differ  ref {a[key][idx].tolist()}  got {b[key][idx].tolist()}")
            elif key in ("avg", "theta"):
                names = []
                for name in eng.plan.param_names:
                    lo = eng.plan.offsets[name]
                    n = 1
                    for d in eng.plan.param_shapes[name]:
                        n *= d
                    u, v = a[key][lo:lo + n], b[key][lo:lo + n]
                    if not torch.equal(u, v):
                        names.append(f"{name}({float((u - v).norm() / (u.norm() + 1e-30)):.1e})")
                print(f"  step {step} {key}: {len(names)} of {len(eng.plan.param_names)} tensors differ: {' '.join(names[:12])}{' ...' if len(names) > 12 else ''}")
            else:
                print(f"  step {step} {key}: differs, rel {float((a[key] - b[key]).norm() / a[key].norm()):.2e}")
        if bad:
            break
    return bad


def main():
    repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    G = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dtype = dict(bf16=torch.bfloat16, f32=torch.float32)[sys.argv[4] if len(sys.argv) > 4 else "bf16"]
    pixels, chunk = 32, 128
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    lrs = [0.0] + [0.4] * (steps - 1)
    x, y = make_data(chunk * n_chunks, pixels)
    base = dict(os.environ)
    os.environ["FB_ACC_OVERLAP"] = "0"
    os.environ["FB_WGRAD_STREAM"] = "0"
    eng, ref = run(x, y, pixels, chunk, G, dtype, steps, lrs)
    eng2, ref2 = run(x, y, pixels, chunk, G, dtype, steps, lrs)
    finite = all(bool(torch.isfinite(t[k]).all()) for t in ref for k in t)
    print("reference trace (one stream) reproducible:", not report(eng, ref, ref2), " finite:", finite, " losses", [float(t["loss"].mean()) for t in ref])
    os.environ.clear()
    os.environ.update(base)
    n_bad = 0
    for rep in range(repeats):
        eng, got = run(x, y, pixels, chunk, G, dtype, steps, lrs)
        bad = report(eng, ref, got)
        n_bad += bad
        print(f"run {rep}: {'MISMATCH' if bad else 'identical'}", flush=True)
    print(f"{n_bad} of {repeats} runs differ from the overlap-off trace  (env: " +
          " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("FB_")) + ")")


if __name__ == "__main__":
    main()
