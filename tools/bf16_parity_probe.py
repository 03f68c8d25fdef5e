"""bf16 parity at the benchmark's real shape (ResNet-18, 32x32, chunks of 128): how far is the bf16 engine's K-chunk MEAN gradient from
the f32 engine's (itself 2.5e-6..2e-3 from the float64 oracle, tests/test_gpu_engine.py) as K grows, at random init with random labels
(the hardest state: the gradient is the residual of a cancelling sum) and after f32 training steps on a learnable dataset?

    python tools/bf16_parity_probe.py            (GPU box)

Prints a table; the numbers are kept in DESIGN.md section 2 and bound tests/test_gpu_bf16_parity.py.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(dtype, seed, G):
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.engine import Engine
    from fullbatchtraining_amd.models import construct_model

    torch.manual_seed(seed)
    model = construct_model(compose([]).model, 3, 10)
    return model, Engine(model, 32, 128, G, compute_dtype=dtype)


def mean_grad(eng, x, y, K):
    from fullbatchtraining_amd.engine import stem_patches

    patches = stem_patches(x[:K * 128].cuda(), eng.plan.stem, eng.dt)
    rm, rv, nbt = eng.running_mean.clone(), eng.running_var.clone(), eng.num_batches_tracked
    loss, _, sq = eng.full_gradient(patches, y[:K * 128].cuda(), 0.1)
    eng.running_mean.copy_(rm), eng.running_var.copy_(rv)            # a measurement, not a training step
    eng.num_batches_tracked = nbt
    torch.cuda.synchronize()
    return eng.avg.double().clone(), loss.double().clone(), sq.double().clone()


def table(tag, e32, e16, x, y, Ks):
    rows = []
    for K in Ks:
        a, la, sa = mean_grad(e32, x, y, K)
        b, lb, sb = mean_grad(e16, x, y, K)
        rel = float((a - b).norm() / a.norm())
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        rows.append(dict(state=tag, K=K, mean_grad_rel_err=rel, cosine=cos, mean_grad_norm=float(a.norm()),
                         loss_abs_err=float((la.mean() - lb.mean()).abs()), chunk_norm_rel_err=float(((sa.sqrt() - sb.sqrt()).abs() / sa.sqrt()).mean())))
        print(json.dumps(rows[-1]))
    return rows


def main():
    Kmax = int(os.environ.get("FB_PROBE_K", "64"))
    Ks = [k for k in (1, 4, 16, 64, 256) if k <= Kmax]
    gen = torch.Generator().manual_seed(1234)
    n = Kmax * 128
    # (a) random init, random labels, N(0,1) inputs: the benchmark's synthetic workload
    x = torch.randn(n, 3, 32, 32, generator=gen)
    y = torch.randint(0, 10, (n,), generator=gen)
    m32, e32 = build(torch.float32, 1, 16)
    m16, e16 = build(torch.bfloat16, 1, 16)
    rows = table("random-init/random-labels", e32, e16, x, y, Ks)
    # (b) learnable dataset (class prototype + noise), same init
    protos = torch.randn(10, 3, 32, 32, generator=gen)
    xl = protos[y] + 0.5 * torch.randn(n, 3, 32, 32, generator=gen)
    rows += table("random-init/learnable", e32, e16, xl, y, Ks)
    # (c) after 10 f32 full-batch steps on the learnable dataset (first 16 chunks): both engines get the trained state
    from fullbatchtraining_amd.engine import stem_patches
    p = stem_patches(xl[:16 * 128].cuda(), e32.plan.stem, torch.float32)
    for step in range(10):
        e32.full_gradient(p, y[:16 * 128].cuda(), 0.05)
        e32.grad_and_param_sqnorm()
        e32.sgd_step(0.05, 5e-4, 0.9, 0.0, True, 1.0)
    torch.cuda.synchronize()
    e16.theta.copy_(e32.theta), e16.running_mean.copy_(e32.running_mean), e16.running_var.copy_(e32.running_var)
    rows += table("after-10-f32-steps/learnable", e32, e16, xl, y, Ks)
    rows += table("after-10-f32-steps/random-labels", e32, e16, x, y, Ks)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bf16_parity_probe.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as handle:
        json.dump(rows, handle, indent=1)


if __name__ == "__main__":
    main()
