"""How many significand bits do the operands of the finite-difference passes need?  (CPU, float64 oracle.)

The oracle's storage-rounding hook `q` is applied to every convolution operand (activations, activation gradients, weights); between
the rounding points the arithmetic is exact (float64).  Measured on ResNet-18, 16 px, one chunk of 64 images, forward differences:

    operands rounded to      raw gradient err   regularised gradient err
    fp32 (24 bits)           2.4e-3             3.5e-2        <- what the reference's fp32 run has
    bf16 x 2 pieces (16 b)   1.4e-2             1.4e-1        <- "bf16x3" products: 4x the fp32 error, outside the parity tolerance
    bf16 x 3 pieces (24 b)   3e-8               2e-7          <- "bf16x6": exact split of an fp32 value
    bf16 (8 bits)            2.7e-1             1.6

and both operands matter alike (weights only at 16 bits: 9.8e-2; activations only: 9.8e-2): the second pass must resolve a
perturbation of ~1e-4 |w| through round(w + eps v) - round(w).  Hence the fp32 path of the engine splits each fp32 operand into THREE
bf16 pieces inside the convolution kernels (csrc/common.h: split_f32x8 / mma_split6) instead of two.
"""
import sys, torch, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import fb_oracle as orc
from fullbatchtraining_amd.cfg import compose
from fullbatchtraining_amd.models import construct_model
from tests.helpers import make_data
torch.set_num_threads(8)
def bf(t): return t.to(torch.bfloat16).to(t.dtype)
def split2(t):
    hi = bf(t); lo = bf(t - hi); return hi + lo
def split3(t):
    hi = bf(t); r = t - hi; mid = bf(r); lo = bf(r - mid); return hi + mid + lo
def f32(t): return t.float().double()
pixels, chunk = 16, 64
for seed in (7,):
    cfg = compose(["data.pixels=16"])
    torch.manual_seed(seed)
    model = construct_model(cfg.model, 3, 10)
    x, y = make_data(chunk, pixels)
    res = {}
    for name, q in (("exact", orc.identity), ("f32round", f32), ("split2", split2), ("split3", split3), ("bf16", bf)):
        state = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
        params, buffers = orc.split_state(state)
        spec = orc.Spec(18)
        g, loss, _ = orc.chunk_gradient(spec, params, buffers, x.double(), y, q)
        raw = torch.cat([t.reshape(-1) for t in g]).clone()
        reg = orc.gradreg(spec, params, buffers, [t.clone() for t in g], x.double(), y, 0.1, 0.5, 1e-2, "forward-differences", q)
        reg = torch.cat([t.reshape(-1) for t in reg])
        res[name] = (raw, reg, reg - raw)
    t = res["exact"]
    for name in res:
        r = res[name]
        print(f"{name:9s} raw err {float((r[0]-t[0]).norm()/t[0].norm()):.2e}  reg err {float((r[1]-t[1]).norm()/t[1].norm()):.2e}  vhp-term err {float((r[2]-t[2]).norm()/t[2].norm()):.2e}  |vhp term|/|g| {float(t[2].norm()/t[0].norm()):.3f}")
print("---- which operand matters")
def is_w(t): return t.dim()==4 and t.shape[-1]==t.shape[-2] and t.shape[-1] in (1,3,7)
def mk(qa, qw):
    def q(t): return qw(t) if is_w(t) else qa(t)
    return q
idn = lambda t: t
res = {}
for name, q in (("exact", orc.identity), ("w:split2 a:exact", mk(idn, split2)), ("w:exact a:split2", mk(split2, idn)), ("w:split3 a:split2", mk(split2, split3)),
                ("w:split2 a:split3", mk(split3, split2)), ("w:f32 a:f32", mk(f32, f32)), ("w:exact a:f32", mk(f32, idn))):
    state = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    params, buffers = orc.split_state(state)
    spec = orc.Spec(18)
    g, loss, _ = orc.chunk_gradient(spec, params, buffers, x.double(), y, q)
    raw = torch.cat([t.reshape(-1) for t in g]).clone()
    reg = orc.gradreg(spec, params, buffers, [t.clone() for t in g], x.double(), y, 0.1, 0.5, 1e-2, "forward-differences", q)
    reg = torch.cat([t.reshape(-1) for t in reg])
    res[name] = (raw, reg, reg - raw)
t = res["exact"]
for name in res:
    r = res[name]
    print(f"{name:20s} raw err {float((r[0]-t[0]).norm()/t[0].norm()):.2e}  reg err {float((r[1]-t[1]).norm()/t[1].norm()):.2e}  vhp-term err {float((r[2]-t[2]).norm()/t[2].norm()):.2e}")
