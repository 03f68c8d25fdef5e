"""How well conditioned is the finite-difference regulariser at BASELINE config 5's shape (ResNet-152, standard stem, 224 px, one chunk of 128)?

Runs the raw and the regularised (forward differences, block_strength 0.5, eps 1e-2, lr 0.1) chunk gradient in several fp32-class arithmetic
variants and prints their pairwise relative distances:
  f16x2            two scaled fp16 pieces per operand (22 bits), the engine's default for the regulariser
  bf16x6           three bf16 pieces per operand: products exact to 2^-23
  bf16x6/order     the same arithmetic with another K-slice count of the weight gradients (nominal_group): fp32 summation ORDER only
  exact-f32 MFMA   (a run with FB_F32_EXACT=1 first) v_mfma_f32_16x16x4_f32 chains: fp32 operands, another rounding of the partial sums
"order" differs in nothing but the order of the fp32 additions of the weight gradients; "exact-f32 MFMA" vs bf16x6 is the distance between two
legitimate fp32 evaluations of the same chunk gradient -- the noise floor any 32-bit implementation (the reference's included) sits on."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fullbatchtraining_amd.cfg import compose  # noqa: E402
from fullbatchtraining_amd.engine import Engine, stem_patches  # noqa: E402
from fullbatchtraining_amd.models import construct_model  # noqa: E402

depth, pixels, chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 152, int(sys.argv[2]) if len(sys.argv) > 2 else 224, 128
gen = torch.Generator().manual_seed(1234)
x = torch.randn(chunk, 3, pixels, pixels, generator=gen)
y = torch.randint(0, 10, (chunk,), generator=gen).cuda()
res = {}
variants = [("f16x2", "f16x2", 1), ("bf16x6", "bf16x6", 1), ("bf16x6/order", "bf16x6", 8)]
if os.environ.get("FB_F32_EXACT", "0") not in ("", "0"):      # (read once per process by the library) the exact-f32 MFMA chain: run as its own process,
    variants = [("exact-f32", "bf16x6", 1)]                     # its vectors are saved and compared by the next ordinary run
for label, split, nominal in variants:
    os.environ["FB_F32_SPLIT"] = split
    cfg = compose([f"model=resnet{depth}", "model.stem=standard"])
    torch.manual_seed(0)
    model = construct_model(cfg.model, 3, 10)
    eng = Engine(model, pixels, chunk, 1, compute_dtype=torch.float32, fd_sets=1, nominal_group=nominal)
    patches = stem_patches(x.cuda(), eng.plan.stem, torch.float32)
    eng.prep_weights(eng.theta, 1)
    eng.group_gradient(patches, y, 1, eng.g)
    raw = eng.g[0].double().clone()
    loss, _, sq = eng.full_gradient(patches, y, 0.1, block_strength=0.5, eps=1e-2)
    torch.cuda.synchronize()
    res[label] = (raw.cpu(), eng.avg.double().cpu().clone(), float(loss[0]), float(eng.eps_n[0]))
    print(f"{label}: loss {float(loss[0]):.6f} |g| {float(raw.norm()):.2f} eps_n {float(eng.eps_n[0]):.3e} |regularised| {float(eng.avg.double().norm()):.2f}", flush=True)
    del eng, patches
    torch.cuda.empty_cache()


save = f"/tmp/fd_cond_exact_{depth}_{pixels}.pt"          # (hundreds of MB: scratch outside the repository, removed after the comparison)
if "exact-f32" in res:
    torch.save(res["exact-f32"], save)
    sys.exit(0)
pairs = [("f16x2", "bf16x6"), ("bf16x6/order", "bf16x6")]
if os.path.isfile(save):
    res["exact-f32 MFMA"] = torch.load(save)
    os.remove(save)
    pairs.append(("exact-f32 MFMA", "bf16x6"))


def rel(a, b):
    return float((a - b).norm() / b.norm())


print("\n| pair | raw chunk gradient | regularised chunk gradient | finite-difference term alone |\n|---|---|---|---|")
for a, b in pairs:
    fa, fb = res[a][1] - res[a][0], res[b][1] - res[b][0]
    print(f"| {a} vs {b} | {rel(res[a][0], res[b][0]):.3e} | {rel(res[a][1], res[b][1]):.3e} | {rel(fa, fb):.3e} (|FD term| / |g| = {float(fb.norm() / res[b][0].norm()):.3f}) |")
