"""fp32 chunk groups whose tensors exceed 2^31 bytes: one group of 8 chunks vs two of 4 (ResNet-50 @224, chunks of 128), plain and with the regulariser."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fullbatchtraining_amd.cfg import compose
from fullbatchtraining_amd.engine import Engine, stem_patches
from fullbatchtraining_amd.models import construct_model
from tests.helpers import make_data

depth = int(os.environ.get("DEPTH", "50"))
pixels, chunk, n_chunks = 224, 128, int(os.environ.get("NCH", "8"))
x, y = make_data(chunk * n_chunks, pixels)
cfg = compose([f"model=resnet{depth}", "model.stem=standard"])
out = {}
for G in (n_chunks // 2, n_chunks):
    torch.manual_seed(0)
    model = construct_model(cfg.model, 3, 10)
    eng = Engine(model, pixels, chunk, G, compute_dtype=torch.float32, nominal_group=n_chunks, fd_sets=1)
    big = max(t.numel() * t.element_size() for t in (eng.stem_out, eng.plan.blocks[0].out))
    patches, yd = stem_patches(x.cuda(), eng.plan.stem, torch.float32), y.cuda()
    res = []
    for bs in (0.0, 0.5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, correct, sq = eng.full_gradient(patches, yd, 0.1, block_strength=bs)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        loss, correct, sq = eng.full_gradient(patches, yd, 0.1, block_strength=bs)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res.append((loss.cpu(), sq.cpu(), eng.avg.cpu().double(), eng.mean_tab[0, :G].cpu().clone()))
        print(f"G={G} block_strength={bs}: {1000 * dt:.1f} ms per evaluation of {n_chunks} chunks ({chunk * n_chunks / dt:.0f} images/s); largest tensor {big / 2**30:.2f} GiB; "
              f"mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    out[G] = res
    del eng, patches, model
    gc.collect(); torch.cuda.empty_cache()
a, b = out[n_chunks // 2], out[n_chunks]
for i, bs in enumerate((0.0, 0.5)):
    print(f"block_strength={bs}: losses equal {torch.equal(a[i][0], b[i][0])}; chunk sqnorm max rel {float(((a[i][1] - b[i][1]).abs() / a[i][1]).max()):.2e}; "
          f"mean gradient rel L2 {float((a[i][2] - b[i][2]).norm() / a[i][2].norm()):.2e}; finite {bool(torch.isfinite(b[i][2]).all())}; "
          f"stats of the first half equal {torch.equal(a[i][3], b[i][3][:n_chunks // 2])}")
