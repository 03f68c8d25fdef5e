"""Experiment: the float64 oracle (plain torch ops) with its tensors on the GPU vs on the host: agreement and time."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fullbatchtraining_amd.cfg import compose
from fullbatchtraining_amd.models import construct_model
from oracle import fb_oracle as orc
from tests.helpers import make_data

for depth, stem, pixels, chunk in ((18, "CIFAR", 32, 128), (50, "standard", 64, 32)):
    cfg = compose([f"model=resnet{depth}", f"model.stem={stem}"])
    torch.manual_seed(0)
    model = construct_model(cfg.model, 3, 10)
    x, y = make_data(chunk, pixels)
    spec = orc.Spec(depth, stem=stem)
    res = {}
    for dev in ("cpu", "cuda", "cuda"):
        state = {k: (v.clone().double() if v.is_floating_point() else v.clone()).to(dev) for k, v in model.state_dict().items()}
        params, buffers = orc.split_state(state)
        t0 = time.perf_counter()
        g, loss, correct = orc.chunk_gradient(spec, params, buffers, x.double().to(dev), y.to(dev))
        reg = orc.gradreg(spec, params, buffers, [t.clone() for t in g], x.double().to(dev), y.to(dev), 0.1, 0.5, 1e-2, "forward-differences")
        if dev == "cuda":
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[dev] = (torch.cat([t.reshape(-1).cpu() for t in g]), torch.cat([t.reshape(-1).cpu() for t in reg]), float(loss))
        print(f"resnet{depth} {pixels}px chunk {chunk} on {dev}: {dt:.2f} s, loss {float(loss):.12f}", flush=True)
    for i, name in enumerate(("raw", "regularised")):
        a, b = res["cpu"][i], res["cuda"][i]
        print(f"  {name}: cuda-vs-cpu rel {float((a - b).norm() / a.norm()):.2e}")
