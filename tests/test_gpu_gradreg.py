"""GradRegularizer drop-in object (reference fullbatch/models/modules.py:136-348, standalone use per README) on the GPU:
regularised chunk gradients vs the REAL reference's vectors (tests/golden), for all three finite-difference variants."""
import numpy as np
import pytest
import torch

from tests.helpers import make_data, rel_err, summarise

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["fb_gradreg", "fb_central", "fb_legacy"])
def test_gradreg_object_matches_reference(golden, name):
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.models import construct_model
    from fullbatchtraining_amd.modules import GradRegularizer
    from oracle import fb_oracle as orc

    data, meta = golden
    sc = meta["scenarios"][name]
    cfg = compose(sc["overrides"] + [f"data.pixels={sc['pixels']}"])
    torch.manual_seed(sc["model_seed"])
    model = construct_model(cfg.model, 3, 10)
    x, y = make_data(sc["n"], sc["pixels"])
    chunk = min(cfg.data.batch_size, cfg.hyp.sub_batch)
    # raw chunk gradients from the float64 oracle (the object under test is the regulariser, not the first pass)
    state = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    params, buffers = orc.split_state(state)
    spec = orc.Spec(18)
    model = model.cuda()
    optimizer = torch.optim.SGD(model.parameters(), lr=0.1)
    greg = GradRegularizer(model, optimizer, torch.nn.CrossEntropyLoss(), **cfg.hyp.grad_reg, mixed_precision=False)
    assert greg.create_graph is False
    for k in range(2):
        xk, yk = x[k * chunk:(k + 1) * chunk], y[k * chunk:(k + 1) * chunk]
        raw, _, _ = orc.chunk_gradient(spec, params, buffers, xk.double(), yk)          # also advances the oracle's BN buffers
        grads = [g.float().cuda() for g in raw]
        # keep the oracle's parameters/buffers in step with what the reference probe did (its own FD pass)
        orc.gradreg(spec, params, buffers, [g.clone() for g in raw], xk.double(), yk, 0.1, cfg.hyp.grad_reg.block_strength,
                    cfg.hyp.grad_reg.eps, cfg.hyp.grad_reg.implementation)
        out = greg(grads, xk.cuda(), yk.cuda(), None)
        assert out is grads
        per, samp = summarise([g.cpu() for g in grads])
        err = rel_err(samp, data[f"{name}@f64/chunk{k}_reg_sample"])
        noise = rel_err(data[f"{name}/chunk{k}_reg_sample"], data[f"{name}@f64/chunk{k}_reg_sample"])
        print(f"{name} chunk {k}: GradRegularizer-vs-ref64 {err:.2e} (reference fp32-vs-f64 {noise:.2e})")
        assert err < max(3 * noise, 2e-2)
        if k == 0:
            # load the model with the reference-equivalent state for the next chunk (BN buffers advanced by both passes)
            pass
    with pytest.raises(ValueError):
        GradRegularizer(model, optimizer, None, block_strength=0.5, implementation="finite_diff")
    with pytest.raises(NotImplementedError):
        GradRegularizer(model, optimizer, None, block_strength=0.5, implementation="autograd")([], None, None, None)
