"""GradRegularizer drop-in object (reference fullbatch/models/modules.py:136-348, standalone use per README) on the GPU:
regularised chunk gradients vs the REAL reference's vectors (tests/golden), for all three finite-difference variants."""
import numpy as np
import pytest
import torch

from tests.helpers import make_data, oracle_state, rel_err, summarise, to_oracle

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
    params, buffers = oracle_state(model)
    spec = orc.Spec(18)
    model = model.cuda()
    optimizer = torch.optim.SGD(model.parameters(), lr=0.1)
    greg = GradRegularizer(model, optimizer, torch.nn.CrossEntropyLoss(), **cfg.hyp.grad_reg, mixed_precision=False)
    assert greg.create_graph is False
    for k in range(2):
        xk, yk = x[k * chunk:(k + 1) * chunk], y[k * chunk:(k + 1) * chunk]
        xo, yo = to_oracle(xk, yk)
        raw, _, _ = orc.chunk_gradient(spec, params, buffers, xo, yo)          # also advances the oracle's BN buffers
        grads = [g.float().cuda() for g in raw]
        # keep the oracle's parameters/buffers in step with what the reference probe did (its own FD pass)
        orc.gradreg(spec, params, buffers, [g.clone() for g in raw], xo, yo, 0.1, cfg.hyp.grad_reg.block_strength,
                    cfg.hyp.grad_reg.eps, cfg.hyp.grad_reg.implementation)
        out = greg(grads, xk.cuda(), yk.cuda(), None)
        assert out is grads
        per, samp = summarise([g.cpu() for g in grads])
        err = rel_err(samp, data[f"{name}@f64/chunk{k}_reg_sample"])
        noise = rel_err(data[f"{name}/chunk{k}_reg_sample"], data[f"{name}@f64/chunk{k}_reg_sample"])
        # the reference's own fp32 run against its float64 run, on either chunk of the scenario: the floor any fp32 evaluation of this quotient sits on
        floor = max(rel_err(data[f"{name}/chunk{j}_reg_sample"], data[f"{name}@f64/chunk{j}_reg_sample"]) for j in range(2))
        print(f"{name} chunk {k}: GradRegularizer-vs-ref64 {err:.2e} (reference fp32-vs-f64 {noise:.2e}, worst chunk {floor:.2e})")
        assert err < max(3 * noise, floor, 2e-2)
        if k == 0:
            # load the model with the reference-equivalent state for the next chunk (BN buffers advanced by both passes)
            pass
    with pytest.raises(ValueError):
        GradRegularizer(model, optimizer, None, block_strength=0.5, implementation="finite_diff")
    with pytest.raises(NotImplementedError):
        GradRegularizer(model, optimizer, None, block_strength=0.5, implementation="autograd")([], None, None, None)


@pytest.mark.parametrize("implementation,block_strength", [("forward-differences", 0.5), ("central-differences", 0.0)])
def test_gradreg_object_with_pre_grads(implementation, block_strength):
    """acc_strength / pre_grads (reference modules.py:217-221, 273-275): the finite-difference direction becomes
    block_strength*g + acc_strength*pre_grads.  Object under test vs the float64 oracle (itself pinned to the reference's
    acc_strength runs, tests/test_oracle_golden.py::test_training_float64_pin[fb_acc*])."""
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.models import construct_model
    from fullbatchtraining_amd.modules import GradRegularizer
    from oracle import fb_oracle as orc

    cfg = compose(["data.pixels=16"])
    torch.manual_seed(21)
    model = construct_model(cfg.model, 3, 10)
    x, y = make_data(64, 16)
    params, buffers = oracle_state(model)
    spec = orc.Spec(18)
    xo, yo = to_oracle(x, y)
    raw, _, _ = orc.chunk_gradient(spec, params, buffers, xo[:32], yo[:32])
    pre, _, _ = orc.chunk_gradient(spec, params, buffers, xo[32:], yo[32:])       # any other gradient-shaped list
    want = orc.gradreg(spec, params, buffers, [g.clone() for g in raw], xo[:32], yo[:32], 0.1, block_strength, 1e-2, implementation,
                       acc_strength=0.25, pre_grads=pre)
    # the model on the GPU: parameters as at init, BN buffers as the oracle had them BEFORE its FD pass (after the two first passes)
    model = model.cuda()
    optimizer = torch.optim.SGD(model.parameters(), lr=0.1)
    greg = GradRegularizer(model, optimizer, torch.nn.CrossEntropyLoss(), block_strength=block_strength, acc_strength=0.25, eps=1e-2,
                           implementation=implementation)
    grads = [g.float().cuda() for g in raw]
    out = greg(grads, x[:32].cuda(), y[:32].cuda(), [g.float().cuda() for g in pre])
    assert out is grads
    a = torch.cat([g.reshape(-1).double().cpu() for g in grads])
    t = torch.cat([g.reshape(-1).cpu() for g in want])
    err = float((a - t).norm() / t.norm())
    print(f"{implementation} with pre_grads: GradRegularizer-vs-oracle64 {err:.2e}")
    assert err < 2e-2          # same class as the pre_grads-free variants above (fp32 finite differences of a cancelling sum)
    if block_strength != 0:      # (with block_strength 0 and no pre_grads the direction is 0 and eps_n infinite -- in the reference too)
        grads0 = [g.float().cuda() for g in raw]
        greg(grads0, x[:32].cuda(), y[:32].cuda(), None)
        assert float((torch.cat([g.reshape(-1) for g in grads0]) - torch.cat([g.reshape(-1) for g in grads])).norm()) > 0
