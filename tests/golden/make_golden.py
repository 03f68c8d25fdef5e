"""Generate golden vectors by running the REAL reference (``/root/reference``) on CPU.

Runs only in the build container (the reference never travels to the GPU box).  It follows the harness recipe of
SURVEY.md Appendix A: stub the absent optional imports, restore torch-1.9 foreach semantics, build cfg
with this repo's composer, call ``fullbatch.models.construct_model`` and ``fullbatch.training.train`` directly.

Outputs (committed, data only -- tensors, scalars, key lists):
  tests/golden/scenarios.npz      per-scenario stats, per-chunk internals, sampled/summarised parameter state
  tests/golden/meta.json          LR sequences, state_dict key/shape lists, checkpoint structure, scenario table

Usage:  python tests/golden/make_golden.py
"""
import json
import logging.config  # noqa: F401  (reference utils.get_log uses logging.config without importing it)
import os
import sys
import tempfile
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

SAMPLE_STRIDE = 997  # every 997th element of each flat tensor is kept verbatim
NOISE_SEED = 4242


def import_reference():
    for name in ["torchvision", "torchvision.transforms", "torchvision.datasets", "torchvision.datasets.utils",
                 "torchvision.models", "torchvision.models.densenet", "hydra", "hydra.core",
                 "hydra.core.hydra_config", "hydra.utils", "omegaconf", "lmdb"]:
        sys.modules[name] = mock.MagicMock(name=name)

    class _DL(torch.nn.Module):
        pass

    sys.modules["torchvision"].models.densenet._DenseLayer = _DL
    sys.modules["torchvision.models.densenet"]._DenseLayer = _DL
    sys.modules["omegaconf"].OmegaConf.to_container = lambda c, resolve=True: c
    for n in ["_foreach_add_", "_foreach_sub_", "_foreach_div_", "_foreach_mul_"]:
        setattr(torch, n, torch.no_grad()(getattr(torch, n)))
    sys.path.insert(0, "/root/reference")
    import fullbatch  # noqa

    return fullbatch


def summarise(tensors):
    """Per-tensor (sum, sqnorm, absmax) + a strided sample of the concatenation."""
    flat = torch.cat([t.detach().reshape(-1).double() for t in tensors])
    per = np.array([[float(t.double().sum()), float(t.double().pow(2).sum()), float(t.abs().max())] for t in tensors])
    return per, flat[::SAMPLE_STRIDE].numpy()


def make_data(n, pixels=32, classes=10, seed=1234):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, pixels, pixels, generator=gen)
    y = torch.randint(0, classes, (n,), generator=gen)
    return x, y


def loaders(x, y, batch, shuffle=False):
    ds = torch.utils.data.TensorDataset(x, y)
    if shuffle:            # what the reference's data preparation builds for hyp.shuffle=True: a RandomSampler, a new permutation per pass
        own = torch.Generator().manual_seed(0)
        train = torch.utils.data.DataLoader(ds, batch_size=min(batch, len(ds)), shuffle=True, drop_last=True, generator=own)
        valid = torch.utils.data.DataLoader(ds, batch_size=min(batch, len(ds)), shuffle=False, drop_last=False, generator=own)
        train.sampler.set_epoch = lambda *a, **k: None
        return train, valid
    sampler = torch.utils.data.SequentialSampler(ds)
    sampler.set_epoch = lambda *a, **k: None
    # the loaders get a generator of their own: a DataLoader iterator draws its base seed at the start of every pass, and with the
    # default generator that draw would sit between the seeding and the gradient-noise draws of the fb_noise scenario
    own = torch.Generator().manual_seed(0)
    train = torch.utils.data.DataLoader(ds, batch_size=min(batch, len(ds)), sampler=sampler, drop_last=True, generator=own)
    valid = torch.utils.data.DataLoader(ds, batch_size=min(batch, len(ds)), shuffle=False, drop_last=False, generator=own)
    return train, valid


SCENARIOS = {
    # name: (N, pixels, overrides, model_seed)
    "fb_plain": (512, 32, ["hyp=fb1", "hyp.steps=2", "hyp.warmup=0"], 0),
    "fb_gradreg": (512, 32, ["hyp=fb1", "hyp.steps=2", "hyp.warmup=0", "hyp.grad_reg.block_strength=0.5"], 0),
    "fb_clip_warm": (256, 32, ["hyp=fbclip", "hyp.steps=3", "hyp.warmup=2", "data.batch_size=64", "hyp.sub_batch=64"], 3),
    "fb_gradreg_c32": (128, 32, ["hyp=gradreg", "hyp.steps=3", "hyp.warmup=1", "data.batch_size=32", "hyp.sub_batch=32"], 5),
    "fb_central": (128, 16, ["hyp=fb1", "hyp.steps=2", "hyp.warmup=0", "hyp.grad_reg.block_strength=0.5",
                             "hyp.grad_reg.implementation=central-differences", "data.batch_size=64",
                             "hyp.sub_batch=64"], 7),
    "fb_legacy": (128, 16, ["hyp=fb1", "hyp.steps=2", "hyp.warmup=0", "hyp.grad_reg.block_strength=0.5",
                            "hyp.grad_reg.implementation=forward-differences-legacy", "data.batch_size=64",
                            "hyp.sub_batch=64"], 7),
}


# Scenarios added after the first fixture set was committed; they live in their own files (scenarios_extra.npz, meta_extra.json,
# written by `python tests/golden/make_golden.py --extra`) so that the original vectors stay byte-identical.
SCENARIOS_EXTRA = {
    # acc_strength: pre-pass over the dataset + full-gradient term in the finite-difference direction (training.py:128-142,
    # modules.py:217-221 / 273-275)
    "fb_acc": (128, 16, ["hyp=fb1", "hyp.steps=2", "hyp.warmup=0", "hyp.grad_reg.block_strength=0.5", "hyp.grad_reg.acc_strength=0.25",
                         "data.batch_size=32", "hyp.sub_batch=32"], 3),
    "fb_acc_central": (128, 16, ["hyp=fb1", "hyp.steps=2", "hyp.warmup=0", "hyp.grad_reg.block_strength=0.0", "hyp.grad_reg.acc_strength=0.5",
                                 "hyp.grad_reg.implementation=central-differences", "data.batch_size=64", "hyp.sub_batch=64"], 9),
}


# optimizer wrappers that consume the full-batch closure (optimizers.py:57-67, additional_optimizers/sam.py:84-92,
# lars.py:61-94); own files again (scenarios_n4.npz, meta_n4.json, `--n4`).  SAM evaluates the closure twice per step (stats are
# recorded twice); LARS/LARC call the wrapped SGD with the closure AFTER rescaling p.grad, so the closure's fresh gradients replace
# the rescaled ones and the wrapper's only effect is the weight decay it zeroes around the step.
SCENARIOS_N4 = {
    "fb_sam": (128, 16, ["hyp=fbclip", "hyp/optim_modification=SAM", "hyp.steps=3", "hyp.warmup=1", "data.batch_size=32", "hyp.sub_batch=32"], 11),
    "fb_sam_gradreg": (128, 16, ["hyp=fb1", "hyp/optim_modification=SAM", "hyp.optim_modification.rho=0.1", "hyp.steps=2", "hyp.warmup=0",
                                 "hyp.grad_reg.block_strength=0.5", "data.batch_size=64", "hyp.sub_batch=64"], 13),
    "fb_lars": (128, 16, ["hyp=fb1", "hyp/optim_modification=LARS", "hyp.steps=3", "hyp.warmup=0", "data.batch_size=64", "hyp.sub_batch=64"], 15),
    "fb_larc": (128, 16, ["hyp=fbclip", "hyp/optim_modification=LARC", "hyp.steps=3", "hyp.warmup=0", "data.batch_size=64", "hyp.sub_batch=64"], 17),
}
# Options of the closure's gradient modification and of the evaluation that are off by default: the
# L-infinity clip (training.py:199-200), the external norm bias (training.py:188-196) and evaluation of an exponential moving
# average of the model (training.py:289-294, training/utils.py:22-29).  Files scenarios_a9.npz / meta_a9.json (`--a9`).
SCENARIOS_A9 = {
    "fb_clip_inf": (128, 16, ["hyp=fbclip", "hyp.grad_clip=0.01", "hyp.grad_clip_norm=inf", "hyp.steps=3", "hyp.warmup=0", "data.batch_size=64",
                              "hyp.sub_batch=64"], 19),
    "fb_normbias1": (128, 16, ["hyp=fbclip", "hyp.norm_bias.strength=0.01", "hyp.norm_bias.norm_type=1", "hyp.norm_bias.bias=10", "hyp.steps=3",
                               "hyp.warmup=0", "data.batch_size=64", "hyp.sub_batch=64"], 21),
    "fb_normbias2": (128, 16, ["hyp=fb1", "hyp.norm_bias.strength=1e-4", "hyp.norm_bias.norm_type=2", "hyp.norm_bias.bias=70", "hyp.steps=3",
                               "hyp.warmup=0", "hyp.optim.weight_decay=0.0", "data.batch_size=64", "hyp.sub_batch=64"], 23),
    "fb_linwd": (128, 16, ["hyp=fb1", "hyp.only_linear_layers_weight_decay=True", "hyp.optim.weight_decay=0.05", "hyp.steps=3", "hyp.warmup=0",
                           "data.batch_size=64", "hyp.sub_batch=64"], 29),
    "fb_smooth": (128, 16, ["hyp=fb1", "hyp.label_smoothing=0.1", "hyp.steps=2", "hyp.warmup=0", "hyp.grad_reg.block_strength=0.5",
                            "data.batch_size=64", "hyp.sub_batch=64"], 31),
    "fb_incorrect": (128, 16, ["hyp=fbclip", "hyp.label_smoothing=0.05", "hyp.loss_modification=incorrect-xent", "hyp.steps=3", "hyp.warmup=0",
                               "data.batch_size=64", "hyp.sub_batch=64"], 33),
    # gradient noise (training.py:212-215) draws torch.randn_like per parameter from the default generator: the generator is seeded
    # right before train() for scenarios with "noise" in their name (NOISE_SEED), and the consumers of these vectors do the same
    "fb_noise": (128, 16, ["hyp=fbclip", "hyp.grad_noise.additive=0.01", "hyp.grad_noise.multiplicative=0.1", "hyp.steps=3", "hyp.warmup=0",
                           "data.batch_size=64", "hyp.sub_batch=64"], 35),
    "fb_clip_l1": (128, 16, ["hyp=fbclip", "hyp.grad_clip=50.0", "hyp.grad_clip_norm=1", "hyp.steps=3", "hyp.warmup=0", "data.batch_size=64",
                             "hyp.sub_batch=64"], 37),
    # a shuffling train loader (hyp.shuffle=True in the reference's data preparation): chunk composition changes every step
    "fb_shuffle": (192, 16, ["hyp=fb1", "hyp.steps=3", "hyp.warmup=0", "data.batch_size=64", "hyp.sub_batch=64"], 39),
    # acc_strength with sub-chunked blocks: the pre-pass runs WHOLE blocks (BN batch = data.batch_size), the main loop sub_batch chunks
    "fb_acc_sub": (128, 16, ["hyp=fb1", "hyp.steps=2", "hyp.warmup=0", "hyp.grad_reg.block_strength=0.5", "hyp.grad_reg.acc_strength=0.25",
                             "data.batch_size=64", "hyp.sub_batch=32"], 41),
    "fb_tta": (128, 16, ["hyp=fb1", "hyp.test_time_flips=True", "hyp.steps=2", "hyp.warmup=0", "data.batch_size=64", "hyp.sub_batch=64"], 27),
    "fb_ema": (128, 16, ["hyp=fb1", "hyp.evaluate_ema=True", "hyp.eval_ema_momentum=0.6", "hyp.steps=3", "hyp.warmup=0", "data.batch_size=64",
                         "hyp.sub_batch=64"], 25),
}
ALL_SCENARIOS = {**SCENARIOS, **SCENARIOS_EXTRA, **SCENARIOS_N4, **SCENARIOS_A9}


def run_scenario(fullbatch, compose, scen, out, dtype=torch.float):
    n, pixels, overrides, mseed = ALL_SCENARIOS[scen]
    name = scen if dtype == torch.float else f"{scen}@f64"
    tmp = tempfile.mkdtemp()
    extra = ["impl.accumulation_dtype=double"] if dtype == torch.double else []  # else the f64 run accumulates/updates in fp32
    cfg = compose(overrides + extra + ["impl.validate_every_nth_step=1000", f"data.pixels={pixels}"], original_cwd=tmp,
                  name=name, seed=mseed)
    x, y = make_data(n, pixels)
    trainloader, validloader = loaders(x, y, cfg.data.batch_size, shuffle="shuffle" in scen)
    setup = dict(device=torch.device("cpu"), dtype=dtype, memory_format=torch.contiguous_format)
    x = x.to(dtype)
    torch.manual_seed(mseed)
    model = fullbatch.models.construct_model(cfg.model, 3, 10)
    model.to(**setup)
    init_state = {k: v.clone() for k, v in model.state_dict().items()}
    per, samp = summarise([v.to(dtype) for v in init_state.values()])
    out[f"{name}/init_per"], out[f"{name}/init_sample"] = per, samp

    # per-chunk internals at the initial parameters, using the reference's own model + GradRegularizer objects
    chunk = min(cfg.data.batch_size, cfg.hyp.sub_batch)
    probe = fullbatch.models.construct_model(cfg.model, 3, 10)
    probe.to(**setup)
    probe.load_state_dict(init_state)
    probe.train()
    lr_probe = 0.1
    opt = torch.optim.SGD(probe.parameters(), lr=lr_probe)
    loss_fn = torch.nn.CrossEntropyLoss()
    greg = fullbatch.models.modules.GradRegularizer(probe, opt, loss_fn, **cfg.hyp.grad_reg, mixed_precision=False)
    for k in range(2):
        xk, yk = x[k * chunk:(k + 1) * chunk], y[k * chunk:(k + 1) * chunk]
        outputs = probe(xk)
        loss = loss_fn(outputs, yk)
        correct = (outputs.argmax(dim=-1) == yk).float().sum()
        grads = torch.autograd.grad(loss, probe.parameters())
        grads = [g.clone() for g in grads]
        if k == 0 and dtype == torch.double:
            out[f"{name}/chunk0_raw_fc_weight"] = grads[-2].numpy().copy()
            out[f"{name}/chunk0_raw_stem_weight"] = grads[0].numpy().copy()
        per, samp = summarise(grads)
        out[f"{name}/chunk{k}_raw_per"], out[f"{name}/chunk{k}_raw_sample"] = per, samp
        out[f"{name}/chunk{k}_scalars"] = np.array([float(loss.detach()), float(correct), float(sum(g.pow(2).sum() for g in grads))])
        if k == 0:
            out[f"{name}/chunk0_logits"] = outputs.detach().numpy()
        grads = greg(grads, xk, yk, None)
        per, samp = summarise(grads)
        out[f"{name}/chunk{k}_reg_per"], out[f"{name}/chunk{k}_reg_sample"] = per, samp
    per, samp = summarise([v.to(dtype) for v in probe.state_dict().values()])
    out[f"{name}/probe_state_per"], out[f"{name}/probe_state_sample"] = per, samp

    if "noise" in scen:
        torch.manual_seed(NOISE_SEED)
    stats = fullbatch.training.train(model, trainloader, validloader, setup, cfg)
    keys = sorted(k for k in stats if k != "train_time")
    out[f"{name}/stat_keys"] = np.array(keys)
    for k in keys:
        out[f"{name}/stat/{k}"] = np.array(stats[k], dtype=np.float64)
    final = model.state_dict()
    per, samp = summarise([v.to(dtype) for v in final.values()])
    out[f"{name}/final_per"], out[f"{name}/final_sample"] = per, samp
    out[f"{name}/final_fc_bias"] = final["fc.bias"].numpy()
    out[f"{name}/final_stem_running_mean"] = final["stem.1.running_mean"].numpy()
    out[f"{name}/final_num_batches_tracked"] = np.array([int(final["stem.1.num_batches_tracked"])])
    print(name, {k: stats[k] for k in ("train_loss", "full_loss", "grad_norm", "train_acc")})
    return cfg, model


def lr_sequences(fullbatch, compose):
    seqs = {}
    for hyp, count in (("fb1", 300), ("fb2", 3000), ("fbclip", 3000), ("gradreg", 3000)):
        cfg = compose([f"hyp={hyp}"])
        model = torch.nn.Linear(2, 2)
        optimizer, scheduler = fullbatch.training.optimizers.optim_interface(model, cfg.hyp)
        seq = []
        for _ in range(count):
            seq.append(optimizer.param_groups[0]["lr"])
            optimizer.step()
            scheduler.step()
        seqs[hyp] = seq
    return seqs


def checkpoint_structure(fullbatch, compose):
    """Key layout of the 5-list checkpoint (reference training/utils.py:43-51) for ResNet-18 after one SGD step."""
    cfg = compose(["hyp=gradreg"])
    torch.manual_seed(0)
    model = fullbatch.models.construct_model(cfg.model, 3, 10)
    optimizer, scheduler = fullbatch.training.optimizers.optim_interface(model, cfg.hyp)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    optimizer.step()
    scheduler.step()

    class Counter:
        step = 1

    path = os.path.join(tempfile.mkdtemp(), "ck.pth")
    fullbatch.training.utils._save_to_checkpoint(model, optimizer, scheduler, None, Counter, file=path)
    optim_state, model_state, sched_state, scaler_state, step = torch.load(path, weights_only=False)

    def describe(obj):
        if torch.is_tensor(obj):
            return {"tensor": list(obj.shape), "dtype": str(obj.dtype)}
        if isinstance(obj, dict):
            return {str(k): describe(v) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return [describe(v) for v in obj] if len(obj) < 8 else {"list_len": len(obj), "first": describe(obj[0])}
        return repr(obj) if not isinstance(obj, (int, float, bool, type(None), str)) else obj

    return dict(
        model_state={k: [list(v.shape), str(v.dtype)] for k, v in model_state.items()},
        optim_state=dict(param_groups=describe(optim_state["param_groups"]),
                         state0=describe(optim_state["state"][0]), n_state=len(optim_state["state"])),
        scheduler_state=describe(sched_state), scaler_state=scaler_state, step=step,
    )


def main_extra():
    torch.set_num_threads(8)
    fullbatch = import_reference()
    from fullbatchtraining_amd.cfg import compose

    out = {}
    for name in SCENARIOS_EXTRA:
        run_scenario(fullbatch, compose, name, out)
        run_scenario(fullbatch, compose, name, out, dtype=torch.double)
    np.savez_compressed(os.path.join(HERE, "scenarios_extra.npz"), **out)
    meta = dict(scenarios={k: dict(n=v[0], pixels=v[1], overrides=v[2], model_seed=v[3]) for k, v in SCENARIOS_EXTRA.items()})
    with open(os.path.join(HERE, "meta_extra.json"), "w") as handle:
        json.dump(meta, handle, indent=1)
    print("wrote", os.path.join(HERE, "scenarios_extra.npz"), os.path.join(HERE, "meta_extra.json"))


def main_n4():
    torch.set_num_threads(8)
    fullbatch = import_reference()
    from fullbatchtraining_amd.cfg import compose

    out = {}
    for name in SCENARIOS_N4:
        run_scenario(fullbatch, compose, name, out)
        run_scenario(fullbatch, compose, name, out, dtype=torch.double)
    np.savez_compressed(os.path.join(HERE, "scenarios_n4.npz"), **out)
    meta = dict(scenarios={k: dict(n=v[0], pixels=v[1], overrides=v[2], model_seed=v[3]) for k, v in SCENARIOS_N4.items()})
    with open(os.path.join(HERE, "meta_n4.json"), "w") as handle:
        json.dump(meta, handle, indent=1)
    print("wrote", os.path.join(HERE, "scenarios_n4.npz"), os.path.join(HERE, "meta_n4.json"))


def main_a9():
    torch.set_num_threads(8)
    fullbatch = import_reference()
    from fullbatchtraining_amd.cfg import compose

    out = {}
    for name in SCENARIOS_A9:
        run_scenario(fullbatch, compose, name, out)
        run_scenario(fullbatch, compose, name, out, dtype=torch.double)
    np.savez_compressed(os.path.join(HERE, "scenarios_a9.npz"), **out)
    meta = dict(scenarios={k: dict(n=v[0], pixels=v[1], overrides=v[2], model_seed=v[3]) for k, v in SCENARIOS_A9.items()})
    with open(os.path.join(HERE, "meta_a9.json"), "w") as handle:
        json.dump(meta, handle, indent=1)
    print("wrote", os.path.join(HERE, "scenarios_a9.npz"), os.path.join(HERE, "meta_a9.json"))


def main():
    torch.set_num_threads(8)
    fullbatch = import_reference()
    from fullbatchtraining_amd.cfg import compose

    out = {}
    for name in SCENARIOS:
        run_scenario(fullbatch, compose, name, out)
        run_scenario(fullbatch, compose, name, out, dtype=torch.double)
    np.savez_compressed(os.path.join(HERE, "scenarios.npz"), **out)

    meta = dict(sample_stride=SAMPLE_STRIDE, scenarios={k: dict(n=v[0], pixels=v[1], overrides=v[2], model_seed=v[3])
                                                        for k, v in SCENARIOS.items()})
    meta["lr"] = {k: [v[i] for i in list(range(0, 12)) + list(range(395, 410)) + [1000, 2000, 2999] if i < len(v)]
                  for k, v in lr_sequences(fullbatch, compose).items()}
    meta["lr_index"] = list(range(0, 12)) + list(range(395, 410)) + [1000, 2000, 2999]
    meta["checkpoint"] = checkpoint_structure(fullbatch, compose)
    cfg152 = compose(["model=resnet152"])
    torch.manual_seed(0)
    m152 = fullbatch.models.construct_model(cfg152.model, 3, 10)
    meta["resnet152_keys"] = {k: list(v.shape) for k, v in m152.state_dict().items()}
    meta["resnet152_nparams"] = sum(p.numel() for p in m152.parameters())
    with open(os.path.join(HERE, "meta.json"), "w") as handle:
        json.dump(meta, handle, indent=1)
    print("wrote", os.path.join(HERE, "scenarios.npz"), os.path.join(HERE, "meta.json"))


if __name__ == "__main__":
    if "--a9" in sys.argv:
        main_a9()
    else:
        main_n4() if "--n4" in sys.argv else (main_extra() if "--extra" in sys.argv else main())
