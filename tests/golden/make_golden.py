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
# Round 2 (files scenarios_r2.npz / meta_r2.json, `--r2`):
#  * hyp.batch_clip (training.py:138-139, 166-167 + training/utils.py:4-19).  The reference's own `_record_stats` dies on it with
#    `NameError: clipped_batches` (training.py:118 reads a local of another closure); the harness gives the name a module-level value so
#    that the run -- and with it the clip arithmetic the reference really executes -- goes through; that artefact stat is not recorded.
#  * chunk sizes that are not a multiple of the statistics-block size of the HIP kernels (data.batch_size=25 at 16 px: 100 pixels per
#    chunk on the 2x2 maps): the all-50 000-images variant data.batch_size=125 of SURVEY 8(d) in small.
#  * Bottleneck + finite differences: ResNet-50, 'standard' stem, 64 px (BASELINE config 5 in small).
SCENARIOS_R2 = {
    "fb_batchclip": (128, 16, ["hyp=fb1", "hyp.batch_clip=11.6", "hyp.steps=3", "hyp.warmup=0", "data.batch_size=32", "hyp.sub_batch=32"], 43),
    "fb_batchclip_gradreg": (128, 16, ["hyp=fb1", "hyp.batch_clip=7.8", "hyp.grad_reg.block_strength=0.5", "hyp.steps=2", "hyp.warmup=0",
                                       "data.batch_size=64", "hyp.sub_batch=64"], 45),
    "fb_ragged": (100, 16, ["hyp=fbclip", "hyp.steps=3", "hyp.warmup=0", "data.batch_size=25", "hyp.sub_batch=25"], 47),
    "fb_ragged_gradreg": (150, 16, ["hyp=fb1", "hyp.grad_reg.block_strength=0.5", "hyp.steps=2", "hyp.warmup=0", "data.batch_size=50",
                                    "hyp.sub_batch=25"], 49),
    "fb_r50_gradreg": (64, 64, ["hyp=fb1", "model=resnet50", "model.stem=standard", "hyp.grad_reg.block_strength=0.5", "hyp.steps=2",
                                "hyp.warmup=0", "hyp.optim.lr=0.02", "data.batch_size=32", "hyp.sub_batch=32"], 51),
}
# Round 3 (files scenarios_r3.npz / meta_r3.json, `--r3`): the benchmark's REAL shapes -- ResNet-18, 32 x 32 inputs, chunks of 128 -- with more
# chunks than any earlier scenario (8 chunks = 1024 images, three chunk groups in the engine test), plain + clip + warm-up, and the regulariser
# on 4 chunks.  (The earlier scenarios run 16-pixel inputs and at most 512 images: the full-size workload met the reference only in bench.py.)
SCENARIOS_R3 = {
    "fb_real_clip": (1024, 32, ["hyp=fbclip", "hyp.steps=3", "hyp.warmup=1", "data.batch_size=128", "hyp.sub_batch=128"], 53),
    "fb_real_gradreg": (512, 32, ["hyp=fb1", "hyp.steps=2", "hyp.warmup=0", "hyp.grad_reg.block_strength=0.5", "data.batch_size=128",
                                  "hyp.sub_batch=128"], 57),
}
# (files scenarios_r3b.npz / meta_r3b.json, `--r3b`): the all-50 000-images variant of the benchmark (bench.py configs.k400) at its real shape --
# chunks of 125 images at 32 x 32 (stored padded to 128 by the engine), clip + warm-up
SCENARIOS_R3B = {
    "fb_real_k125": (500, 32, ["hyp=fbclip", "hyp.steps=2", "hyp.warmup=1", "data.batch_size=125", "hyp.sub_batch=125"], 59),
}
ALL_SCENARIOS = {**SCENARIOS, **SCENARIOS_EXTRA, **SCENARIOS_N4, **SCENARIOS_A9, **SCENARIOS_R2, **SCENARIOS_R3, **SCENARIOS_R3B}


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
    if cfg.hyp.batch_clip is not None:      # see SCENARIOS_R2: lets the reference get past its own NameError
        fullbatch.training.training.clipped_batches = float("nan")
    stats = fullbatch.training.train(model, trainloader, validloader, setup, cfg)
    keys = sorted(k for k in stats if k not in ("train_time", "clipped_batches"))
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


def clip_list_vectors(fullbatch, out):
    """`_clip_gradient_list` (training/utils.py:4-19) called directly on a seeded gradient list, for the norms it supports."""
    from fullbatchtraining_amd.cfg import AttrDict
    gen = torch.Generator().manual_seed(77)
    base = [torch.randn(7, 5, generator=gen), torch.randn(11, generator=gen) * 3, torch.randn(2, 3, 3, 3, generator=gen) * 0.1]
    out["clip_list/input"] = torch.cat([t.reshape(-1) for t in base]).numpy()
    for norm in (2.0, 1.0, float("inf")):
        for clip in (0.5, 1e3):
            grads = [t.clone() for t in base]
            hit = fullbatch.training.utils._clip_gradient_list(grads, clip, AttrDict(hyp=AttrDict(grad_clip_norm=norm)))
            out[f"clip_list/p{norm}/clip{clip}"] = torch.cat([t.reshape(-1) for t in grads]).numpy()
            out[f"clip_list/p{norm}/clip{clip}/hit"] = np.array([hit])


def checkpoint_interchange(fullbatch, compose, out):
    """File-level checkpoint interchange (training/utils.py:43-70), both directions, at the real ResNet-18 size: a file WRITTEN BY THE
    REFERENCE is loaded by this package's `_load_from_checkpoint`, a file written by this package's `_save_to_checkpoint` is loaded
    by the reference's `_load_from_checkpoint` into the reference's own model / optimizer / scheduler objects.  The files are 90 MB
    each and are not committed; what is recorded is the outcome (maximum absolute differences, which must be exactly 0)."""
    from fullbatchtraining_amd import training as mine
    from fullbatchtraining_amd.models import construct_model as my_construct

    cfg = compose(["hyp=gradreg"])
    tmp = tempfile.mkdtemp()

    class Counter:
        step = 0

    def advance(model, optimizer, scheduler, n):
        gen = torch.Generator().manual_seed(5)
        for _ in range(n):
            for p in model.parameters():
                p.grad = torch.randn(p.shape, generator=gen) * 1e-2
            optimizer.step()
            scheduler.step()

    def state_diff(a, b):
        worst = 0.0
        for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert ka == kb
            worst = max(worst, float((va.double() - vb.double()).abs().max()))
        return worst

    def momentum_diff(oa, ob):
        sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
        assert sa.keys() == sb.keys() and len(sa) == 62
        return max(float((sa[k]["momentum_buffer"] - sb[k]["momentum_buffer"]).abs().max()) for k in sa)

    # reference writes ...
    torch.manual_seed(0)
    ref_model = fullbatch.models.construct_model(cfg.model, 3, 10)
    ref_opt, ref_sched = fullbatch.training.optimizers.optim_interface(ref_model, cfg.hyp)
    advance(ref_model, ref_opt, ref_sched, 3)
    Counter.step = 3
    f_ref = os.path.join(tmp, "written_by_reference.pth")
    fullbatch.training.utils._save_to_checkpoint(ref_model, ref_opt, ref_sched, None, Counter, file=f_ref)
    # ... this package loads
    torch.manual_seed(123)
    my_model = my_construct(cfg.model, 3, 10)
    my_opt, my_sched = mine.optim_interface(my_model, cfg.hyp)

    class C2:
        step = 0

    mine._load_from_checkpoint(my_model, my_opt, my_sched, None, C2, cfg.hyp.steps, device="cpu", file=f_ref)
    res = dict(we_load_reference=dict(step=C2.step, state_maxabs=state_diff(my_model, ref_model), momentum_maxabs=momentum_diff(my_opt, ref_opt),
                                      lr=[my_opt.param_groups[0]["lr"], ref_opt.param_groups[0]["lr"]],
                                      sched_last_epoch=[my_sched.last_epoch, ref_sched.last_epoch]))
    # this package writes (after two more steps) ...
    advance(my_model, my_opt, my_sched, 2)
    C2.step = 5
    f_mine = os.path.join(tmp, "written_by_package.pth")
    mine._save_to_checkpoint(my_model, my_opt, my_sched, None, C2, file=f_mine)
    # ... the reference loads into fresh reference objects
    torch.manual_seed(321)
    ref2 = fullbatch.models.construct_model(cfg.model, 3, 10)
    ref2_opt, ref2_sched = fullbatch.training.optimizers.optim_interface(ref2, cfg.hyp)

    class C3:
        step = 0

    fullbatch.training.utils._load_from_checkpoint(ref2, ref2_opt, ref2_sched, None, C3, cfg.hyp.steps, device="cpu", file=f_mine)
    res["reference_loads_ours"] = dict(step=C3.step, state_maxabs=state_diff(ref2, my_model), momentum_maxabs=momentum_diff(ref2_opt, my_opt),
                                       lr=[ref2_opt.param_groups[0]["lr"], my_opt.param_groups[0]["lr"]],
                                       sched_last_epoch=[ref2_sched.last_epoch, my_sched.last_epoch])
    # top-level structure of the two files
    a, b = torch.load(f_ref, weights_only=False), torch.load(f_mine, weights_only=False)
    res["same_structure"] = bool(len(a) == len(b) == 5 and list(a[1]) == list(b[1]) and a[0]["param_groups"][0].keys() == b[0]["param_groups"][0].keys()
                                 and a[2].keys() == b[2].keys() and a[3] is None and b[3] is None)
    # a small slice of the reference-written file travels as data: the sampled model state + momentum (strided) and the scalars
    per, samp = summarise([v.double() for v in a[1].values()])
    out["ckpt_ref/model_sample"] = samp
    out["ckpt_ref/momentum_sample"] = summarise([a[0]["state"][k]["momentum_buffer"] for k in sorted(a[0]["state"])])[1]
    out["ckpt_ref/scalars"] = np.array([a[4], a[0]["param_groups"][0]["lr"], a[2]["last_epoch"]], dtype=np.float64)
    return res


def lmdb_vectors(fullbatch, out):
    """the record databases of `fullbatch/data/lmdb_datasets.py`.  The third-party `lmdb` package is not installed here, so
    the reference's writer (`_create_database`) and reader (`LMDBDataset`) run against a dict-backed stand-in for the small part of
    its API they use (tests/helpers.py::DictLMDB), plus minimal stand-ins for the three torchvision transform classes they touch.
    What is recorded: every key/value pair the reference WROTE (CHW and HWC layouts, one and two `rounds`), the base images / labels,
    and items the reference's own reader returned (cursor and get access)."""
    from PIL import Image

    from fullbatchtraining_amd.cfg import AttrDict
    from tests.helpers import DictLMDB

    import importlib
    mod = importlib.import_module("fullbatch.data.lmdb_datasets")      # lazily imported by the reference as well (data_preparation.py:27)

    class ToTensor:
        def __call__(self, pic):
            return torch.from_numpy(np.asarray(pic, dtype=np.uint8).copy()).permute(2, 0, 1).to(torch.float) / 255

    class PILToTensor:
        def __call__(self, pic):
            return torch.from_numpy(np.asarray(pic, dtype=np.uint8).copy()).permute(2, 0, 1)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

        def __call__(self, t):
            return (t - self.mean) / self.std

    class Compose:
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    for tv in (sys.modules["torchvision"].transforms, sys.modules["torchvision.transforms"], mod.torchvision.transforms):
        tv.ToTensor, tv.PILToTensor, tv.Normalize, tv.Compose = ToTensor, PILToTensor, Normalize, Compose
    mod.lmdb = DictLMDB
    identity = Compose([])
    mod._parse_data_augmentations = lambda cfg, PIL_only=False: (identity, identity)      # rounds > 1 re-write the base images un-augmented

    class Base(torch.utils.data.Dataset):                   # a torchvision-style dataset: PIL image in, `transform` applied
        classes = list(range(10))

        def __init__(self, images, labels, transform):
            self.images, self.labels, self.transform = images, labels, transform

        def __len__(self):
            return len(self.labels)

        def __getitem__(self, i):
            return self.transform(Image.fromarray(self.images[i])), self.labels[i]

    rng = np.random.default_rng(5)
    n = 23
    images = rng.integers(0, 256, size=(n, 32, 32, 3), dtype=np.uint8)
    labels = [int(v) for v in rng.integers(0, 10, size=n)]
    mean, std = [0.4914, 0.4822, 0.4465], [0.2470, 0.2435, 0.2616]
    out["lmdb/base_images"], out["lmdb/base_labels"] = images, np.array(labels)
    out["lmdb/mean_std"] = np.array([mean, std])
    threads = torch.get_num_threads()
    torch.set_num_threads(1)                                 # the writer sizes its DataLoader workers from it
    try:
        for tag, tfs, rounds in (("chw_r1", [ToTensor(), Normalize(mean, std)], 1), ("hwc_r2", [Normalize(mean, std)], 2)):
            path = os.path.join(tempfile.mkdtemp(), f"{tag}.lmdb")
            cfg_db = AttrDict(path=os.path.dirname(path), temporary_database=False, rounds=rounds, first_round_clean=True, shuffle_while_writing=False,
                              augmentations_train=None, augmentations_val=None, rebuild_existing_database=False, access="cursor", max_readers=8,
                              readahead=False, meminit=False, max_spare_txns=8, write_frequency=7, pixels=32, mean=mean, normalize=True)
            ds = Base(images, labels, Compose(list(tfs)))
            chw = isinstance(tfs[0], ToTensor)
            if chw:                                          # the reference's own constructor: chooses the path, writes, re-opens, reads
                reader = mod.LMDBDataset(ds, cfg_db, name="train", can_create=True)
                store = DictLMDB._stores[str(reader.path)]
                items = [reader[i] for i in (0, 1, 11, 22, 5)]                 # cursor access, out of order
                reader.access = "get"
                items += [reader[i] for i in (3, 22)]
                out[f"lmdb/{tag}/item_index"] = np.array([0, 1, 11, 22, 5, 3, 22])
                out[f"lmdb/{tag}/item_images"] = np.stack([it[0].numpy() for it in items])
                out[f"lmdb/{tag}/item_labels"] = np.array([it[1] for it in items])
            else:                                            # HWC records (a dataset whose first transform is not ToTensor): the writer only
                mod._create_database(ds, path, cfg_db, db_channels_first=False, name="train")
                store = DictLMDB._stores[str(path)]
            keys = sorted(store)
            out[f"lmdb/{tag}/keys"] = np.array([k.decode("latin1") for k in keys])
            out[f"lmdb/{tag}/value_lengths"] = np.array([len(store[k]) for k in keys])
            out[f"lmdb/{tag}/values"] = np.frombuffer(b"".join(store[k] for k in keys), dtype=np.uint8)
    finally:
        torch.set_num_threads(threads)


def main_r2():
    torch.set_num_threads(8)
    fullbatch = import_reference()
    from fullbatchtraining_amd.cfg import compose

    out = {}
    lmdb_vectors(fullbatch, out)             # first: its DataLoader forks a worker, which must happen before the OpenMP pool has been used
    for name in SCENARIOS_R2:
        run_scenario(fullbatch, compose, name, out)
        run_scenario(fullbatch, compose, name, out, dtype=torch.double)
    clip_list_vectors(fullbatch, out)
    meta = dict(scenarios={k: dict(n=v[0], pixels=v[1], overrides=v[2], model_seed=v[3]) for k, v in SCENARIOS_R2.items()})
    meta["checkpoint_interchange"] = checkpoint_interchange(fullbatch, compose, out)
    print(meta["checkpoint_interchange"])
    np.savez_compressed(os.path.join(HERE, "scenarios_r2.npz"), **out)
    with open(os.path.join(HERE, "meta_r2.json"), "w") as handle:
        json.dump(meta, handle, indent=1)
    print("wrote", os.path.join(HERE, "scenarios_r2.npz"), os.path.join(HERE, "meta_r2.json"))


def main_r3():
    torch.set_num_threads(8)
    fullbatch = import_reference()
    from fullbatchtraining_amd.cfg import compose

    out = {}
    for name in SCENARIOS_R3:
        run_scenario(fullbatch, compose, name, out)
        run_scenario(fullbatch, compose, name, out, dtype=torch.double)
    meta = dict(scenarios={k: dict(n=v[0], pixels=v[1], overrides=v[2], model_seed=v[3]) for k, v in SCENARIOS_R3.items()})
    np.savez_compressed(os.path.join(HERE, "scenarios_r3.npz"), **out)
    with open(os.path.join(HERE, "meta_r3.json"), "w") as handle:
        json.dump(meta, handle, indent=1)
    print("wrote", os.path.join(HERE, "scenarios_r3.npz"), os.path.join(HERE, "meta_r3.json"))


def main_r3b():
    torch.set_num_threads(8)
    fullbatch = import_reference()
    from fullbatchtraining_amd.cfg import compose

    out = {}
    for name in SCENARIOS_R3B:
        run_scenario(fullbatch, compose, name, out)
        run_scenario(fullbatch, compose, name, out, dtype=torch.double)
    meta = dict(scenarios={k: dict(n=v[0], pixels=v[1], overrides=v[2], model_seed=v[3]) for k, v in SCENARIOS_R3B.items()})
    np.savez_compressed(os.path.join(HERE, "scenarios_r3b.npz"), **out)
    with open(os.path.join(HERE, "meta_r3b.json"), "w") as handle:
        json.dump(meta, handle, indent=1)
    print("wrote", os.path.join(HERE, "scenarios_r3b.npz"), os.path.join(HERE, "meta_r3b.json"))


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
    if "--r3b" in sys.argv:
        main_r3b()
    elif "--r3" in sys.argv:
        main_r3()
    elif "--r2" in sys.argv:
        main_r2()
    elif "--a9" in sys.argv:
        main_a9()
    else:
        main_n4() if "--n4" in sys.argv else (main_extra() if "--extra" in sys.argv else main())
