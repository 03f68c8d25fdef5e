"""GPU parity of the drop-in ``train(model, trainloader, validloader, setup, cfg)`` against the REAL reference's recorded
runs (tests/golden, produced by tests/golden/make_golden.py) and against the CPU oracle.

Tolerances are tied to the reference's own fp32-vs-float64 spread on each scenario (printed by the tests):
stats of plain full-batch steps agree to ~1e-5, finite-difference scenarios to ~1e-3.
"""
import os

import numpy as np
import pytest
import torch

from tests.helpers import NOISE_SEED, make_data, rel_err, shuffling_loaders, summarise

pytestmark = pytest.mark.gpu

STAT_KEYS = ("train_loss", "train_acc", "param_norm", "grad_norm", "full_loss", "preclip_gradnorm", "clipped_step")


def _run(meta, name, extra=(), tmp_path=None, valid_full=False):
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.models import construct_model
    from fullbatchtraining_amd.training import train

    sc = meta["scenarios"][name]
    cfg = compose(sc["overrides"] + [f"data.pixels={sc['pixels']}", "impl.validate_every_nth_step=1000"] + list(extra),
                  original_cwd=str(tmp_path) if tmp_path else os.getcwd(), name=name)
    torch.manual_seed(sc["model_seed"])
    model = construct_model(cfg.model, 3, 10)
    x, y = make_data(sc["n"], sc["pixels"])
    setup = dict(device=torch.device("cuda:0"), dtype=torch.float, memory_format=torch.contiguous_format)
    if "noise" in name:
        torch.manual_seed(NOISE_SEED)          # the generator state the reference run drew its gradient noise from
    if "shuffle" in name:                      # the loaders the reference run was given (RandomSampler, shared generator)
        tl, vl = shuffling_loaders(x, y, cfg.data.batch_size)
        return cfg, model, train(model, tl, vl, setup, cfg)
    stats = train(model, (x, y), (x, y) if valid_full else (x[:64], y[:64]), setup, cfg)
    return cfg, model, stats


@pytest.mark.parametrize("name,tol,group", [("fb_plain", 2e-4, 3), ("fb_clip_warm", 2e-4, 13), ("fb_gradreg", 3e-3, 2),
                                            ("fb_gradreg_c32", 8e-3, 4), ("fb_central", 3e-3, 2), ("fb_legacy", 8e-3, 1),
                                            ("fb_acc", 8e-3, 3), ("fb_acc_central", 3e-3, 2),        # acc_strength pre-pass
                                            ("fb_acc_sub", 8e-3, 4),                                 # ... over whole blocks of 2 sub-chunks
                                            # optimizer wrappers around the closure: SAM records two closures per step
                                            ("fb_sam", 2e-3, 3), ("fb_sam_gradreg", 8e-3, 2),     # SAM + regulariser: second step on the noise floor
                                            # (3 steps without warm-up: steps 1-2 agree with the float64 run to 1e-7..1e-5, the third sits on the fp32
                                            # noise floor of the moved parameters: the last-step losses of ALL scenarios scatter between 1e-7 and 5e-3 around the float64 run
                                            # for the exact-f32 engine, the split-bf16 engine and the reference's own fp32 run alike, without order --
                                            # profiles/r2_split_vs_exact_parity.md)
                                            ("fb_lars", 5e-3, 1), ("fb_larc", 5e-3, 2),
                                            # off-by-default options of the gradient modification: L-inf clip, norm bias
                                            # (the L-inf norm is ONE gradient element: its fp32 noise, ~1e-3, scales the whole update -- the CPU oracle in fp32
                                            # lands 3.6e-3 from the reference's float64 loss at step 3 as well)
                                            ("fb_clip_inf", 1e-2, 2), ("fb_normbias1", 1e-3, 1), ("fb_normbias2", 1e-3, 2),
                                            # per-tensor weight decay (a10); lr 0.4 x wd 0.05 shrinks the weights by 2 % per step and the fp32 noise with them: the
                                            # CPU oracle in fp32 is 1.2e-3 from the float64 loss at step 3 (the option itself is checked exactly in
                                            # test_engine_per_tensor_weight_decay_matches_torch_sgd and in the float64 oracle pin)
                                            ("fb_linwd", 3e-3, 2),
                                            # loss functions of get_loss_fn (a12): label smoothing (with the regulariser), incorrect-xent
                                            ("fb_smooth", 8e-3, 2), ("fb_incorrect", 1e-2, 1),
                                            ("fb_clip_l1", 2e-4, 2),                                 # p-norm clip, p = 1
                                            # shuffling train loader (lr 0.4 without warm-up: the loss climbs 2.4 -> 3.9 -> 7.0 and fp32 noise with it; steps 1-2 agree to 3e-4)
                                            ("fb_shuffle", 1e-2, 2),
                                            # round 2: per-chunk clip (hyp.batch_clip), chunk sizes off the 128-pixel statistics grid (stored padded
                                            # with zero images), Bottleneck + finite differences (ResNet-50, standard stem, 64 px)
                                            # (3 steps at 16 px with 25..32-image chunks: step-3 losses of the 3-step scenarios above sit 1e-5..5e-3 from the
                                            # float64 run -- fb_lars 6e-4, fb_clip_inf 4e-3 -- and here the reference's own fp32 run happens to land within
                                            # 3e-6..1e-5 of it, so the 5x-spread bound does not help: steps 1-2 agree to 1e-5, step 3 to 9e-4)
                                            ("fb_batchclip", 3e-3, 2), ("fb_batchclip_gradreg", 8e-3, 2), ("fb_ragged", 3e-3, 3),
                                            ("fb_ragged_gradreg", 8e-3, 2), ("fb_r50_gradreg", 8e-3, 2),
                                            # round 3: the benchmark's real shapes (32 px, chunks of 128), 8 chunks in three chunk groups with clip + warm-up;
                                            # 4 chunks with the regulariser (bf16x6 passes: the default, reference precision; f16x2: test_train_f16x2_opt_in...)
                                            ("fb_real_clip", 2e-4, 3), ("fb_real_gradreg", 3e-3, 2),
                                            # ... and the all-50 000-images variant of the benchmark (bench.py configs.k400): chunks of 125 images at 32 x 32,
                                            # stored padded to 128 (scenarios_r3b.npz)
                                            ("fb_real_k125", 3e-3, 2)])
def test_train_matches_reference_run_f32(golden, name, tol, group, tmp_path):
    data, meta = golden
    cfg, model, stats = _run(meta, name, [f"impl.engine.chunk_group={group}"], tmp_path)
    for key in STAT_KEYS:
        if f"{name}@f64/stat/{key}" not in data:
            continue
        r64, r32 = data[f"{name}@f64/stat/{key}"], data[f"{name}/stat/{key}"]
        print(f"{name} {key}: engine {np.array(stats[key])} ref32 {r32} ref64 {r64}")
        if key == "train_acc" and ("gradreg_c32" in name or "clip_inf" in name):
            continue  # the reference's own fp32 and float64 runs disagree on the accuracy here / single-sample flips on the noise floor
        # within `tol`, or within 5x the reference's own fp32-vs-float64 spread on this statistic, whichever is larger
        bound = np.maximum(tol * np.abs(r64) + 1e-6, 5 * np.abs(r32 - r64))
        if key == "train_acc":       # one prediction may flip on the fp32 noise floor (two with the regulariser and a per-chunk clip on top)
            bound = np.maximum(bound, (2.0 if name == "fb_batchclip_gradreg" else 1.0) / meta["scenarios"][name]["n"] + 1e-9)
        if key == "preclip_gradnorm" and "clip_inf" in name:
            # max |g_i| of later steps: which element is the largest is itself decided on the noise floor (reference fp32 vs float64:
            # 2.5 %, CPU oracle fp32: 7 % at step 3); only the first step is a sharp check of fb_mt_absmax2
            bound[1:] = 0.1 * np.abs(r64[1:])
        assert np.all(np.abs(np.array(stats[key]) - r64) <= bound), (key, stats[key], r32, r64)
    if name == "fb_batchclip":       # the count the reference means to log: chunks whose (here: raw) norm exceeds the clip
        assert stats["clipped_batches"] == [sum(1 for k in range(4) if data[f"{name}@f64/stat/grad_norm_train_{k}"][s] > 11.6) for s in range(3)]
    n_chunks = len([k for k in stats if k.startswith("grad_norm_train_")])
    assert n_chunks == meta["scenarios"][name]["n"] // min(cfg.data.batch_size, cfg.hyp.sub_batch)
    for k in range(n_chunks):
        r64, r32 = data[f"{name}@f64/stat/grad_norm_train_{k}"], data[f"{name}/stat/grad_norm_train_{k}"]
        # a single chunk gradient sits on the fp32 noise floor (the reference's own fp32 chunk gradient is ~3e-3 rel-L2 from
        # its float64 run, tests/test_oracle_golden.py; any other summation order lands equally far away), so its norm is
        # only meaningful to ~1e-3 once the parameters have moved (steps >= 3); the |r32 - r64| of one sample underestimates that
        bound = np.maximum(max(tol, 1e-3) * np.abs(r64), 5 * np.abs(r32 - r64))
        assert np.all(np.abs(np.array(stats[f"grad_norm_train_{k}"]) - r64) <= bound), (k, stats[f"grad_norm_train_{k}"], r32, r64)
    # final parameters + BN buffers (state_dict order) against the reference's float64 run
    ordered = [v.double() for v in model.state_dict().values()]
    per, samp = summarise(ordered)
    err = rel_err(samp, data[f"{name}@f64/final_sample"])
    noise = rel_err(data[f"{name}/final_sample"], data[f"{name}@f64/final_sample"])
    print(f"{name}: final state engine-vs-ref64 {err:.2e} (reference fp32-vs-f64 {noise:.2e})")
    assert err < max(10 * noise, 1e-5)
    # stem running mean = mean of a zero-mean quantity (N(0,1) inputs): ill-conditioned, so judge it against the
    # reference's own fp32-vs-float64 spread on the same scenario
    rm_err = rel_err(model.state_dict()["stem.1.running_mean"].double().numpy(), data[f"{name}@f64/final_stem_running_mean"])
    rm_noise = rel_err(data[f"{name}/final_stem_running_mean"], data[f"{name}@f64/final_stem_running_mean"])
    print(f"{name}: stem running_mean engine-vs-ref64 {rm_err:.2e} (reference fp32-vs-f64 {rm_noise:.2e})")
    assert rm_err < max(10 * rm_noise, 2e-3)
    assert int(model.state_dict()["stem.1.num_batches_tracked"]) == int(data[f"{name}@f64/final_num_batches_tracked"][0])
    assert len(stats["valid_loss"]) >= 1 and np.isfinite(stats["valid_loss"][-1])
    # closure contract: p.grad populated with the last full gradient
    assert all(p.grad is not None and p.grad.shape == p.shape for p in model.parameters())


@pytest.mark.parametrize("name,tol,group", [("fb_real_gradreg", 3e-3, 2), ("fb_gradreg", 3e-3, 2), ("fb_sam_gradreg", 1.2e-2, 2)])
def test_train_f16x2_opt_in_matches_reference_run(golden, name, tol, group, tmp_path):
    """``impl.engine.fd_arithmetic=f16x2``: the regulariser's fp32 passes with 22-bit operands (two scaled fp16 pieces, three MFMAs per product, 0.6x
    the step time of the default bf16x6) against the same reference runs, same statistics; SAM + regulariser needs 1.2e-2 on |g| of its
    second step in this arithmetic (bf16x6: 8e-3)."""
    data, meta = golden
    cfg, model, stats = _run(meta, name, [f"impl.engine.chunk_group={group}", "impl.engine.fd_arithmetic=f16x2"], tmp_path)
    for key in STAT_KEYS:
        if f"{name}@f64/stat/{key}" not in data:
            continue
        r64, r32 = data[f"{name}@f64/stat/{key}"], data[f"{name}/stat/{key}"]
        bound = np.maximum(tol * np.abs(r64) + 1e-6, 5 * np.abs(r32 - r64))
        if key == "train_acc":
            bound = np.maximum(bound, 1.0 / meta["scenarios"][name]["n"] + 1e-9)
        assert np.all(np.abs(np.array(stats[key]) - r64) <= bound), (key, stats[key], r32, r64)
    err = rel_err(summarise([v.double() for v in model.state_dict().values()])[1], data[f"{name}@f64/final_sample"])
    noise = rel_err(data[f"{name}/final_sample"], data[f"{name}@f64/final_sample"])
    print(f"{name} [f16x2]: final state engine-vs-ref64 {err:.2e} (reference fp32-vs-f64 {noise:.2e})")
    assert err < max(10 * noise, 1e-5)


def test_train_ema_evaluation_matches_reference(golden, tmp_path):
    """hyp.evaluate_ema (reference training.py:289-294, training/utils.py:22-29): the validation statistics come from an EMA of
    parameters and BN buffers, the training statistics from the live model."""
    data, meta = golden
    name = "fb_ema"
    cfg, model, stats = _run(meta, name, ["impl.engine.chunk_group=2"], tmp_path, valid_full=True)
    for key in ("train_loss", "param_norm", "grad_norm", "full_loss", "valid_loss", "valid_acc"):
        r64, r32 = data[f"{name}@f64/stat/{key}"], data[f"{name}/stat/{key}"]
        print(f"{name} {key}: engine {np.array(stats[key])} ref32 {r32} ref64 {r64}")
        bound = np.maximum(1e-3 * np.abs(r64) + 1e-6, 5 * np.abs(r32 - r64))
        assert len(stats[key]) == len(r64) and np.all(np.abs(np.array(stats[key]) - r64) <= bound), (key, stats[key], r32, r64)
    # the live model (not the EMA) is what train() leaves in the container
    err = rel_err(summarise([v.double() for v in model.state_dict().values()])[1], data[f"{name}@f64/final_sample"])
    assert err < max(10 * rel_err(data[f"{name}/final_sample"], data[f"{name}@f64/final_sample"]), 1e-5)


def test_train_gradient_noise_matches_reference(golden, tmp_path):
    """hyp.grad_noise (reference training.py:212-215): additive + multiplicative noise on the clipped gradient, one randn_like per
    parameter from the default generator.  The fp32 reference run is the yardstick here (its float64 twin draws float64 noise, a
    different stream); the float64 pin of the same scenario is in tests/test_oracle_golden.py."""
    data, meta = golden
    name = "fb_noise"
    cfg, model, stats = _run(meta, name, ["impl.engine.chunk_group=2"], tmp_path)
    for key in ("train_loss", "param_norm", "grad_norm", "full_loss", "preclip_gradnorm", "clipped_step"):
        r32 = data[f"{name}/stat/{key}"]
        print(f"{name} {key}: engine {np.array(stats[key])} ref32 {r32}")
        assert np.allclose(stats[key], r32, rtol=2e-3, atol=1e-6), (key, stats[key], r32)
    err = rel_err(summarise([v.double() for v in model.state_dict().values()])[1], data[f"{name}/final_sample"])
    print(f"{name}: final state engine-vs-ref32 {err:.2e}")
    assert err < 1e-3
    # the noise is live: the noise-free run of the same configuration ends somewhere else
    quiet = [o for o in meta["scenarios"][name]["overrides"] if "grad_noise" not in o]
    assert abs(data[f"{name}/stat/train_loss"][-1] - data["fb_clip_warm/stat/train_loss"][-1]) > 0 and len(quiet) < len(meta["scenarios"][name]["overrides"])


def test_train_test_time_flips_matches_reference(golden, tmp_path):
    """hyp.test_time_flips (reference training.py:370-373): validation on softmax(image) + softmax(mirror)."""
    data, meta = golden
    name = "fb_tta"
    cfg, model, stats = _run(meta, name, ["impl.engine.chunk_group=2"], tmp_path, valid_full=True)
    for key in ("train_loss", "valid_loss", "valid_acc"):
        r64, r32 = data[f"{name}@f64/stat/{key}"], data[f"{name}/stat/{key}"]
        print(f"{name} {key}: engine {np.array(stats[key])} ref32 {r32} ref64 {r64}")
        bound = np.maximum(1e-3 * np.abs(r64) + 1e-6, 5 * np.abs(r32 - r64))
        assert len(stats[key]) == len(r64) and np.all(np.abs(np.array(stats[key]) - r64) <= bound), (key, stats[key], r32, r64)
    # the mirrored evaluation differs from the plain one (the option is live)
    from fullbatchtraining_amd.training import evaluate
    from fullbatchtraining_amd.cfg import compose
    x, y = make_data(meta["scenarios"][name]["n"], meta["scenarios"][name]["pixels"])
    setup = dict(device=torch.device("cuda:0"), dtype=torch.float, memory_format=torch.contiguous_format)
    plain = evaluate(model, (x, y), None, setup, cfg.impl, compose(["hyp=fb1"]).hyp)
    flips = evaluate(model, (x, y), None, setup, cfg.impl, cfg.hyp)
    assert abs(flips["valid_loss"][-1] - stats["valid_loss"][-1]) < 1e-5 * abs(stats["valid_loss"][-1])
    assert abs(plain["valid_loss"][-1] - flips["valid_loss"][-1]) > 1e-3


def test_train_bf16_tracks_fp32_statistics(golden, tmp_path):
    """bf16 compute path (impl.mixed_precision=True): training statistics over 2 steps vs the reference's float64 run.
    Tolerance 2e-2 on losses/norms: storage rounding 2^-9 per activation, averaged over 4 chunks of 128."""
    data, meta = golden
    cfg, model, stats = _run(meta, "fb_plain", ["impl.mixed_precision=True", "impl.engine.chunk_group=4"], tmp_path)
    for key in ("train_loss", "full_loss", "param_norm", "grad_norm"):
        r64 = data[f"fb_plain@f64/stat/{key}"]
        print(f"bf16 {key}: engine {np.array(stats[key])} ref64 {r64}")
        assert np.allclose(stats[key], r64, rtol=2e-2 if key != "grad_norm" else 0.1), key
    assert abs(stats["train_acc"][0] - data["fb_plain@f64/stat/train_acc"][0]) < 0.02


def test_train_bf16_at_the_benchmark_shape_tracks_the_reference(golden, tmp_path):
    """The arithmetic behind bench.py's headline number against a run of the REFERENCE at the benchmark's real shapes: ResNet-18, 32 x 32, 8 chunks
    of 128 (three chunk groups), clip + warm-up, 3 steps (`fb_real_clip`, tests/golden/make_golden.py --r3).  bf16 storage moves a chunk
    gradient by ~0.2 (ReLU-mask flips) and the mean of 8 chunks by ~0.1, so the gradient norm is held to 5e-2 and losses / parameter norms,
    which average over all images, to 5e-3; accuracy to two predictions."""
    data, meta = golden
    name = "fb_real_clip"
    cfg, model, stats = _run(meta, name, ["impl.mixed_precision=True", "impl.engine.chunk_group=3"], tmp_path)
    for key, tol in (("train_loss", 5e-3), ("full_loss", 5e-3), ("param_norm", 1e-3), ("grad_norm", 5e-2), ("preclip_gradnorm", 0.1)):
        r64 = data[f"{name}@f64/stat/{key}"]
        print(f"bf16 {key}: engine {np.array(stats[key])} ref64 {r64}")
        assert np.allclose(stats[key], r64, rtol=tol), key
    assert np.all(np.abs(np.array(stats["train_acc"]) - data[f"{name}@f64/stat/train_acc"]) <= 2.0 / 1024 + 1e-9)
    assert stats["clipped_step"] == list(data[f"{name}@f64/stat/clipped_step"])
    ordered = [v.double() for v in model.state_dict().values()]
    err = rel_err(summarise(ordered)[1], data[f"{name}@f64/final_sample"])
    print(f"bf16 final state vs the reference's float64 run: {err:.2e}")
    assert err < 2e-3


def test_train_bf16_chunks_of_125_at_the_benchmark_shape_track_the_reference(golden, tmp_path):
    """bench.py's configs.k400 arithmetic (bf16, chunks of 125 images stored padded to 128, 32 x 32) against a run of the REFERENCE: `fb_real_k125`
    (4 chunks of 125, clip + warm-up, 2 steps; tests/golden/make_golden.py --r3b).  Four chunks average less bf16 noise away than the eight of
    `fb_real_clip`: gradient norms to 0.1, losses and parameter norms to 5e-3 / 1e-3, accuracy to two predictions."""
    data, meta = golden
    name = "fb_real_k125"
    cfg, model, stats = _run(meta, name, ["impl.mixed_precision=True", "impl.engine.chunk_group=3"], tmp_path)
    for key, tol in (("train_loss", 5e-3), ("full_loss", 5e-3), ("param_norm", 1e-3), ("grad_norm", 0.1), ("preclip_gradnorm", 0.15)):
        r64 = data[f"{name}@f64/stat/{key}"]
        print(f"bf16 k125 {key}: engine {np.array(stats[key])} ref64 {r64}")
        assert np.allclose(stats[key], r64, rtol=tol), key
    assert np.all(np.abs(np.array(stats["train_acc"]) - data[f"{name}@f64/stat/train_acc"]) <= 2.0 / 500 + 1e-9)
    assert stats["clipped_step"] == list(data[f"{name}@f64/stat/clipped_step"])


@pytest.mark.parametrize("mixed", [False, True])
def test_full_batch_descent_fits_a_learnable_dataset(mixed, tmp_path):
    """End-to-end sanity beyond the 2-3 steps of the parity scenarios: 25 full-batch steps (clip + warm-up + Nesterov momentum, the
    fbclip recipe at a small scale) on a dataset whose labels are a function of the inputs.  f32 and bf16 both fit it, and the bf16
    trajectory stays near the f32 one."""
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.models import construct_model
    from fullbatchtraining_amd.training import train

    gen = torch.Generator().manual_seed(7)
    n, pixels = 512, 16
    protos = torch.randn(10, 3, pixels, pixels, generator=gen)
    y = torch.randint(0, 10, (n,), generator=gen)
    x = protos[y] + 0.5 * torch.randn(n, 3, pixels, pixels, generator=gen)      # class prototype + noise
    cfg = compose(["hyp=fbclip", "hyp.steps=25", "hyp.warmup=5", "hyp.optim.lr=0.1", "hyp.grad_clip=1.0", "data.batch_size=128", "hyp.sub_batch=128",
                   f"data.pixels={pixels}", "impl.validate_every_nth_step=1000", f"impl.mixed_precision={mixed}", "impl.engine.chunk_group=4"],
                  original_cwd=str(tmp_path), name="fit")
    torch.manual_seed(0)
    model = construct_model(cfg.model, 3, 10)
    setup = dict(device=torch.device("cuda:0"), dtype=torch.float, memory_format=torch.contiguous_format)
    stats = train(model, (x, y), (x, y), setup, cfg)
    loss, acc = np.array(stats["train_loss"]), np.array(stats["train_acc"])
    print(f"mixed={mixed}: loss {loss[[0, 5, 10, 15, 20, 24]]}, acc {acc[[0, 5, 10, 15, 20, 24]]}, valid_acc {stats['valid_acc']}")
    assert np.all(np.isfinite(loss)) and loss[-1] < 0.25 * loss[0] and acc[-1] > 0.95
    assert stats["valid_acc"][-1] > 0.9          # BN running statistics are usable in eval mode
    assert len(loss) == 25 and len(stats["grad_norm_train_3"]) == 25


def test_checkpoint_roundtrip_and_reference_layout(golden, tmp_path):
    """5-list checkpoint (reference training/utils.py:43-51): same structure as the reference's file, resume continues."""
    data, meta = golden
    os.makedirs(tmp_path / "checkpoints", exist_ok=True)
    cfg, model, stats = _run(meta, "fb_clip_warm", ["impl.checkpoint.name=ck.pth", "hyp.steps=2"], tmp_path)
    optim_state, model_state, sched_state, scaler_state, step = torch.load(tmp_path / "checkpoints" / "ck.pth", weights_only=False)
    ref = meta["checkpoint"]
    assert step == 2 and scaler_state is None
    assert {k: [list(v.shape), str(v.dtype)] for k, v in model_state.items()} == ref["model_state"]
    assert len(optim_state["state"]) == ref["optim_state"]["n_state"]
    assert set(optim_state["state"][0].keys()) == set(ref["optim_state"]["state0"].keys())
    assert set(optim_state["param_groups"][0].keys()) == set(ref["optim_state"]["param_groups"][0].keys())
    assert set(sched_state.keys()) == set(ref["scheduler_state"].keys())
    assert set(sched_state["after_scheduler"].keys()) == set(ref["scheduler_state"]["after_scheduler"].keys())
    # resume for the third step: must reproduce the uninterrupted 3-step run
    cfg2, model2, stats2 = _run(meta, "fb_clip_warm", ["impl.checkpoint.name=ck.pth", "hyp.steps=3"], tmp_path)
    full = data["fb_clip_warm@f64/stat/train_loss"]
    assert np.allclose(stats2["train_loss"][-1], full[2], rtol=2e-4)
    with pytest.raises(ValueError):
        _run(meta, "fb_clip_warm", ["impl.checkpoint.name=ck.pth", "hyp.steps=3"], tmp_path)
    # verify_model_checkpoint.py path: load the model state of the file into a fresh container and evaluate it with
    # the reference-signature `evaluate`: same numbers as the validation pass at the end of the run that wrote the checkpoint
    from fullbatchtraining_amd.models import construct_model
    from fullbatchtraining_amd.training import evaluate
    fresh = construct_model(cfg2.model, 3, 10)
    fresh.load_state_dict(torch.load(tmp_path / "checkpoints" / "ck.pth", weights_only=False)[1])
    sc = meta["scenarios"]["fb_clip_warm"]
    x, y = make_data(sc["n"], sc["pixels"])
    setup = dict(device=torch.device("cuda:0"), dtype=torch.float, memory_format=torch.contiguous_format)
    ev = evaluate(fresh, (x[:64], y[:64]), None, setup, cfg2.impl, cfg2.hyp)
    assert np.allclose(ev["valid_loss"][-1], stats2["valid_loss"][-1], rtol=1e-5) and ev["valid_acc"][-1] == stats2["valid_acc"][-1]


@pytest.mark.parametrize("over,clip", [(["hyp=fb1"], False), (["hyp=fbclip", "hyp.grad_clip=0.05"], True),
                                       (["hyp=gradreg", "hyp.grad_reg.block_strength=0.5", "hyp.warmup=0"], True),
                                       (["hyp=fb1", "impl.mixed_precision=True"], False)])
def test_implementation_noise_protocol_is_exactly_zero(over, clip, tmp_path):
    """SURVEY 8f N2, reference training.py:429-600: two evaluations of the full-batch gradient from the same checkpoint.  The
    reference measures a run-to-run FP error there (atomics in cuDNN wgrad); here every reduction has a fixed order, so the two
    gradients (regularised and clipped, f32 and bf16) must be bit-identical."""
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.models import construct_model
    from fullbatchtraining_amd.training import _measure_implementation_noise

    cfg = compose(over + ["data.batch_size=32", "hyp.sub_batch=32", "data.pixels=16", "impl.engine.chunk_group=3", "hyp.steps=4"],
                  original_cwd=str(tmp_path), name="noise")
    torch.manual_seed(3)
    model = construct_model(cfg.model, 3, 10)
    x, y = make_data(5 * 32, 16)
    setup = dict(device=torch.device("cuda:0"), dtype=torch.float, memory_format=torch.contiguous_format)
    out = _measure_implementation_noise(model, (x, y), None, setup, cfg)
    assert out["error_linf"] == 0.0 and out["error_l2"] == 0.0 and out["error_l1"] == 0.0
    assert out["loss"][0] == out["loss"][1] and np.isfinite(out["loss"][0]) and out["norm_l2"] > 0
    if clip:
        assert out["norm_l2"] <= float(cfg.hyp.grad_clip) * (1 + 1e-5)


def test_device_augmentation_changes_the_feed_every_step(tmp_path):
    """impl.engine.device_augment=True: every step trains on a freshly cropped / flipped copy of the resident images;
    the run is reproducible for a given seed, differs from the static-feed run, and the patches are valid crops (ops test)."""
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.models import construct_model
    from fullbatchtraining_amd.training import FullBatchTrainer

    def run(extra):
        cfg = compose(["hyp=fb1", "hyp.steps=3", "hyp.warmup=0", "data.batch_size=32", "hyp.sub_batch=32", "impl.engine.chunk_group=2"] + extra,
                      original_cwd=str(tmp_path), name="aug", seed=5)
        torch.manual_seed(3)
        model = construct_model(cfg.model, 3, 10)
        x, y = make_data(4 * 32, 32)
        setup = dict(device=torch.device("cuda:0"), dtype=torch.float, memory_format=torch.contiguous_format)
        tr = FullBatchTrainer(model, (x, y), None, setup, cfg)
        feeds = []
        for _ in range(3):
            tr.step()
            feeds.append(tr.patches.float().cpu().clone())
        return tr.stats["train_loss"], feeds

    loss_a, feeds_a = run(["impl.engine.device_augment=True"])
    loss_b, feeds_b = run(["impl.engine.device_augment=True"])
    loss_s, feeds_s = run([])
    assert loss_a == loss_b and all(torch.equal(a, b) for a, b in zip(feeds_a, feeds_b))          # reproducible
    assert not torch.equal(feeds_a[0], feeds_a[1]) and not torch.equal(feeds_a[1], feeds_a[2])      # new draw every step
    assert torch.equal(feeds_s[0], feeds_s[2]) and not torch.equal(feeds_s[0], feeds_a[0])          # static feed without the switch
    assert loss_a[0] != loss_s[0] and all(np.isfinite(loss_a))


def test_train_from_lmdb_record_database(golden, tmp_path):
    """N x CIFAR-style feed: a record database written by the REFERENCE's LMDB writer (2 rounds over 23 images, HWC records;
    tests/test_cpu_data.py) decoded by ``LMDBRecords.as_feed`` and trained on -- chunks of 23 images (stored padded to 24) -- against the
    float64 oracle on the same tensors."""
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.data import LMDBRecords
    from fullbatchtraining_amd.models import construct_model
    from fullbatchtraining_amd.training import train
    from oracle import fb_oracle as orc
    from tests.helpers import DictLMDB, hyp_from_cfg
    from tests.test_cpu_data import _store

    data, _ = golden
    rec = LMDBRecords(env=DictLMDB(_store(data, "hwc_r2")), access="cursor")
    x, y = rec.as_feed(*data["lmdb/mean_std"])
    assert x.shape == (46, 3, 32, 32)
    cfg = compose(["hyp=fbclip", "hyp.steps=2", "hyp.warmup=0", "data.batch_size=23", "hyp.sub_batch=23", "impl.validate_every_nth_step=1000",
                   "impl.engine.chunk_group=2"], original_cwd=str(tmp_path), name="lmdb")
    torch.manual_seed(2)
    model = construct_model(cfg.model, 3, 10)
    from tests.helpers import oracle_device, to_oracle
    state = {k: (v.clone().double() if v.is_floating_point() else v.clone()).to(oracle_device()) for k, v in model.state_dict().items()}
    want = orc.train(orc.Spec(18), state, *to_oracle(x, y), hyp_from_cfg(cfg), 2, 23, cfg.hyp.scheduler, cfg.hyp.warmup)
    setup = dict(device=torch.device("cuda:0"), dtype=torch.float, memory_format=torch.contiguous_format)
    stats = train(model, (x, y), None, setup, cfg)
    for key in ("train_loss", "grad_norm", "param_norm", "full_loss", "preclip_gradnorm", "train_acc"):
        assert np.allclose(stats[key], want[key], rtol=2e-3, atol=1.01 / 46 if key == "train_acc" else 1e-6), (key, stats[key], want[key])
    assert abs(stats["train_loss"][0] - want["train_loss"][0]) < 1e-5 * want["train_loss"][0]
