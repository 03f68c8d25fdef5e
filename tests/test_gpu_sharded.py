"""Sharded (multi-process) training path on ONE GPU: two ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one
device), which exercises everything of the data-parallel path except the transport: chunk-range sharding, local running
mean x K_r/K, reduce-scatter, sharded clip + Nesterov SGD with sharded momentum, parameter all-gather, BN running-stat
recombination, stats gathering.  The result must equal the 1-process run (1-process semantics are the target)."""
import os
import socket

import numpy as np
import pytest
import torch

from tests.helpers import pg_timeout, spawn_bounded

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPAWN_CAP_S = 200        # wall-clock cap of one spawned run (measured: 12-20 s); tests/helpers.spawn_bounded kills the ranks and fails the test

OVERRIDES = ["hyp=fbclip", "hyp.steps=3", "hyp.warmup=1", "data.batch_size=32", "hyp.sub_batch=32", "data.pixels=16",
             "impl.validate_every_nth_step=1000", "impl.engine.chunk_group=2"]
N, PIXELS, SEED = 7 * 32, 16, 11       # 7 chunks: ranks own 4 and 3


OPTIONS = ["hyp/optim_modification=SAM", "hyp.grad_clip_norm=inf", "hyp.grad_clip=0.02", "hyp.only_linear_layers_weight_decay=True",
           "hyp.norm_bias.strength=1e-3", "hyp.norm_bias.norm_type=2", "hyp.norm_bias.bias=50", "hyp.grad_noise.additive=1e-3"]


def _run(rank, world, port, out_dir, grad_reg, backend="gloo", tag=None):
    import sys
    sys.path.insert(0, REPO)
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.models import construct_model
    from fullbatchtraining_amd.training import train
    from tests.helpers import make_data

    dev = rank if backend == "nccl" and world > 1 else 0          # RCCL: one device per rank; gloo: the ranks share cuda:0
    torch.cuda.set_device(dev)
    # "onegroup*": all chunks of a rank in ONE chunk group (what a rank of an 8-GPU job runs: 49 chunks <= chunk_group 98) -- the only
    # schedule in which the late bucket of the exchange leaves from inside the backward pass (Engine.full_gradient(late_bucket=...))
    one = ["impl.engine.chunk_group=8"]
    # "r50*": a Bottleneck plan (ResNet-50, 'standard' stem: 7x7/s2 patches + MaxPool, 1x1 / 3x3 / 1x1 blocks, 25.6 M-parameter arena) at 64 px -- the block types of
    # BASELINE config 5; its late bucket starts at layers.3.0.conv1 and leaves from the on_block_done callback inside the last backward pass ("onegroup")
    r50 = ["model=resnet50", "model.stem=standard", "data.pixels=64"]
    extra = {False: [], True: ["hyp.grad_reg.block_strength=0.5"], "options": OPTIONS, "shuffle": [], "ckpt": [], "ckpt_resume": ["hyp.steps=5"],
             "r50": r50, "r50_gradreg": r50 + ["hyp.grad_reg.block_strength=0.5"], "r50_onegroup": r50 + one,
             "r50_onegroup_gradreg": r50 + one + ["hyp.grad_reg.block_strength=0.5"],
             "onegroup": one, "onegroup_gradreg": one + ["hyp.grad_reg.block_strength=0.5"],
             "onegroup_bf16_px32": one + ["impl.mixed_precision=True", "data.pixels=32"],
             "onegroup_central": one + ["hyp.grad_reg.block_strength=0.5", "hyp.grad_reg.implementation=central-differences"],
             "acc": ["hyp.grad_reg.block_strength=0.0", "hyp.grad_reg.acc_strength=0.5", "hyp.grad_reg.implementation=central-differences"]}
    n_run = N
    if str(grad_reg).startswith("r152_224"):
        # BASELINE config 5 literally: ResNet-152, 'standard' stem, 224 px, chunks of 128, with the regulariser -- 3 chunks (ranks own 2 and 1), 2 steps (lr 0, then the
        # first real update), everything of a rank in one group: the late bucket of the 60 M-parameter arena leaves from inside the finite-difference pass
        extra[grad_reg] = ["model=resnet152", "model.stem=standard", "data.pixels=224", "data.batch_size=128", "hyp.sub_batch=128", "hyp.steps=2",
                           "hyp.grad_reg.block_strength=0.5", "impl.engine.chunk_group=8"] + (["impl.mixed_precision=True", "hyp.grad_reg.block_strength=0.0"] if grad_reg.endswith("bf16") else [])
        n_run = 3 * 128
    over = list(OVERRIDES) + extra[grad_reg]
    if world > 1 or backend == "nccl":
        kw = dict(device_id=torch.device("cuda", dev)) if backend == "nccl" else {}
        torch.distributed.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=pg_timeout(), **kw)
        over.append("impl/setup=distributed")
    cfg = compose(over, original_cwd=out_dir, name="sharded")
    torch.manual_seed(SEED)
    model = construct_model(cfg.model, 3, 10)
    x, y = make_data(n_run, 32 if "px32" in str(grad_reg) else (64 if "r50" in str(grad_reg) else (224 if "r152_224" in str(grad_reg) else PIXELS)))
    setup = dict(device=torch.device("cuda", dev), dtype=torch.float, memory_format=torch.contiguous_format)
    feed = (x, y)
    if "ckpt" in str(grad_reg):                # checkpoint written by rank 0 of a sharded run / resumed by every rank
        cfg.impl.checkpoint.name = "sharded.pth"
        cfg.impl.checkpoint.save_every_nth_step = 1000
    if grad_reg == "shuffle":          # a shuffling train loader: every rank follows rank 0's permutation of each step
        from tests.helpers import shuffling_loaders
        feed = shuffling_loaders(x, y, 32)[0]
    stats = train(model, feed, None, setup, cfg)
    keep = {k: v for k, v in stats.items() if k != "train_time"}
    torch.save(dict(stats=keep, state={k: v.cpu() for k, v in model.state_dict().items()},
                    grads=[p.grad.cpu() for p in model.parameters()]),
               os.path.join(out_dir, f"{tag or 'w' + str(world)}_r{rank}.pt"))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def _single(out, mode, tag=None):
    """The 1-process run the sharded ones are compared with: in THIS process (no process group; saves an interpreter + HIP start-up per test)."""
    _run(0, 1, 0, out, mode, "gloo", tag)


def _ranks(world, out, mode, backend="gloo", tag=None):
    spawn_bounded(_run, (world, _free_port(), out, mode, backend, tag), world, timeout=SPAWN_CAP_S)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _compare(got, ref, grad_reg=False):
    for key in ("train_loss", "train_acc", "param_norm", "grad_norm", "full_loss", "preclip_gradnorm", "clipped_step"):
        atol = 1.01 / N if key == "train_acc" else 1e-6
        assert np.allclose(got["stats"][key], ref["stats"][key], rtol=2e-4 if grad_reg is False else 5e-3, atol=atol), (key, got["stats"][key], ref["stats"][key])
    for name, t in ref["state"].items():
        if t.is_floating_point():
            scale = max(float(t.abs().max()), 2e-2)
            assert float((got["state"][name] - t).abs().max()) < 1e-3 * scale + 1e-6, name
        else:
            assert torch.equal(got["state"][name], t), name


def test_rccl_collectives_one_rank(tmp_path, monkeypatch):
    """The RCCL branch of the exchange (`reduce_scatter_tensor`, `all_gather_into_tensor`, the norm all-reduce) through the `nccl`
    backend with a process group of ONE rank on cuda:0 (FB_FORCE_DIST=1 routes the step through the sharded path): on a 1-GPU box this
    is the only way to execute those calls; the arithmetic must equal the plain 1-process step bit for bit."""
    out = str(tmp_path)
    # "onegroup*": the late bucket's asynchronous reduce-scatter starts on the side stream, from inside the last backward pass; "+poison": the
    # local values of that bucket are overwritten with NaN behind the collective (FB_EXCHANGE_POISON) -- any consumer of the stale slice
    # (a launch ordered on the wrong stream, an update reading ``avg`` instead of the reduced shard) would poison the step
    for mode in (False, "onegroup", "onegroup_gradreg", "onegroup+poison", "onegroup_gradreg+poison"):
        poison = mode is not False and mode.endswith("+poison")
        mode = mode[:-len("+poison")] if poison else mode
        monkeypatch.delenv("FB_FORCE_DIST", raising=False)
        monkeypatch.delenv("FB_EXCHANGE_POISON", raising=False)
        _single(out, mode)
        monkeypatch.setenv("FB_FORCE_DIST", "1")
        if poison:
            monkeypatch.setenv("FB_EXCHANGE_POISON", "1")
        _ranks(1, out, mode, "nccl", "rccl1")
        ref, got = torch.load(os.path.join(out, "w1_r0.pt")), torch.load(os.path.join(out, "rccl1_r0.pt"))
        _compare(got, ref, mode)
        for key in ("train_loss", "param_norm"):
            assert got["stats"][key] == ref["stats"][key], (mode, key)
        # (with an early bucket |g_k|^2 is the sum of two partial sums: last-bit differences in the recorded chunk norms only)
        assert np.allclose(got["stats"]["grad_norm"], ref["stats"]["grad_norm"], rtol=0 if mode is False else 1e-6), mode
        # p.grad = the CLIPPED mean gradient: the clip norm is the sum of the exchange's two bucket norms here and one reduction over the
        # whole arena in the plain step -- the same number up to the association of the last addition, i.e. the scale may differ by one ulp
        if "gradreg" not in str(mode):
            for a, b in zip(got["grads"], ref["grads"]):
                assert torch.allclose(a, b, rtol=5e-7, atol=0), mode
        else:
            # ... and with the regulariser that ulp is in theta after the first real update (step 2), and the finite-difference quotient of step 3
            # amplifies it by 1 / eps_n ~ 1e4 (tools/scratch/rccl1_probe.py: 1.1e-2 on p.grad with bf16x6 whatever the overlap / replay / stream
            # switches; bit-identical when the two clip norms happen to round alike, as they do with f16x2).  NaN from a poisoned slice fails either way
            a, b = (torch.cat([t.reshape(-1) for t in side["grads"]]).double() for side in (got, ref))
            assert bool(torch.isfinite(a).all()) and float((a - b).norm() / b.norm()) < 3e-2, mode


def test_chained_weight_gradients_under_the_early_late_bucket(tmp_path, monkeypatch):
    """FB_WGRAD_CHAIN=1 (the 4x4 layers' weight gradients as group sums + per-chunk sums of squares) in the sharded step whose late bucket is
    folded on the side stream while the main stream folds the early range: the two folds must not share scratch (round 4's ``sq_seg[0]`` was
    written and read by both with no order between them -- the per-chunk norms of one range could come out as the other's).  One RCCL rank,
    bf16 at 32 px (the shape that has 4x4 maps), chained vs unchained: same per-chunk gradient norms, same mean gradient."""
    out = str(tmp_path)
    mode = "onegroup_bf16_px32"
    monkeypatch.setenv("FB_FORCE_DIST", "1")
    monkeypatch.delenv("FB_WGRAD_CHAIN", raising=False)
    _ranks(1, out, mode, "nccl", "plain")
    monkeypatch.setenv("FB_WGRAD_CHAIN", "1")
    for rep in range(2):                                 # (a race: twice)
        _ranks(1, out, mode, "nccl", f"chain{rep}")
    ref = torch.load(os.path.join(out, "plain_r0.pt"))
    for rep in range(2):
        got = torch.load(os.path.join(out, f"chain{rep}_r0.pt"))
        # the chained kernel sums a chunk group in another order (a chain of chunks per tile): last bits of the bf16 step, amplified by
        # two updates -- statistics to 1e-3, the per-chunk norms of every step included
        for key in ("train_loss", "grad_norm", "full_loss", "preclip_gradnorm", "param_norm"):
            assert np.allclose(got["stats"][key], ref["stats"][key], rtol=2e-3, atol=1e-6), (rep, key, got["stats"][key], ref["stats"][key])
        keys = [k for k in ref["stats"] if k.startswith("grad_norm_train_")] or []
        for key in keys:
            assert np.allclose(got["stats"][key], ref["stats"][key], rtol=2e-3), (rep, key)
        a, b = (torch.cat([t.reshape(-1) for t in side["grads"]]).double() for side in (got, ref))
        assert bool(torch.isfinite(a).all()) and float((a - b).norm() / b.norm()) < 5e-3, rep


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL wants one device per rank")
@pytest.mark.parametrize("grad_reg", [False, True])
def test_two_rank_rccl_run_equals_single_process(tmp_path, grad_reg):
    """Two ranks on two GPUs over RCCL (reduce-scatter + sharded update + all-gather over xGMI) == the 1-process run."""
    out = str(tmp_path)
    _single(out, grad_reg)
    _ranks(2, out, grad_reg, "nccl", "rccl2")
    ref = torch.load(os.path.join(out, "w1_r0.pt"))
    for r in range(2):
        _compare(torch.load(os.path.join(out, f"rccl2_r{r}.pt")), ref, grad_reg)


def test_sharded_checkpoint_holds_whole_momentum(tmp_path):
    """Rank 0 of a 2-rank run saves the checkpoint: its momentum buffers (each rank updates only its shard) and the exposed p.grad
    must be the whole vectors -- equal to the 1-process run's -- and a 2-rank run resumed from it continues like the 1-process run
    resumed from its own checkpoint (reference workflow: train_distributed_with_checkpoints.sh)."""
    out1, out2 = str(tmp_path / "one"), str(tmp_path / "two")
    os.makedirs(out1), os.makedirs(out2)
    _single(out1, "ckpt")
    _ranks(2, out2, "ckpt")
    c1 = torch.load(os.path.join(out1, "checkpoints", "sharded.pth"), weights_only=False)
    c2 = torch.load(os.path.join(out2, "checkpoints", "sharded.pth"), weights_only=False)
    assert c1[4] == c2[4] == 3
    worst = 0.0
    for idx, st in c1[0]["state"].items():
        a, b = st["momentum_buffer"], c2[0]["state"][idx]["momentum_buffer"]
        # zeros / stale values outside rank 0's shard would be off by the tensor's own scale; the two runs themselves differ by fp32
        # chunk-gradient noise (the ranks sum the full-batch gradient in another order), ~1e-3 of a typical momentum entry (1e-2)
        scale = max(float(a.abs().max()), 2e-2)
        worst = max(worst, float((a - b).abs().max()) / scale)
        assert float((a - b).abs().max()) < 1e-2 * scale, idx
        assert float(b.abs().max()) > 0.2 * float(a.abs().max()), idx         # not zeros
    print(f"momentum buffers of the 2-rank checkpoint vs the 1-process one: worst relative difference {worst:.2e}")
    ref = torch.load(os.path.join(out1, "w1_r0.pt"))
    for r in range(2):
        got = torch.load(os.path.join(out2, f"w2_r{r}.pt"))
        for a, b in zip(got["grads"], ref["grads"]):      # p.grad of the last step, whole on every rank (fp32 chunk-gradient noise apart)
            assert float((a - b).abs().max()) < 1e-2 * max(float(b.abs().max()), 2e-2)
            assert float(a.abs().max()) > 0.2 * float(b.abs().max())
    # resume both runs from their own checkpoints for two more steps
    _single(out1, "ckpt_resume", "resumed")
    _ranks(2, out2, "ckpt_resume", tag="resumed")
    ref = torch.load(os.path.join(out1, "resumed_r0.pt"))
    assert len(ref["stats"]["train_loss"]) == 2
    for r in range(2):
        got = torch.load(os.path.join(out2, f"resumed_r{r}.pt"))
        for key in ("train_loss", "grad_norm", "param_norm", "full_loss"):
            # steps 4-5 of the run: the ranks sum the full-batch gradient in another order than one process, and fp32 chunk-gradient noise
            # has grown to ~2e-3 by step 5 (a resumed shard without its momentum would be off by tens of per cent)
            assert np.allclose(got["stats"][key], ref["stats"][key], rtol=1e-2), (key, got["stats"][key], ref["stats"][key])


# (the 2-rank gloo matrix comes AFTER the RCCL and checkpoint cases: on a slow box those are not the first casualties)
@pytest.mark.parametrize("grad_reg", [False, True, "options", "acc", "shuffle", "onegroup", "onegroup_gradreg", "onegroup_central",
                                      "r50", "r50_gradreg", "r50_onegroup", "r50_onegroup_gradreg"])
def test_two_rank_run_equals_single_process(tmp_path, grad_reg):
    """plain step and regulariser: sharded update (reduce-scatter / all-gather); "options": SAM + L-infinity clip + norm bias +
    per-tensor weight decay + gradient noise (rank 0's draw, broadcast), which all-reduce the gradient and replicate the 1-process update; "acc": the acc_strength pre-pass
    (its own all-reduce and BN recombination inside the closure)."""
    out = str(tmp_path)
    _single(out, grad_reg)
    _ranks(2, out, grad_reg)
    ref = torch.load(os.path.join(out, "w1_r0.pt"))
    for r in range(2):
        got = torch.load(os.path.join(out, f"w2_r{r}.pt"))
        for key in ("train_loss", "train_acc", "param_norm", "grad_norm", "full_loss", "preclip_gradnorm", "clipped_step"):
            atol = 1.01 / N if key == "train_acc" else 1e-6          # one prediction may flip once the parameters differ in the last bits
            # ("acc": the central-difference term divides the difference of two fp32 gradients by 2 eps_n -- a last-bit difference in the
            # all-reduced pre-pass mean is 7e-3 on the clip norm of step 3; steps 1-2 are asserted bit-equal below)
            # ("r50*": 53 convolution layers -- the fp32 conditioning of a chunk gradient is ~5x ResNet-18's, test_bottleneck_standard_stem_chunk_gradients_vs_oracle)
            rtol = {False: 2e-4, "onegroup": 2e-4, "acc": 2e-2, "r50": 1e-3, "r50_onegroup": 1e-3, "r50_gradreg": 2.5e-2, "r50_onegroup_gradreg": 2.5e-2}.get(grad_reg, 5e-3)
            assert np.allclose(got["stats"][key], ref["stats"][key], rtol=rtol, atol=atol), (key, got["stats"][key], ref["stats"][key])
        if grad_reg == "acc":      # warm-up step (lr = 0) and the step after it see identical parameters: the exchange itself is exact
            for key in ("train_loss", "grad_norm", "full_loss", "param_norm"):
                assert got["stats"][key][:2] == ref["stats"][key][:2], key
        if str(grad_reg).startswith("r50"):
            # the same on the Bottleneck plan: at identical parameters (steps 1-2) every chunk's loss and gradient norm are the 1-process run's BITS whatever rank
            # and chunk group the chunk ran in -- the K-slice counts of the weight gradients follow the nominal group, the fp16x2 / bf16x6 arithmetic is per chunk
            assert got["stats"]["train_loss"][:2] == ref["stats"]["train_loss"][:2] and got["stats"]["param_norm"][:2] == ref["stats"]["param_norm"][:2]
            for k in range(7):
                a, b = got["stats"][f"grad_norm_train_{k}"][:2], ref["stats"][f"grad_norm_train_{k}"][:2]
                assert np.allclose(a, b, rtol=1e-6 if "onegroup" in grad_reg else 0), (k, a, b)       # (an early late bucket: |g_k|^2 is the sum of two partial sums)
        for k in range(7):
            # steps 1-2 agree to the bit; from step 3 on the parameters differ in the last bits (the ranks sum the full-batch
            # gradient in a different order) and a chunk gradient amplifies that to ~1e-4 (fp32 noise floor, cf. test_gpu_training)
            assert np.allclose(got["stats"][f"grad_norm_train_{k}"], ref["stats"][f"grad_norm_train_{k}"],
                               rtol={False: 1e-3, "onegroup": 1e-3, "r50_gradreg": 5e-2, "r50_onegroup_gradreg": 5e-2}.get(grad_reg, 5e-3)), (
                k, got["stats"][f"grad_norm_train_{k}"], ref["stats"][f"grad_norm_train_{k}"])
        for name, t in ref["state"].items():
            if t.is_floating_point():
                # running means of zero-mean conv outputs are ~1e-3 with an fp32 noise floor of ~1e-4 (cf. test_gpu_training)
                # BN shifts / running means are ~1e-3 after three steps and sit on the fp32 noise floor of the cancelling
                # chunk-gradient sums (cf. test_gpu_training): judge them on the scale of a typical parameter (2e-2)
                scale = max(float(t.abs().max()), 2e-2)
                # fp32 chunk-gradient noise (order of sums differs); the central-difference acc term divides the difference of two such
                # gradients by 2 eps_n: a last-bit difference in the all-reduced pre-pass mean moves a weight by up to 4-7 % after the
                # first real update, depending on where the last bits fall (6.4 % after round 4 respelled the BatchNorm dx expression with explicit
                # fmafs; steps 1-2, taken at identical parameters, agree to the bit -- asserted below)
                tol = {False: 1e-3, True: 1e-2, "options": 1e-2, "acc": 1.2e-1, "shuffle": 1e-2, "onegroup": 1e-3, "onegroup_gradreg": 1e-2,
                       "onegroup_central": 1e-2, "r50": 5e-3, "r50_onegroup": 5e-3, "r50_gradreg": 5e-2, "r50_onegroup_gradreg": 5e-2}[grad_reg]
                assert float((got["state"][name] - t).abs().max()) < tol * scale + 1e-6, name
            else:
                assert torch.equal(got["state"][name], t), name


@pytest.mark.parametrize("mode", ["r152_224_gradreg", "r152_224_bf16"])
def test_config5_two_rank_run_equals_single_process(tmp_path, mode):
    """BASELINE config 5's sharded form at its real shape -- ResNet-152, 'standard' stem, 224 x 224, chunks of 128; with the GradRegularizer (fp32 storage, bf16x6) and in
    its bf16 form without -- on two ranks (gloo, one GPU, the real kernels): 3 chunks (2 + 1), the rank's chunks in one group so that the 60 M-parameter arena's late
    bucket (from ``layers.3.0.conv1``) leaves through the ``on_block_done`` callback inside the last backward pass (the finite-difference pass with per-chunk weight sets),
    shard-local clip + SGD, all-gather.  Step 1 (lr 0) and step 2 see identical parameters: every chunk's loss and gradient norm equal the 1-process run's bits; the
    state after the first real update agrees to fp32 summation order."""
    out = str(tmp_path)
    _single(out, mode)
    _ranks(2, out, mode)
    ref = torch.load(os.path.join(out, "w1_r0.pt"))
    assert len(ref["stats"]["train_loss"]) == 2 and all(np.isfinite(ref["stats"]["train_loss"]))
    for r in range(2):
        got = torch.load(os.path.join(out, f"w2_r{r}.pt"))
        assert got["stats"]["train_loss"] == ref["stats"]["train_loss"] and got["stats"]["param_norm"] == ref["stats"]["param_norm"]
        for k in range(3):
            # (|g_k|^2: two partial sums with an early late bucket; bf16: the BatchNorm-backward partial rows cover 128-1024 pixels depending on the LAUNCH's pixel count,
            # fb_bn_bwd_reduce_rows, and another fp32 sum order moves a few bf16 roundings of dx: test_chunk_group_beyond_2g_byte_tensors_equals_smaller_groups)
            assert np.allclose(got["stats"][f"grad_norm_train_{k}"], ref["stats"][f"grad_norm_train_{k}"], rtol=3e-4 if mode.endswith("bf16") else 1e-6), k       # (measured: 0 / 2.9e-5)
        # (the norm of the exchanged mean gradient: the ranks' partial means are summed in another order than the 1-process running mean -- 1.4e-7 at step 1; at step 2 the
        # finite-difference term is thousands of times the gradient at random initialisation (profiles/r3_fd_conditioning.md) and carries that order at 8e-5)
        for key in ("grad_norm", "full_loss", "preclip_gradnorm", "clipped_step", "train_acc"):
            assert np.allclose(got["stats"][key], ref["stats"][key], rtol=2e-3 if mode.endswith("bf16") else 1e-3, atol=1e-6), (key, got["stats"][key], ref["stats"][key])
        norms = max(abs(a / b - 1.0) for k in range(3) for a, b in zip(got["stats"][f"grad_norm_train_{k}"], ref["stats"][f"grad_norm_train_{k}"]))
        print(f"{mode}, rank {r}: chunk gradient norms: worst relative difference {norms:.2e}; preclip norm {got['stats']['preclip_gradnorm']} vs {ref['stats']['preclip_gradnorm']}")
        worst = 0.0
        for name, t in ref["state"].items():
            if t.is_floating_point():
                scale = max(float(t.abs().max()), 2e-2)
                worst = max(worst, float((got["state"][name] - t).abs().max()) / scale)
            else:
                assert torch.equal(got["state"][name], t), name
        print(f"{mode}, rank {r}: parameters / buffers after the first real update: worst difference {worst:.2e} of the tensor's scale")
        assert worst < (5e-3 if mode.endswith("bf16") else 2e-2), worst


def test_bench_two_ranks_spawn_and_tear_down_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` as the driver calls it: the parent (which has not touched the GPU) starts two ranks through
    torch.distributed.run, every rank takes its chunk range, runs warm-up + timed + event-instrumented steps, rank 0 prints ONE JSON line, the
    process group is destroyed and the launcher exits 0.  On a 1-GPU box the two ranks share cuda:0 over gloo (FB_BENCH_SHARE_DEVICE=1: RCCL
    refuses two ranks on one device), so the number is not a scaling result -- the orchestration is what runs."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, FB_BENCH_SHARE_DEVICE="1")
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--images", "2048", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=200)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0 and np.isfinite(out["train_loss_last"])
    assert out["config"]["parallelism"].startswith("dp2") and "roofline" in out
