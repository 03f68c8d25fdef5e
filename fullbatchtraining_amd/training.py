"""Drop-in counterpart of the reference's full-batch branch of ``fullbatch.training.train``.

Mirrors, for ``hyp.train_stochastic=False``:
  train                      reference fullbatch/training/training.py:50-75, 217-239, 288-340
  gradient_evaluation        :226-234   (closure: accumulate, stats, clip)          -> Engine.full_gradient / sgd_step
  _accumulate_full_gradient  :121-185
  _record_stats              :85-119    (same keys, same formulas)
  _modify_gradient_params    :187-215   (norm bias, global L2 / L-infinity clip, gradient noise)
  evaluate                   :343-388
  get_loss_fn                :391-413   (cross entropy, label smoothing, incorrect-xent: fused into the head kernel)
  optim_interface            fullbatch/training/optimizers.py:10-93 (Gradient Descent / line_search none; cosine-*, warm-up)
  _save_to_checkpoint / _load_from_checkpoint   fullbatch/training/utils.py:43-70 (same 5-list layout)

All arithmetic of a step runs in ``libfbengine.so`` on the GPU; torch optimizers/schedulers are instantiated only as
*state containers* so that ``optimizer.state_dict()`` / ``scheduler.state_dict()`` in checkpoints have the reference's
exact layout and the LR sequence is produced by the same torch code the reference calls.
"""
import logging
import os
import re
import time
from collections import defaultdict

import numpy as np
import torch
from torch.optim.lr_scheduler import _LRScheduler

from . import lib
from .engine import Engine, stem_patches

log = logging.getLogger("fullbatchtraining_amd")


# ----------------------------------------------------------------------------------------------------------------------
# LR schedule objects (state containers; reference optimizers.py:69-93, additional_optimizers/scheduler.py:32-111)
# ----------------------------------------------------------------------------------------------------------------------
class GradualWarmupScheduler(_LRScheduler):
    """Linear warm-up from 0 (multiplier 1.0) to the base lr over ``total_epoch`` steps, then hands over to
    ``after_scheduler``.  Attribute names and ``state_dict`` layout follow the reference so checkpoints interchange."""

    def __init__(self, optimizer, multiplier, total_epoch, after_scheduler=None):
        if multiplier < 1.0:
            raise ValueError("multiplier should be greater thant or equal to 1.")
        self.multiplier, self.total_epoch, self.after_scheduler, self.finished = multiplier, total_epoch, after_scheduler, False
        super().__init__(optimizer)

    def get_lr(self):
        if self.last_epoch > self.total_epoch:
            if self.after_scheduler:
                if not self.finished:
                    self.after_scheduler.base_lrs = [b * self.multiplier for b in self.base_lrs]
                    self.finished = True
                return self.after_scheduler.get_last_lr()
            return [b * self.multiplier for b in self.base_lrs]
        if self.multiplier == 1.0:
            return [b * (float(self.last_epoch) / self.total_epoch) for b in self.base_lrs]
        return [b * ((self.multiplier - 1.0) * self.last_epoch / self.total_epoch + 1.0) for b in self.base_lrs]

    def step(self, epoch=None):
        if self.finished and self.after_scheduler:
            self.after_scheduler.step(None if epoch is None else epoch - self.total_epoch)
            self._last_lr = self.after_scheduler.get_last_lr()
        else:
            return super().step(epoch)

    def state_dict(self):
        state = {k: v for k, v in self.__dict__.items() if k != "optimizer"}
        state["after_scheduler"] = {k: v for k, v in self.after_scheduler.__dict__.items() if k != "optimizer"}
        return state

    def load_state_dict(self, state_dict):
        after = state_dict.pop("after_scheduler")
        self.after_scheduler.__dict__.update(after)
        self.__dict__.update(state_dict)


def optim_interface(model, cfg_hyp):
    """Only the branch on the hot path: ``Gradient Descent`` + ``line_search: none`` -> torch.optim.SGD (state container)."""
    if cfg_hyp.optim.name != "Gradient Descent" or cfg_hyp.optim.get("line_search", "none") != "none":
        raise NotImplementedError(f"optimizer {cfg_hyp.optim.name!r}/{cfg_hyp.optim.get('line_search')!r}: only plain gradient "
                                  "descent with Nesterov momentum is fused into the engine")
    mod = cfg_hyp.optim_modification.name
    if mod not in ("none", "SAM", "LARS", "LARC"):
        raise ValueError(f"Invalid optim_modification {mod} provided.")
    params = {k: v for k, v in cfg_hyp.optim.items() if k not in ("name", "line_search")}
    if cfg_hyp.only_linear_layers_weight_decay:      # reference optimizers.py:14-21: one param group per tensor, no decay on biases / gains
        parameter_iterable = []
        for key, value in model.named_parameters():
            if len(re.findall("(bias|gain)|skip_gain", key)) > 0:
                parameter_iterable += [{"params": [value], "weight_decay": 0.0}]
            else:
                parameter_iterable += [{"params": [value]}]
    else:
        parameter_iterable = model.parameters()
    optimizer = wrapped = torch.optim.SGD(parameter_iterable, **params)
    if mod == "SAM":
        wrapped = SAM(optimizer, rho=cfg_hyp.optim_modification.rho)
    elif mod in ("LARS", "LARC"):
        wrapped = LARS(optimizer, trust_coefficient=cfg_hyp.optim_modification.trust_coefficient, clip=mod == "LARC",
                       eps=cfg_hyp.optim_modification.eps)
    sched = cfg_hyp.scheduler
    if sched == "cosine-decay-floored":
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, cfg_hyp.steps, eta_min=cfg_hyp.optim.lr / 25)
    elif sched == "cosine-decay":
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, cfg_hyp.steps, eta_min=0.0)
    elif sched == "cosine-4000":
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, 4000, eta_min=0.0)
    elif sched == "linear":
        scheduler = torch.optim.lr_scheduler.MultiStepLR(
            optimizer, milestones=[cfg_hyp.steps // 2.667, cfg_hyp.steps // 1.6, cfg_hyp.steps // 1.142], gamma=0.1)
    elif sched == "exponential":
        scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=0.99)
    elif sched in ["", " ", None]:
        scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[], gamma=1)
    else:
        raise ValueError(f"Invalid scheduler {sched} provided.")
    if cfg_hyp.warmup > 0:
        scheduler = GradualWarmupScheduler(optimizer, multiplier=1.0, total_epoch=cfg_hyp.warmup, after_scheduler=scheduler)
    # The torch optimizer is a STATE CONTAINER here (param_groups / momentum buffers in the checkpoint layout): the update itself is the engine's
    # fb_mt_clip_sgd on the arena, so ``optimizer.step()`` is never called -- tell the schedulers that updates do happen, or the first
    # ``scheduler.step()`` warns "lr_scheduler.step() before optimizer.step()" on every run
    optimizer._opt_called = True
    return wrapped, scheduler            # the scheduler drives the wrapped SGD (reference optimizers.py:67 `optimizer.optim`)


class _OptimizerWrapper:
    """State container with the reference wrappers' surface (``.optim``, shared ``param_groups``, attribute pass-through, pickling
    via the wrapped optimizer; reference additional_optimizers/sam.py:34-54, lars.py:41-59).  The arithmetic runs in the engine."""

    def __init__(self, optimizer):
        self.optim = optimizer

    @property
    def param_groups(self):
        """Always the wrapped optimizer's CURRENT list: ``Optimizer.load_state_dict`` replaces it, and the scheduler (which drives the
        wrapped SGD) writes the lr there -- a reference held from construction would freeze the lr of a resumed SAM / LARS run."""
        return self.optim.param_groups

    @param_groups.setter
    def param_groups(self, value):
        self.optim.param_groups = value

    def __getstate__(self):
        return self.optim.__getstate__()

    def __setstate__(self, state):
        self.optim.__setstate__(state)

    def __repr__(self):
        return self.optim.__repr__()

    def __getattr__(self, name):
        if name == "optim":
            raise AttributeError(name)
        return getattr(self.optim, name)


class SAM(_OptimizerWrapper):
    """Sharpness-aware minimisation (reference additional_optimizers/sam.py): the step evaluates the full-batch closure twice --
    ``FullBatchTrainer.step`` runs closure -> ``Engine.sam_ascent`` -> closure -> ``Engine.sam_restore`` -> SGD update."""

    def __init__(self, optimizer, rho=0.05):
        assert rho >= 0.0, f"Invalid rho, should be non-negative: {rho}"
        super().__init__(optimizer)
        self.rho = rho


class LARS(_OptimizerWrapper):
    """LARS / LARC wrapper (reference additional_optimizers/lars.py).  Around a closure the reference's wrapper has exactly one
    effect: it rescales ``p.grad`` of the PREVIOUS step, zeroes the group weight decay and then calls ``SGD.step(closure)``, whose
    closure assigns fresh gradients (training.py:183-184) -- the update is plain SGD without weight decay, independent of
    ``trust_coefficient`` / ``clip`` / ``eps`` (pinned against the reference in tests/golden/scenarios_n4.npz)."""

    def __init__(self, optimizer, trust_coefficient=0.02, clip=False, eps=1e-8):
        super().__init__(optimizer)
        self.trust_coefficient, self.clip, self.eps = trust_coefficient, clip, eps


# ----------------------------------------------------------------------------------------------------------------------
# checkpoints (reference training/utils.py:43-70)
# ----------------------------------------------------------------------------------------------------------------------
def _sync_optimizer_state(engine, model, optimizer):
    """Expose the arena's momentum as torch SGD state so ``optimizer.state_dict()`` has the reference layout."""
    if engine.first_step:
        return
    for p, buf in zip(model.parameters(), engine.momentum_state()):
        optimizer.state[p]["momentum_buffer"] = buf.to(p.device, p.dtype)
    if getattr(engine, "e_w", None) is not None:       # SAM keeps its last ascent step in the wrapped optimizer's state (sam.py:66)
        for p, e in zip(model.parameters(), engine.sam_state()):
            optimizer.state[p]["e_w"] = e.to(p.device, p.dtype)


def _save_to_checkpoint(model, optimizer, scheduler, scaler, counter, file="checkpoints/fb.pth", engine=None):
    if engine is not None:
        engine.store_to_model(model)
        _sync_optimizer_state(engine, model, optimizer)
    optim_state, model_state, scheduler_state = optimizer.state_dict(), model.state_dict(), scheduler.state_dict()
    scaler_state = scaler.state_dict() if scaler is not None else None
    torch.save([optim_state, model_state, scheduler_state, scaler_state, counter.step], file)


def _load_from_checkpoint(model, optimizer, scheduler, scaler, counter, max_steps, device=None, file="checkpoints/fb.pth", engine=None):
    try:
        optim_state, model_state, scheduler_state, scaler_state, step = torch.load(file, map_location=device, weights_only=False)
    except FileNotFoundError:
        print("No existing checkpoint found. Starting to train from step 0.")
        return
    model.load_state_dict(model_state)
    optimizer.load_state_dict(optim_state)
    scheduler.load_state_dict(scheduler_state)
    counter.step = step
    if engine is not None:
        engine.load_from_model(model)
        bufs = [optimizer.state[p].get("momentum_buffer") for p in model.parameters() if p in optimizer.state]
        if len(bufs) == len(list(model.parameters())) and all(b is not None for b in bufs):
            engine.load_momentum(bufs)
    if step >= max_steps:
        raise ValueError("Maximum step size reached. Terminating computations.")
    print(f"Existing checkpoint loaded successfully. Continuing to train from step {step}.")


# ----------------------------------------------------------------------------------------------------------------------
class _HeadLoss(torch.nn.Module):
    """The loss functions of the reference's ``get_loss_fn`` that the head kernel implements: mean cross entropy with label smoothing
    ``s`` (LabelSmoothCrossEntropyLoss, reference modules.py:86-101) and optionally only on misclassified samples
    (IncorrectCrossEntropyLoss, modules.py:104-119).  Callable like the reference's modules (host-side, for callers that hold a
    loss_fn); the engine reads ``smoothing`` / ``only_incorrect`` and runs the fused kernel."""

    def __init__(self, smoothing=0.0, only_incorrect=False):
        super().__init__()
        self.smoothing, self.only_incorrect = float(smoothing), bool(only_incorrect)

    def forward(self, input, target):
        log_prob = torch.nn.functional.log_softmax(input, dim=-1)
        weight = torch.ones_like(input) * self.smoothing / (input.shape[-1] - 1.0)
        weight.scatter_(-1, target.unsqueeze(-1), (1.0 - self.smoothing))
        loss_per_sample = (-weight * log_prob).sum(dim=-1)
        if self.only_incorrect:
            loss_per_sample = loss_per_sample * (1 - (input.argmax(dim=1) == target).float())
        return loss_per_sample.mean()


def get_loss_fn(cfg_hyp, batch_size):
    """Reference training.py:391-413.  The maxup losses need the augmented-trials layout of the stochastic pipeline (off this path)."""
    smoothing = cfg_hyp.label_smoothing if cfg_hyp.label_smoothing not in [None, ""] else 0.0
    if cfg_hyp.loss_modification is None:
        return torch.nn.CrossEntropyLoss() if smoothing == 0 else _HeadLoss(smoothing)
    if cfg_hyp.loss_modification == "incorrect-xent":
        return _HeadLoss(smoothing, only_incorrect=True)
    if "maxup" in str(cfg_hyp.loss_modification):
        raise NotImplementedError("maxup losses are off the full-batch hot path")
    raise ValueError(f"Invalid loss modification {cfg_hyp.loss_modification}.")


def _is_shuffling(loader):
    """A DataLoader whose sampler is not sequential (the reference's loaders for hyp.shuffle=True): every pass has a new order."""
    return isinstance(loader, torch.utils.data.DataLoader) and not isinstance(loader.sampler, torch.utils.data.SequentialSampler)


def _stage_dataset(loader, device):
    """The DATASET behind a loader, in dataset order, on the device -- read through a private sequential loader so that the
    generator of the caller's loader is not advanced."""
    probe = torch.utils.data.DataLoader(loader.dataset, batch_size=1024, shuffle=False, drop_last=False, generator=torch.Generator())
    xs, ys = zip(*[(x, y) for x, y in probe])
    return torch.cat(xs).to(device), torch.cat(ys).to(device=device, dtype=torch.long)


def _pass_indices(loader):
    """Sample indices of one pass over ``loader`` in its order; advances its generator (or the default one) exactly like the
    reference's ``for ... in trainloader`` does: the DataLoader iterator draws a base seed, then the sampler its permutation."""
    torch.empty((), dtype=torch.int64).random_(generator=loader.generator)
    return torch.tensor([i for batch in loader.batch_sampler for i in batch], dtype=torch.long)


def _stage(loader, device):
    """Materialise a (static, unaugmented) loader as tensors in loader order (drop_last honoured), on ``device`` -- or, with
    ``device=None``, wherever they are (the trainer then moves only its own rank's slice into HBM)."""
    if isinstance(loader, (tuple, list)) and torch.is_tensor(loader[0]):
        X, Y = loader[0], loader[1]
    else:
        xs, ys = [], []
        for x, y in loader:
            xs.append(x)
            ys.append(y)
        X, Y = torch.cat(xs), torch.cat(ys)
    if device is None:
        return X, Y.to(dtype=torch.long)
    return X.to(device), Y.to(device=device, dtype=torch.long)


def _check_scope(cfg):
    hyp = cfg.hyp
    if hyp.train_stochastic or hyp.train_switch_stochastic is not None:
        raise NotImplementedError("the engine implements the full-batch branch (hyp.train_stochastic=False) only")
    if hyp.grad_reg.acc_strength != 0 and max(cfg.data.batch_size // hyp.sub_batch, 1) != 1 \
            and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        raise NotImplementedError("grad_reg.acc_strength with sub_batch < batch_size on several ranks (the pre-pass runs whole blocks; "
                                  "chunk ranges of the ranks would have to be cut at block boundaries)")
    if hyp.batch_clip is not None and float(hyp.grad_clip_norm) != 2.0:
        raise NotImplementedError("hyp.batch_clip with grad_clip_norm != 2 (the per-chunk clip kernel implements the L2 norm)")
    if hyp.grad_clip is not None and not float(hyp.grad_clip_norm) >= 1.0:
        raise NotImplementedError("grad_clip_norm must be a p-norm with p >= 1 (or inf)")


def _device_augmentation(cfg, trainloader):
    """On-device RandomCrop / RandomHorizontalFlip (reference config/data/CIFAR10.yaml:11-13).  Opt-in with
    `impl.engine.device_augment=True`: the caller vouches that the feed (tensor pair or loader) yields UN-augmented, normalised
    images; they stay resident and every step sees a freshly cropped / flipped copy.  Without it a feed is staged once as it comes
    (static dataset), which is also what the reference runs with `data.augmentations_train=` do.
    Returns None or dict(crop_pad, flip_p, pad_value)."""
    aug = cfg.data.get("augmentations_train")
    if not aug:
        return None
    if not bool(cfg.impl.get("engine", {}).get("device_augment", False)):
        log.warning("data.augmentations_train is set: the device-resident feed stages the loader ONCE (static dataset); "
                    "impl.engine.device_augment=True augments un-augmented inputs on the device every step.")
        return None
    out = dict(crop_pad=0, flip_p=0.0, pad_value=None)
    for key in aug.keys():
        if key == "RandomCrop":
            size, pad = (list(aug[key]) + [0])[:2] if not isinstance(aug[key], int) else (aug[key], 0)
            if int(size) != int(cfg.data.pixels):
                raise NotImplementedError(f"RandomCrop to {size} != data.pixels {cfg.data.pixels}")
            out["crop_pad"] = int(pad)
        elif key == "RandomHorizontalFlip":
            out["flip_p"] = float(aug[key])
        else:
            raise NotImplementedError(f"data.augmentations_train.{key}: only RandomCrop and RandomHorizontalFlip run on the device")
    if cfg.data.get("normalize", False):       # RandomCrop pads the raw image with black; the tensors here are normalised
        out["pad_value"] = [-float(m) / float(sd) for m, sd in zip(cfg.data.mean, cfg.data.std)]
    return out


def _evaluate_batch(eng, xb, yb, test_time_flips):
    """Loss sum and #correct of one validation batch (reference training.py:365-380).  ``test_time_flips``: the reference feeds the SUM
    of the softmax outputs of the image and of its horizontal mirror to the loss function and the argmax (training.py:370-373); the
    two forward passes run through the engine (the mirror is taken by the patch-gather kernel), the epilogue is ``fb_head_tta``."""
    if not test_time_flips:
        l, c = eng.evaluate_batch(stem_patches(xb, eng.plan.stem, eng.dt), yb)
        return l * yb.shape[0], c
    n = xb.shape[0]
    eng.evaluate_batch(stem_patches(xb, eng.plan.stem, eng.dt), yb)
    left = eng.logits[:n].clone()
    mirror = torch.ones(n, dtype=torch.int8, device=xb.device)
    eng.evaluate_batch(stem_patches(xb.float().contiguous(), eng.plan.stem, eng.dt, aug=(None, None, mirror, 0, None)), yb)
    ws = torch.empty(2 * n + 2, device=xb.device, dtype=torch.float32)
    lib.call("fb_head_tta", left.data_ptr(), eng.logits.data_ptr(), yb.data_ptr(), n, eng.plan.classes, ws.data_ptr(), ws.data_ptr() + 8 * n,
             ws.data_ptr() + 8 * n + 4)
    loss_sum, correct = ws[2 * n:].tolist()
    return loss_sum, correct


class _Stats(defaultdict):
    """The ``stats`` dict of a run (reference: ``defaultdict(list)``).  The per-step statistics are produced on the GPU and read back
    asynchronously; any READ of the dict first completes the read-backs that are still pending, so a reader always sees every step that
    ``FullBatchTrainer.step`` has returned from -- the reference's semantics (its `_record_stats` syncs) -- while a loop that does not
    look at the statistics between steps (bench.py) keeps the host one step ahead of the GPU instead of idling it for the 2-3 ms of
    host-side bookkeeping per step (7-9 % of a step at 8 GPUs)."""

    def __init__(self, flush):
        super().__init__(list)
        self._flush = flush

    def raw(self, key):
        return defaultdict.__getitem__(self, key)

    def __getitem__(self, key):
        self._flush()
        return defaultdict.__getitem__(self, key)

    def __contains__(self, key):
        self._flush()
        return defaultdict.__contains__(self, key)

    def __iter__(self):
        self._flush()
        return defaultdict.__iter__(self)

    def __len__(self):
        self._flush()
        return defaultdict.__len__(self)

    def keys(self):
        self._flush()
        return defaultdict.keys(self)

    def items(self):
        self._flush()
        return defaultdict.items(self)

    def values(self):
        self._flush()
        return defaultdict.values(self)

    def get(self, key, default=None):
        self._flush()
        return defaultdict.get(self, key, default)

    def __reduce__(self):                      # pickles / copies as a plain dict of lists
        self._flush()
        return (dict, (dict(defaultdict.items(self)),))


class FullBatchTrainer:
    """Owns the engine, the resident dataset and the optimizer/scheduler state containers for one training run."""

    def __init__(self, model, trainloader, validloader, setup, cfg):
        _check_scope(cfg)
        self.cfg, self.model = cfg, model
        self.device = torch.device(setup["device"]) if not isinstance(setup["device"], torch.device) else setup["device"]
        self.optimizer, self.scheduler = optim_interface(model, cfg.hyp)
        # (the engine applies the update on its arena, fb_mt_clip_sgd; the torch optimizer is the state container the checkpoint layout wants)
        self.loss_fn = get_loss_fn(cfg.hyp, cfg.data.batch_size)
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        self.rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
        # the multi-process code path (collectives, sharded update); FB_FORCE_DIST=1 takes it with an initialised process group of ONE
        # rank as well, which runs the real RCCL reduce-scatter / all-gather calls on a 1-GPU box (tests/test_gpu_sharded.py)
        self.multi = self.world > 1 or (torch.distributed.is_initialized() and os.environ.get("FB_FORCE_DIST") == "1")
        # a shuffling train loader: the dataset stays resident in dataset order and every step regathers it in the order of a new
        # pass over the loader's batch sampler (reference: `for block, (inputs, labels) in enumerate(trainloader)` every step)
        self.shuffler = trainloader if _is_shuffling(trainloader) else None
        if self.shuffler is not None:
            Xall, Yall = _stage_dataset(trainloader, self.device)
            per_pass = len(trainloader.batch_sampler) * (trainloader.batch_size or 1) if trainloader.drop_last else len(trainloader.sampler)
            X, Y = Xall[:per_pass], Yall[:per_pass]                   # shapes only; the first step gathers the first permutation
        else:
            X, Y = _stage(trainloader, None)                         # only this rank's chunk range is copied to the device below
        block = min(cfg.data.batch_size, X.shape[0])
        chunks_in_block = max(block // cfg.hyp.sub_batch, 1)
        if block % chunks_in_block != 0:
            raise NotImplementedError("data.batch_size must be divisible into equal sub_batch chunks")
        self.chunk, self.block = block // chunks_in_block, block
        self.num_blocks = X.shape[0] // block               # drop_last=True
        self.n_chunks = self.num_blocks * chunks_in_block
        self.datapoints = self.n_chunks * self.chunk
        s = cfg.hyp.grad_reg.block_strength
        impl = cfg.hyp.grad_reg.implementation
        fd = s != 0 or cfg.hyp.grad_reg.acc_strength != 0
        fd_sets = 0 if not fd else (2 if impl == "central-differences" else 1)
        self.dtype = torch.bfloat16 if (cfg.impl.mixed_precision and not fd) else torch.float32
        if cfg.impl.mixed_precision and fd:
            log.warning("grad_reg finite differences need matching fp32 passes (perturbation ~1e-6 per weight): running fp32.")
        from .parallel import ShardPlan, group_size
        self.shard = ShardPlan(self.n_chunks, self.world, self.rank)
        from .engine import Plan, max_group, padded_chunk
        plan = Plan(model, X.shape[-1])
        # chunk sizes that do not fill whole 128-pixel statistics blocks (data.batch_size=125: all 50 000 images in 400 chunks,
        # reference data_preparation.py:64-72) are stored padded with zero images (label -1)
        self.chunk_pad = padded_chunk(plan, self.chunk)
        # what this trainer keeps on the device beside the engine: the stem's patches of the rank's whole shard (and the base images where they are re-augmented)
        es = torch.empty((), dtype=self.dtype).element_size()
        own = self.shard.count * self.chunk_pad * plan.stem.hout * plan.stem.wout * plan.stem.cin_pad * es + self.shard.count * self.chunk * X[0].numel() * 4
        want, cap = int(cfg.impl.get("engine", {}).get("chunk_group", 98)), max_group(plan, self.chunk_pad, self.dtype, self.device, reserve_bytes=own, fd_sets=fd_sets)
        # K-slice counts of the weight gradients are sized for the group of the WHOLE problem on one GPU -- the same number on every rank and in every run, so
        # that a chunk's summation order (hence its gradient, bit for bit) does not depend on the number of GPUs, on what else runs on the device or on the
        # allocator's state: the nominal cap comes from the device's TOTAL memory less what the 1-process run would keep resident (not from free memory, not from
        # this rank's share), and a job of several ranks takes the smallest of its ranks' values
        own_whole = self.n_chunks * self.chunk_pad * plan.stem.hout * plan.stem.wout * plan.stem.cin_pad * es + self.n_chunks * self.chunk * X[0].numel() * 4
        cap_nominal = max_group(plan, self.chunk_pad, self.dtype, self.device, reserve_bytes=own_whole, use_free=False, fd_sets=fd_sets)
        if self.world > 1:
            t = torch.tensor([cap_nominal], dtype=torch.int64, device=self.device if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
            cap_nominal = int(t.item())
        nominal = group_size(self.n_chunks, want, cap=cap_nominal)
        # this rank's own group: what is free NOW is a safety clamp on top (a rank's share is never cut larger than the nominal cap allows a whole problem's)
        if cap < min(cap_nominal, max(self.shard.count, 1)):
            log.warning(f"chunk group capped at {cap} chunks by the device's free memory (the nominal cap from its total memory is {cap_nominal}); "
                        "results are unchanged, launches are smaller")
        G = group_size(self.shard.count, want, cap=min(cap, cap_nominal))
        log.info(f"chunk groups: {G} chunks per launch on this rank ({self.shard.count} chunks), K-slice counts sized for a nominal group of {nominal}")
        self.engine = Engine(model, X.shape[-1], self.chunk_pad, G, compute_dtype=self.dtype, device=self.device, fd_sets=fd_sets,
                             arena_align=64 * self.world, chunk_valid=self.chunk, nominal_group=nominal,
                             f32_split=str(cfg.impl.get("engine", {}).get("fd_arithmetic", "bf16x6")) if fd else None)
        self.engine.label_smoothing = getattr(self.loss_fn, "smoothing", 0.0)
        self.engine.only_incorrect = getattr(self.loss_fn, "only_incorrect", False)
        stem = self.engine.plan.stem
        lo, hi = self.shard.first * self.chunk, (self.shard.first + self.shard.count) * self.chunk
        self.patches = torch.zeros(self.shard.count * self.chunk_pad, stem.hout, stem.wout, stem.cin_pad, device=self.device,
                                   dtype=self.dtype) if hi > lo else None
        self.augment = _device_augmentation(cfg, trainloader)
        if self.shuffler is not None:
            if self.augment is not None:
                raise NotImplementedError("a shuffling train loader together with impl.engine.device_augment")
            self._all_images, self._all_labels = Xall.float().contiguous(), Yall
        mine = X[lo:hi].to(self.device) if (hi > lo and self.shuffler is None) else None
        self.images = mine.float().contiguous() if (self.augment is not None and mine is not None) else None   # base images stay resident
        self._aug_step, self._n_total, self._lo = 0, X.shape[0], lo
        if mine is not None and self.augment is None:
            self._gather_patches(mine)
        del mine
        self.labels = self._pad_labels(Y[lo:hi].to(self.device)).clone()      # (a buffer of its own: a shuffling feed rewrites it every step)
        # a validation DataLoader is read once through a private loader (its generator untouched); every validation pass then advances
        # that generator by the one base-seed draw the reference's pass over it makes (keeps shared generators aligned)
        self._valid_loader = validloader if isinstance(validloader, torch.utils.data.DataLoader) else None
        if self._valid_loader is not None and isinstance(validloader.sampler, torch.utils.data.SequentialSampler):
            self.valid = _stage_dataset(validloader, self.device)
        else:
            self.valid = _stage(validloader, self.device) if validloader is not None else None
        self._pending, self._pinned, self._flushing = [], [], False
        self.stats = _Stats(self.flush_stats)
        self.enqueue_times = []

    # ------------------------------------------------------------------------------------------------------------------
    def step(self):
        """One optimizer step = the reference's ``optimizer.step(gradient_evaluation)`` (training.py:226-237)."""
        cfg, eng, hyp = self.cfg, self.engine, self.cfg.hyp
        train_time = time.time()
        # reference `train_time` = wall time of the gradient evaluation (training.py:112,122), which ends in a device sync there.  Here the host
        # runs a step ahead of the GPU and a step's statistics are read back while the next one is queued, so the step's duration is taken
        # on the DEVICE: from this event (it follows the previous step's last kernel in stream order) to the read-back of the statistics
        self._step_begin = torch.cuda.Event(enable_timing=True)
        self._step_begin.record()
        if self.augment is not None and self.images is not None:
            self._regather_augmented()
        if self.shuffler is not None:
            self._regather_shuffled()
        lr = self.optimizer.param_groups[0]["lr"]
        gr = hyp.grad_reg
        if self.multi:
            self._running0 = torch.stack([eng.running_mean, eng.running_var]).clone()
        mod = hyp.optim_modification.name

        # more than one rank, plain step: the gradient exchange leaves in two buckets, the late one (last stage + classifier) as soon as
        # the last backward pass has left the last stage (parallel.BucketExchange, Engine.full_gradient(late_bucket=...))
        noisy = hyp.grad_noise["additive"] is not None or hyp.grad_noise["multiplicative"] is not None
        replicated = self.multi and (mod == "SAM" or hyp.norm_bias.strength > 0 or hyp.only_linear_layers_weight_decay or noisy
                                     or (hyp.grad_clip is not None and float(hyp.grad_clip_norm) != 2.0))
        exchange = late = None
        if self.multi and not replicated:
            from .parallel import BucketExchange, exchange_bounds, shard_ops
            exchange = BucketExchange(eng.avg, eng.theta, self.shard, shard_ops(self, lr, 0.0 if mod in ("LARS", "LARC") else None),
                                      exchange_bounds(self))
            self._last_exchange = exchange
            if self.shard.count > 0 and os.environ.get("FB_EXCHANGE_OVERLAP", "1") != "0":
                late = (exchange.bounds[1], lambda: exchange.start(1))

        def closure():
            """``gradient_evaluation`` (reference training.py:217-225) up to the clip, which is fused into the consumer of ``eng.avg``."""
            hook = None
            if self.multi and gr.acc_strength != 0:
                from .parallel import reduce_pre_pass
                hook = lambda: reduce_pre_pass(self)          # noqa: E731
            out = eng.full_gradient(self.patches, self.labels, lr, gr.block_strength, gr.eps, gr.implementation, acc_strength=gr.acc_strength,
                                    after_pre_pass=hook, pre_block=self.block // self.chunk * self.chunk_pad, batch_clip=hyp.batch_clip,
                                    late_bucket=late)
            self._pre_sqnorm = None
            if gr.acc_strength != 0:             # |pre_grads|^2 for full_loss (reference training.py:98-101)
                lib.call("fb_mt_norms2", eng.pre.data_ptr(), None, eng.plan.P, eng.norms2.data_ptr(), eng.mt_ws.data_ptr())
                self._pre_sqnorm = eng.norms2[0:1].clone()
            return out

        # more than one rank: the plain step shards the update (reduce-scatter, shard-local clip + SGD, all-gather); the options that
        # need the whole averaged gradient on every rank all-reduce it instead and then run the 1-process code below, replicated
        if replicated:
            from .parallel import replicated_reduce
            local_closure = closure

            def closure():                       # noqa: F811  (the closure of the replicated path includes the exchange)
                self._running0 = torch.stack([eng.running_mean, eng.running_var]).clone()
                return replicated_reduce(self, *local_closure())

        loss_k, correct_k, sq_k = closure()
        if self.multi and not replicated:
            from .parallel import sharded_update
            loss_k, correct_k, sq_k = sharded_update(self, loss_k, correct_k, sq_k, lr,
                                                    weight_decay=0.0 if mod in ("LARS", "LARC") else None, exchange=exchange)
            self._state_is_sharded = True        # momentum / clipped gradient complete only on each rank's shard until gather_state()
        else:
            o = hyp.optim

            def modify():
                """``_modify_gradient_params`` up to the clip coefficient (reference training.py:187-204): norm bias on the averaged
                gradient, then the clip norm (L2 or L-infinity) into ``eng.norms2[0]``; ``eng.norms2[1]`` = |theta|^2."""
                eng.grad_and_param_sqnorm()
                if hyp.norm_bias.strength > 0:
                    eng.norm_bias(hyp.norm_bias.strength, hyp.norm_bias.norm_type, hyp.norm_bias.bias)
                    eng.grad_and_param_sqnorm()
                if hyp.grad_clip is not None and float(hyp.grad_clip_norm) == float("inf"):
                    eng.clip_norm_inf()
                elif hyp.grad_clip is not None and float(hyp.grad_clip_norm) != 2.0:
                    eng.clip_norm_p(float(hyp.grad_clip_norm))
                if not noisy:
                    return eng.norms2
                # gradient noise acts on the clipped gradient (reference training.py:205-215): clip in place, then one randn_like per
                # parameter and kind, in the reference's order; the statistics keep the pre-clip norm
                norms = eng.norms2.clone()
                if hyp.grad_clip is not None:
                    eng.apply_clip(hyp.grad_clip)
                for kind in ("additive", "multiplicative"):
                    if hyp.grad_noise[kind] is not None:
                        flat = eng.flatten([torch.randn_like(p) for p in self.model.parameters()])
                        if self.multi:       # every rank must add the same noise: rank 0's draw
                            torch.distributed.broadcast(flat, src=0)
                        eng.grad_noise(flat, hyp.grad_noise[kind], kind == "multiplicative")
                return norms

            clip = None if noisy else hyp.grad_clip          # (already applied in place where noise follows it)
            norms = modify()
            if mod == "SAM":                     # sam.py:84-92: closure, first_step, closure, second_step; stats are recorded twice
                self._record_stats(loss_k, correct_k, sq_k, norms, lr, train_time)
                if noisy:
                    eng.grad_and_param_sqnorm()  # the ascent step is normalised by the norm of the final (clipped, noisy) gradient
                eng.sam_ascent(self.optimizer.rho, clip)
                loss_k, correct_k, sq_k = closure()
                norms = modify()                 # param_norm of the second record is taken at theta + e_w, like the reference's
                self._record_stats(loss_k, correct_k, sq_k, norms, lr, train_time)
                eng.sam_restore()
                self._update(lr, grad_clip=clip)
                self.scheduler.step()
                return
            # LARS / LARC: the wrapper zeroes the weight decay around SGD.step(closure) and nothing else survives the closure (see LARS)
            self._update(lr, zero_wd=mod in ("LARS", "LARC"), grad_clip=clip)
            self._record_stats(loss_k, correct_k, sq_k, norms, lr, train_time)
            self.scheduler.step()
            return
        self._record_stats(loss_k, correct_k, sq_k, eng.norms2, lr, train_time)
        self.scheduler.step()

    def _gather_patches(self, images, aug=None):
        """Stem patches of this rank's images (chunk order) into ``self.patches``; with a padded chunk size every chunk's images go to
        the head of its ``chunk_pad`` rows and the padding rows are (re)zeroed."""
        stem = self.engine.plan.stem
        if self.chunk_pad == self.chunk:
            stem_patches(images, stem, self.dtype, aug=aug, out=self.patches)
            return
        dense = stem_patches(images, stem, self.dtype, aug=aug)
        rows = self.patches.view(self.shard.count, self.chunk_pad, *self.patches.shape[1:])
        rows[:, :self.chunk].copy_(dense.view(self.shard.count, self.chunk, *dense.shape[1:]))
        rows[:, self.chunk:].zero_()

    def _pad_labels(self, labels):
        if self.chunk_pad == self.chunk:
            return labels.contiguous()
        out = torch.full((self.shard.count, self.chunk_pad), -1, dtype=torch.long, device=labels.device)
        out[:, :self.chunk] = labels.view(self.shard.count, self.chunk)
        return out.reshape(-1)

    def gather_state(self):
        """Make the momentum and the averaged-gradient arenas whole on every rank again after sharded updates (a collective; no-op
        for one process or after a replicated update)."""
        if getattr(self, "_state_is_sharded", False):
            from .parallel import gather_sharded_state
            gather_sharded_state(self)
            self._state_is_sharded = False

    def _update(self, lr, zero_wd=False, grad_clip=None):
        """Clip + Nesterov SGD on the arena; per-tensor weight decay when the optimizer has one param group per tensor
        (``hyp.only_linear_layers_weight_decay``, reference optimizers.py:14-21)."""
        eng, hyp, o = self.engine, self.cfg.hyp, self.cfg.hyp.optim
        if hyp.only_linear_layers_weight_decay and not zero_wd:
            wds = [g["weight_decay"] for g in self.optimizer.param_groups]
            eng.sgd_step_per_tensor(lr, wds, o.momentum, o.dampening, o.nesterov, grad_clip)
        else:
            eng.sgd_step(lr, 0.0 if zero_wd else o.weight_decay, o.momentum, o.dampening, o.nesterov, grad_clip)

    def _regather_shuffled(self):
        """A new pass over the shuffling train loader: its batch sampler gives this step's sample order (rank 0's draw on several
        ranks); this rank's slice of it is gathered from the resident dataset and turned into stem patches again."""
        idx = _pass_indices(self.shuffler).to(self.device)
        if self.multi:
            torch.distributed.broadcast(idx, src=0)
        lo, hi = self._lo, self._lo + self.shard.count * self.chunk
        mine = idx[lo:hi]
        if hi > lo:
            self._gather_patches(self._all_images.index_select(0, mine))
            # (into the resident buffer: the recorded launches of a chunk group are keyed by -- and hold -- its address)
            self.labels.copy_(self._pad_labels(self._all_labels.index_select(0, mine)))

    def _regather_augmented(self):
        """A fresh RandomCrop offset / flip per image and step (the reference draws them in its DataLoader workers once per epoch =
        step); drawn for the WHOLE dataset from one seeded CPU generator so that a rank's images get the same augmentation however
        the chunks are sharded, then the stem patches of this rank's images are gathered again on the device."""
        a = self.augment
        seed = self.cfg.seed if getattr(self.cfg, "seed", None) is not None else 0
        gen = torch.Generator().manual_seed(1_000_003 * int(seed) + self._aug_step)
        self._aug_step += 1
        n, lo = self.images.shape[0], self._lo
        oy = ox = fl = None
        if a["crop_pad"] > 0:
            off = torch.randint(0, 2 * a["crop_pad"] + 1, (2, self._n_total), generator=gen, dtype=torch.int8)
            oy, ox = (off[i, lo:lo + n].to(self.device) for i in range(2))
        if a["flip_p"] > 0:
            fl = (torch.rand(self._n_total, generator=gen) < a["flip_p"]).to(torch.int8)[lo:lo + n].to(self.device)
        self._gather_patches(self.images, aug=(oy, ox, fl, a["crop_pad"], a["pad_value"]))

    def _record_stats(self, loss_k, correct_k, sq_k, norms2, lr, train_time):
        """Queue the read-back of one closure's statistics (per-chunk losses / hits / squared gradient norms, the two global norms, the
        pre-pass norm, the per-chunk clip count) and finish the records of earlier closures, whose kernels are long done."""
        hyp = self.cfg.hyp
        self.enqueue_times.append(time.time() - train_time)         # everything of the step is queued
        pre2 = self._pre_sqnorm if getattr(self, "_pre_sqnorm", None) is not None else torch.zeros(1, device=norms2.device)
        clipped = torch.zeros(1, device=norms2.device)
        if hyp.batch_clip is not None:           # the count the reference means to log (its own line, training.py:118, dies with a NameError)
            clipped = self.engine.clipped_all.sum().reshape(1)
            if self.multi:
                torch.distributed.all_reduce(clipped)
        dev = torch.cat([loss_k, correct_k, sq_k, norms2, pre2, clipped])
        host = self._pinned.pop() if self._pinned and self._pinned[-1].numel() == dev.numel() else torch.empty(dev.numel(), dtype=torch.float32, pin_memory=True)
        host.copy_(dev, non_blocking=True)
        done = torch.cuda.Event(enable_timing=True)
        done.record()
        self._pending.append((host, done, lr, (self._step_begin, done)))
        self.flush_stats(keep=1)

    def flush_stats(self, keep=0):
        """Finish the pending statistics records (all but the ``keep`` newest): wait for their read-back, then the host-side formulas."""
        if self._flushing:
            return
        self._flushing = True
        try:
            while len(self._pending) > keep:
                host, done, lr, train_time = self._pending.pop(0)
                done.synchronize()
                self._finish_record(host, lr, train_time)
                self._pinned.append(host)
            if keep == 0:
                self.engine.check_device_errors()
        finally:
            self._flushing = False

    def _finish_record(self, host, lr, train_time):
        """Same keys/formulas as reference training.py:85-119 and :205-211."""
        hyp, stats = self.cfg.hyp, self.stats
        K = self.n_chunks
        loss_k, correct_k, sq_k, (gn2, pn2), pre2, clipped = (host[:K], host[K:2 * K], host[2 * K:3 * K], host[3 * K:3 * K + 2], host[3 * K + 2],
                                                              host[3 * K + 3])
        for idx, entry in enumerate(sq_k.sqrt().tolist()):
            stats.raw(f"grad_norm_train_{idx}").append(entry)
        # sequential fp32 sum, like the reference's `step_loss += chunk_loss` (np.cumsum accumulates strictly left to right)
        step_loss = torch.tensor(float(np.cumsum(loss_k.numpy(), dtype=np.float32)[-1])) if K > 0 else torch.zeros(())
        full_grad_norm = sq_k.mean()
        param_norm = pn2
        train_loss = step_loss / K
        full_loss = train_loss + 0.5 * getattr(hyp.optim, "weight_decay", 0.0) * param_norm
        if hyp.grad_reg.block_strength != 0:
            full_loss = full_loss + lr / 4 * hyp.grad_reg.block_strength * full_grad_norm
        if hyp.grad_reg.acc_strength != 0:
            full_loss = full_loss + lr / 4 * hyp.grad_reg.acc_strength * pre2
        stats.raw("train_loss").append(train_loss.item())
        stats.raw("train_acc").append(correct_k.sum().item() / self.datapoints)
        stats.raw("train_time").append(train_time[0].elapsed_time(train_time[1]) * 1e-3)
        stats.raw("param_norm").append(param_norm.item())
        stats.raw("grad_norm").append(full_grad_norm.sqrt().item())
        stats.raw("full_loss").append(full_loss.item())
        if hyp.batch_clip is not None:
            stats.raw("clipped_batches").append(int(clipped.item()))
        if hyp.grad_clip is not None:
            grad_norm = gn2.sqrt().item()
            stats.raw("preclip_gradnorm").append(grad_norm)
            stats.raw("clipped_step").append(1 if grad_norm > hyp.grad_clip else 0)

    def evaluate(self, stats=None):
        """Reference training.py:343-388: BN in eval mode, mean CE and accuracy over the validation set (optionally with mirrored inputs)."""
        stats = self.stats if stats is None else stats
        if self.valid is None:
            return stats
        X, Y = self.valid
        if getattr(self, "_valid_loader", None) is not None:     # the reference's pass over the validation loader draws one base seed
            torch.empty((), dtype=torch.int64).random_(generator=self._valid_loader.generator)
        eng = self.engine
        cap = eng.G * eng.chunk
        loss_sum, correct, n = 0.0, 0.0, 0
        ema = bool(self.cfg.hyp.evaluate_ema) and getattr(eng, "theta_ema", None) is not None
        if ema:                                  # eval_model = ema_model (reference training.py:290-292)
            eng.swap_ema()
        try:
            for i in range(0, X.shape[0], cap):
                xb, yb = X[i:i + cap], Y[i:i + cap]
                l, c = _evaluate_batch(eng, xb, yb, bool(self.cfg.hyp.test_time_flips))
                loss_sum += l
                correct += c
                n += yb.shape[0]
                if self.cfg.dryrun:
                    break
        finally:
            if ema:
                eng.swap_ema()
        stats["valid_loss"] += [loss_sum / n]
        stats["valid_acc"] += [correct / n]
        return stats


def evaluate(model, dataloader, stats, setup, impl, hyp, dryrun=False):
    """Validation loss / accuracy of a model (signature of reference training.py:343): BN in eval mode (running statistics),
    mean CE, fraction correct.  Used by verify_model_checkpoint.py; `train` evaluates through its own engine instead."""
    if stats is None:
        stats = defaultdict(list)
    device = setup["device"] if torch.device(setup["device"]).type == "cuda" else torch.device("cuda")
    X, Y = _stage(dataloader, device)
    dtype = torch.bfloat16 if impl.mixed_precision else torch.float32
    batch = min(1024, X.shape[0])
    eng = Engine(model, X.shape[-1], batch, 1, compute_dtype=dtype, device=device)
    loss_sum, correct, n = 0.0, 0.0, 0
    for i in range(0, X.shape[0], batch):
        xb, yb = X[i:i + batch], Y[i:i + batch]
        l, c = _evaluate_batch(eng, xb, yb, bool(getattr(hyp, "test_time_flips", False)))
        loss_sum, correct, n = loss_sum + l, correct + c, n + yb.shape[0]
        if dryrun:
            break
    stats["valid_loss"] += [loss_sum / n]
    stats["valid_acc"] += [correct / n]
    return stats


def status_message(optimizer, stats, step):
    def last(key):
        return stats[key][-1] if len(stats[key]) > 0 else float("NaN")

    return (f'Step: {step:<4}| lr: {optimizer.param_groups[0]["lr"]:.4f} | Time: {stats["train_time"][-1]:4.2f}s |'
            f'TRAIN loss {stats["train_loss"][-1]:7.4f} | TRAIN Acc: {stats["train_acc"][-1]:7.2%} |'
            f'VAL loss {last("valid_loss"):7.4f} | VAL Acc: {last("valid_acc"):7.2%} |')


def _measure_implementation_noise(model, trainloader, validloader, setup, cfg):
    """Floating-point difference between two successive evaluations of the (regularised, clipped) full-batch gradient from the
    same checkpoint -- the protocol of reference training.py:429-600 / measure_floating_point_accuracy.py, same printed lines.
    Every reduction of the engine has a fixed order (no atomics), so the two gradients are bit-identical and the errors are 0.0;
    that is the acceptance test for the deterministic split-K / statistics reductions.  Returns the numbers as a dict."""
    trainer = FullBatchTrainer(model, trainloader, validloader, setup, cfg)
    if trainer.world > 1:
        raise NotImplementedError("_measure_implementation_noise: single-process protocol")
    eng, hyp, optimizer, scheduler = trainer.engine, cfg.hyp, trainer.optimizer, trainer.scheduler

    class Counter:
        step: int = 0

    if cfg.impl.checkpoint.name is None:
        print("Could not load checkpoint. Using newly initalized model.")
        cfg.impl.checkpoint.name = cfg.name
        file = os.path.join(cfg.original_cwd, "checkpoints", cfg.impl.checkpoint.name)
        os.makedirs(os.path.dirname(file), exist_ok=True)
        _save_to_checkpoint(model, optimizer, scheduler, None, Counter, file=file, engine=eng)
    file = os.path.join(cfg.original_cwd, "checkpoints", cfg.impl.checkpoint.name)

    def gradient_evaluation():
        _, model_state, _, _, step = torch.load(file, map_location="cpu", weights_only=False)
        model.load_state_dict(model_state)
        eng.load_from_model(model)                       # parameters and BN buffers as in the checkpoint
        log.info(f"Loaded model checkpoint from step {step} successfully.")
        lr, gr = optimizer.param_groups[0]["lr"], hyp.grad_reg
        loss_k, _, _ = eng.full_gradient(trainer.patches, trainer.labels, lr, gr.block_strength, gr.eps, gr.implementation,
                                         acc_strength=gr.acc_strength)
        if hyp.grad_clip is not None:                    # _modify_gradient_params, clip part (reference :198-211)
            grad_norm = float(eng.grad_and_param_sqnorm()[0].sqrt())
            if grad_norm > hyp.grad_clip:
                lib.call("fb_mt_scale", eng.avg.data_ptr(), eng.plan.P, hyp.grad_clip / (grad_norm + 1e-6))
        return loss_k.mean(), eng.avg.clone()

    loss1, g1 = gradient_evaluation()
    print(f"Completed first pass with loss {loss1.item()}.")
    loss2, g2 = gradient_evaluation()
    print(f"Completed first pass with loss {loss2.item()}.")
    out = dict(loss=[loss1.item(), loss2.item()],
               norm_linf=g1.max().item(), norm_l2=g1.double().pow(2).sum().sqrt().item(), norm_l1=g1.double().abs().sum().item(),
               error_linf=(g1 - g2).max().item(), error_l2=(g1 - g2).double().pow(2).sum().sqrt().item(),
               error_l1=(g1 - g2).double().abs().sum().item())
    print(f"Gradient Norms | L^Inf: {out['norm_linf']} | L2: {out['norm_l2']} | L1: {out['norm_l1']}.")
    print(f"Error in L^inf Norm: Total: {out['error_linf']} | Relative: {out['error_linf'] / out['norm_linf']}.")
    print(f"Error in L^2 Norm: Total: {out['error_l2']} | Relative: {out['error_l2'] / out['norm_l2']}.")
    print(f"Error in L^1 Norm: Total: {out['error_l1']} | Relative: {out['error_l1'] / out['norm_l1']}.")
    return out


def train(model, trainloader, validloader, setup, cfg):
    """Train given model based on implementation details and hyperparameters (signature of reference training.py:50)."""
    trainer = FullBatchTrainer(model, trainloader, validloader, setup, cfg)
    stats, optimizer, scheduler, eng = trainer.stats, trainer.optimizer, trainer.scheduler, trainer.engine

    class Counter:
        step: int = 0

    if cfg.impl.checkpoint.name is not None:
        file = os.path.join(cfg.original_cwd, "checkpoints", cfg.impl.checkpoint.name)
        _load_from_checkpoint(model, optimizer, scheduler, None, Counter, cfg.hyp.steps, device="cpu", file=file, engine=eng)

    if cfg.hyp.evaluate_ema:                     # the reference's deepcopy of the (possibly just loaded) model, training.py:72-73
        eng.ema_init()
    while Counter.step < cfg.hyp.steps:
        trainer.step()
        Counter.step += 1
        if cfg.hyp.evaluate_ema:                 # training.py:289-294: update after every step, evaluate the EMA model
            eng.ema_update(cfg.hyp.eval_ema_momentum)
        if (Counter.step - 1) % cfg.impl.validate_every_nth_step == 0 or Counter.step >= cfg.hyp.steps or cfg.dryrun:
            trainer.evaluate()
        if trainer.rank == 0:
            log.info(status_message(optimizer, stats, Counter.step))
        if not torch.as_tensor(stats["train_loss"][-1]).isfinite():
            log.info("Terminating iterations due to divergence of loss...")
            break
        if cfg.hyp.stop_at_full_training_accuracy > 0:
            if min(stats["train_acc"][-cfg.hyp.stop_at_full_training_accuracy:]) == 1:
                log.info("Terminating training after fitting all datapoints.")
                trainer.evaluate()
                break
        if cfg.impl.checkpoint.name is not None:
            if (Counter.step - 1) % cfg.impl.checkpoint.save_every_nth_step == 0 or Counter.step >= cfg.hyp.steps:
                trainer.gather_state()           # every rank: the saved momentum buffers must be whole (sharded update)
                if trainer.rank == 0:
                    file = os.path.join(cfg.original_cwd, "checkpoints", cfg.impl.checkpoint.name)
                    os.makedirs(os.path.dirname(file), exist_ok=True)
                    _save_to_checkpoint(model, optimizer, scheduler, None, Counter, file=file, engine=eng)
        if cfg.dryrun:
            break
    trainer.gather_state()
    eng.store_to_model(model, with_grad=True)    # closure contract: p.grad holds the last (clipped) full gradient
    _sync_optimizer_state(eng, model, optimizer)
    return stats
