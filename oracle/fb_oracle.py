"""CPU ORACLE for the full-batch gradient-descent hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this file.
The product path (``fullbatchtraining_amd``) never routes through it and fails loudly when the HIP library is
missing.

What it is: a plain-torch (CPU, fp32) restatement of the reference algorithm with *explicit* layer forward and
backward passes (no autograd graph on the path).  Every function cites the reference lines it follows
(paths relative to the reference checkout).  The arithmetic of conv/BN/CE lives in PyTorch ATen, which the
reference itself calls (reference ``setup.cfg:33`` pins ``torch>=1.9``; nothing is vendored), so the
restatement calls the same ATen CPU primitives for the raw contractions (``conv2d``, ``conv2d_input``,
``conv2d_weight``) and spells out everything else (BN statistics/gradients, ReLU masks, pooling, CE,
finite-difference regulariser, running mean, clipping, Nesterov SGD, LR schedules) by hand.

Parity pin: ``tests/golden/make_golden.py`` imports the real reference (``/root/reference``) in the build
container and writes ``tests/golden/*.npz|json``; ``tests/test_oracle_golden.py`` checks this oracle against
those vectors (the reference ships no tests or golden vectors of its own).

``q`` hooks: every function takes an optional quantiser ``q`` (identity by default).  With
``q = bf16_round`` the oracle rounds tensors at exactly the points where the HIP bf16 path stores bf16 in HBM,
so kernels can be checked tightly against an oracle that shares their storage precision, while the fp32 mode
is the one pinned to the reference.
"""
import math
from collections import OrderedDict, defaultdict

import re

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm2d default, reference resnets.py:71
BN_MOMENTUM = 0.1


def identity(t):
    return t


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


# ----------------------------------------------------------------------------------------------------------------------
# Architecture description (reference fullbatch/models/resnets.py:12-40, 43-177, 195-230, 271-316)
# ----------------------------------------------------------------------------------------------------------------------
def resnet_layout(depth):
    """block kind + blocks per stage, reference resnets.py:12-40."""
    table = {
        18: ("basic", [2, 2, 2, 2]),
        34: ("basic", [3, 4, 6, 3]),
        50: ("bottleneck", [3, 4, 6, 3]),
        101: ("bottleneck", [3, 4, 23, 3]),
        152: ("bottleneck", [3, 8, 36, 3]),
        20: ("basic", [3, 3, 3]),
        32: ("basic", [5, 5, 5]),
        56: ("basic", [9, 9, 9]),
        110: ("basic", [18, 18, 18]),
    }
    return table[depth]


class Spec:
    """Static description of the network: ordered conv/bn/fc parameter names with shapes and the block graph."""

    def __init__(self, depth=18, channels=3, classes=10, stem="CIFAR", downsample="C"):
        if downsample != "C":
            raise NotImplementedError("only downsample 'C' (AvgPool -> 1x1 conv -> BN), reference resnets.py:147-152")
        self.depth, self.channels, self.classes, self.stem = depth, channels, classes, stem
        kind, layers = resnet_layout(depth)
        self.kind = kind
        expansion = 1 if kind == "basic" else 4
        self.blocks = []  # dicts: prefix, inplanes, planes, stride, has_down
        inplanes = 64  # reference resnets.py:61 (isinstance on a class is always False -> 64), SURVEY T9
        width = 64
        strides = [1, 2, 2, 2]
        for si, nblocks in enumerate(layers):
            for bi in range(nblocks):
                stride = strides[si] if bi == 0 else 1
                has_down = bi == 0 and (stride != 1 or inplanes != width * expansion)
                self.blocks.append(
                    dict(prefix=f"layers.{si}.{bi}", inplanes=inplanes, planes=width, stride=stride, has_down=has_down)
                )
                inplanes = width * expansion
            width *= 2
        self.feat = inplanes
        self.expansion = expansion

    # parameter order == torch registration order == model.parameters() order of the reference module
    def param_shapes(self):
        out = OrderedDict()
        k = 3 if self.stem == "CIFAR" else 7
        out["stem.0.weight"] = (64, self.channels, k, k)
        out["stem.1.weight"] = (64,)
        out["stem.1.bias"] = (64,)
        for b in self.blocks:
            p, cin, w = b["prefix"], b["inplanes"], b["planes"]
            if self.kind == "basic":
                convs = [("conv1", w, cin, 3), ("conv2", w, w, 3)]
            else:
                convs = [("conv1", w, cin, 1), ("conv2", w, w, 3), ("conv3", w * 4, w, 1)]
            for i, (name, co, ci, ks) in enumerate(convs):
                out[f"{p}.{name}.weight"] = (co, ci, ks, ks)
                out[f"{p}.bn{i + 1}.weight"] = (co,)
                out[f"{p}.bn{i + 1}.bias"] = (co,)
            if b["has_down"]:
                co = w * self.expansion
                out[f"{p}.downsample.1.weight"] = (co, cin, 1, 1)
                out[f"{p}.downsample.2.weight"] = (co,)
                out[f"{p}.downsample.2.bias"] = (co,)
        out["fc.weight"] = (self.classes, self.feat)
        out["fc.bias"] = (self.classes,)
        return out

    def bn_names(self):
        names = ["stem.1"]
        for b in self.blocks:
            n = 2 if self.kind == "basic" else 3
            names += [f"{b['prefix']}.bn{i + 1}" for i in range(n)]
            if b["has_down"]:
                names.append(f"{b['prefix']}.downsample.2")
        return names


def split_state(state):
    """state_dict -> (params OrderedDict in registration order, buffers dict)."""
    params, buffers = OrderedDict(), OrderedDict()
    for key, value in state.items():
        if key.endswith(("running_mean", "running_var", "num_batches_tracked")):
            buffers[key] = value
        else:
            params[key] = value
    return params, buffers


# ----------------------------------------------------------------------------------------------------------------------
# Layer primitives with explicit backward
# ----------------------------------------------------------------------------------------------------------------------
def conv_fwd(x, w, stride, pad):
    return F.conv2d(x, w, None, stride, pad)


def conv_bwd(x, w, dy, stride, pad, need_dx=True):
    dw = torch.nn.grad.conv2d_weight(x, w.shape, dy, stride, pad)
    dx = torch.nn.grad.conv2d_input(x.shape, w, dy, stride, pad) if need_dx else None
    return dx, dw


def bn_train_fwd(x, gamma, beta, buffers, name, update):
    """Training-mode BatchNorm2d (ATen native_batch_norm semantics, reference resnets.py:71,207,210,151).

    Batch mean / biased variance over (N,H,W); running stats updated with the *unbiased* variance, momentum 0.1.
    """
    n = x.numel() // x.shape[1]
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    invstd = torch.rsqrt(var + BN_EPS)
    xhat = (x - mean[None, :, None, None]) * invstd[None, :, None, None]
    y = xhat * gamma[None, :, None, None] + beta[None, :, None, None]
    if update:
        rm, rv = buffers[f"{name}.running_mean"], buffers[f"{name}.running_var"]
        rm.mul_(1 - BN_MOMENTUM).add_(mean.detach(), alpha=BN_MOMENTUM)
        rv.mul_(1 - BN_MOMENTUM).add_(var.detach() * (n / (n - 1)), alpha=BN_MOMENTUM)
        buffers[f"{name}.num_batches_tracked"] += 1
    return y, (xhat, invstd)


def bn_train_bwd(dy, gamma, saved):
    xhat, invstd = saved
    n = dy.numel() // dy.shape[1]
    dbeta = dy.sum(dim=(0, 2, 3))
    dgamma = (dy * xhat).sum(dim=(0, 2, 3))
    dx = (gamma * invstd)[None, :, None, None] * (
        dy - dbeta[None, :, None, None] / n - xhat * dgamma[None, :, None, None] / n
    )
    return dx, dgamma, dbeta


def bn_eval_fwd(x, gamma, beta, buffers, name):
    rm, rv = buffers[f"{name}.running_mean"], buffers[f"{name}.running_var"]
    scale = gamma * torch.rsqrt(rv + BN_EPS)
    return x * scale[None, :, None, None] + (beta - rm * scale)[None, :, None, None]


def avgpool2_fwd(x):
    n, c, h, w = x.shape
    return x.view(n, c, h // 2, 2, w // 2, 2).mean(dim=(3, 5))


def avgpool2_bwd(dy):
    return dy.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3) * 0.25


def cross_entropy_fwd_bwd(logits, labels, smoothing=0.0, only_incorrect=False):
    """mean-reduced CE = log_softmax + nll_loss (reference training.py:403, used :79) and its gradient.

    ``smoothing``: LabelSmoothCrossEntropyLoss (reference modules.py:86-101): target weight 1-s, s/(C-1) on the other classes.
    ``only_incorrect``: IncorrectCrossEntropyLoss (modules.py:104-119): samples the model already classifies correctly contribute
    zero loss (and gradient); the mean still runs over all samples."""
    n, classes = logits.shape
    m = logits.max(dim=1, keepdim=True).values
    z = logits - m
    lse = z.exp().sum(dim=1, keepdim=True).log()
    logp = z - lse
    weight = torch.full_like(logits, smoothing / (classes - 1.0))
    weight[torch.arange(n, device=logits.device), labels] = 1.0 - smoothing
    hit = logits.argmax(dim=-1) == labels
    keep = (~hit).to(logits.dtype) if only_incorrect else torch.ones(n, dtype=logits.dtype, device=logits.device)
    loss = ((-weight * logp).sum(dim=-1) * keep).mean()
    dlogits = (logp.exp() - weight) * keep[:, None] / n
    correct = hit.float().sum()  # reference training.py:80
    return loss, correct, dlogits


# ----------------------------------------------------------------------------------------------------------------------
# Network forward (+ saved tensors) and backward.  Reference resnets.py:179-192 (ResNet), 214-230 (BasicBlock),
# 296-316 (Bottleneck), 147-152 (downsample C).
# ----------------------------------------------------------------------------------------------------------------------
def _conv_bn(x, params, buffers, conv_name, bn_name, stride, pad, q, update_bn, train=True):
    w = q(params[f"{conv_name}.weight"])
    raw = conv_fwd(x, w, stride, pad)
    gamma, beta = params[f"{bn_name}.weight"], params[f"{bn_name}.bias"]
    if train:
        # the HIP path takes batch statistics from the fp32 accumulators and stores the conv output rounded
        y, saved = bn_train_fwd(raw, gamma, beta, buffers, bn_name, update_bn)
        if q is not identity:
            rawq = q(raw)
            mean = raw.mean(dim=(0, 2, 3))
            xhat = (rawq - mean[None, :, None, None]) * saved[1][None, :, None, None]
            y = xhat * gamma[None, :, None, None] + beta[None, :, None, None]
            saved = (xhat, saved[1])
    else:
        y, saved = bn_eval_fwd(q(raw), gamma, beta, buffers, bn_name), None
    return y, dict(x=x, w=w, stride=stride, pad=pad, bn=saved, gamma=gamma, conv=conv_name, bnn=bn_name)


def forward(spec, params, buffers, x, q=identity, update_bn=True, train=True):
    """Returns logits and the tape needed by ``backward``."""
    tape = []
    a = q(x)
    if spec.stem == "CIFAR":
        y, rec = _conv_bn(a, params, buffers, "stem.0", "stem.1", 1, 1, q, update_bn, train)
        a = q(torch.relu(y))
        tape.append(dict(kind="stem", rec=rec, out=a))
    else:  # 'standard': 7x7 s2 conv, BN, ReLU, MaxPool(3, 2, 1) (reference resnets.py:74-79)
        y, rec = _conv_bn(a, params, buffers, "stem.0", "stem.1", 2, 3, q, update_bn, train)
        r = q(torch.relu(y))
        a, idx = F.max_pool2d(r, 3, 2, 1, return_indices=True)
        tape.append(dict(kind="stem_std", rec=rec, relu_out=r, pool_idx=idx, out=a))
    for b in spec.blocks:
        p = b["prefix"]
        a0 = a
        recs = []
        if spec.kind == "basic":
            y, r1 = _conv_bn(a0, params, buffers, f"{p}.conv1", f"{p}.bn1", b["stride"], 1, q, update_bn, train)
            a1 = q(torch.relu(y))
            y, r2 = _conv_bn(a1, params, buffers, f"{p}.conv2", f"{p}.bn2", 1, 1, q, update_bn, train)
            recs, mids = [r1, r2], [a1]
        else:
            y, r1 = _conv_bn(a0, params, buffers, f"{p}.conv1", f"{p}.bn1", 1, 0, q, update_bn, train)
            a1 = q(torch.relu(y))
            y, r2 = _conv_bn(a1, params, buffers, f"{p}.conv2", f"{p}.bn2", b["stride"], 1, q, update_bn, train)
            a2 = q(torch.relu(y))
            y, r3 = _conv_bn(a2, params, buffers, f"{p}.conv3", f"{p}.bn3", 1, 0, q, update_bn, train)
            recs, mids = [r1, r2, r3], [a1, a2]
        rd = None
        if b["has_down"]:
            pooled = q(avgpool2_fwd(a0)) if b["stride"] == 2 else a0
            idn, rd = _conv_bn(pooled, params, buffers, f"{p}.downsample.1", f"{p}.downsample.2", 1, 0, q, update_bn, train)
        else:
            idn = a0
        a = q(torch.relu(y + idn))
        tape.append(dict(kind="block", b=b, recs=recs, mids=mids, rd=rd, out=a))
    feat = a.mean(dim=(2, 3))  # AdaptiveAvgPool2d((1,1)) + flatten, reference resnets.py:185-186
    logits = feat @ params["fc.weight"].t() + params["fc.bias"]  # reference resnets.py:187
    tape.append(dict(kind="head", feat=feat, spatial=a.shape[2] * a.shape[3], shape=a.shape))
    return logits, tape


def _conv_bn_bwd(dy, rec, grads, q, need_dx=True):
    """dy = gradient w.r.t. the BN output.  Fills grads for conv weight, gamma, beta; returns grad w.r.t. conv input."""
    dxc, dgamma, dbeta = bn_train_bwd(dy, rec["gamma"], rec["bn"])
    dxc = q(dxc)
    dx, dw = conv_bwd(rec["x"], rec["w"], dxc, rec["stride"], rec["pad"], need_dx)
    grads[f"{rec['conv']}.weight"] = dw
    grads[f"{rec['bnn']}.weight"] = dgamma
    grads[f"{rec['bnn']}.bias"] = dbeta
    return dx


def backward(spec, params, tape, dlogits, q=identity):
    grads = {}
    head = tape[-1]
    grads["fc.weight"] = dlogits.t() @ head["feat"]
    grads["fc.bias"] = dlogits.sum(dim=0)
    dfeat = dlogits @ params["fc.weight"]
    da = q((dfeat / head["spatial"])[:, :, None, None].expand(head["shape"]).contiguous())
    for entry in reversed(tape[:-1]):
        if entry["kind"] == "block":
            b = entry["b"]
            dy = q(da * (entry["out"] > 0))  # ReLU mask of the block output; feeds the last BN and the shortcut
            recs, mids = entry["recs"], entry["mids"]
            d = _conv_bn_bwd(dy, recs[-1], grads, q)
            for rec, mid in zip(reversed(recs[:-1]), reversed(mids)):
                d = q(d)
                d = _conv_bn_bwd(q(d * (mid > 0)), rec, grads, q)
            if entry["rd"] is not None:
                dp = q(_conv_bn_bwd(dy, entry["rd"], grads, q))
                d = d + (avgpool2_bwd(dp) if b["stride"] == 2 else dp)
            else:
                d = d + dy
            da = q(d)
        elif entry["kind"] == "stem":
            dy = q(da * (entry["out"] > 0))
            _conv_bn_bwd(dy, entry["rec"], grads, q, need_dx=False)
        elif entry["kind"] == "stem_std":
            r = entry["relu_out"]
            dr = torch.zeros_like(r).flatten(2)
            dr.scatter_add_(2, entry["pool_idx"].flatten(2), da.flatten(2))
            dy = q(dr.view_as(r) * (r > 0))
            _conv_bn_bwd(dy, entry["rec"], grads, q, need_dx=False)
    return grads


def chunk_gradient(spec, params, buffers, x, y, q=identity, update_bn=True):
    """Restates ``_compute_batched_gradient`` (reference training.py:76-83): fwd, CE, #correct, gradient list."""
    logits, tape = forward(spec, params, buffers, x, q, update_bn, train=True)
    loss, correct, dlogits = cross_entropy_fwd_bwd(logits, y, getattr(spec, "label_smoothing", 0.0), getattr(spec, "only_incorrect", False))
    grads = backward(spec, params, tape, dlogits, q)
    return [grads[name] for name in params], loss, correct


def chunk_gradient_autograd(spec, params, buffers, x, y, q=identity, update_bn=True):
    """The same quantity as ``chunk_gradient`` obtained the way the reference obtains it (training.py:76-83): the forward of this file
    recorded by autograd, then ``torch.autograd.grad(loss, parameters)`` -- torch's own (oneDNN) backward kernels instead of the explicit
    layer backward above.  Checked equal to the explicit path in tests/test_oracle_golden.py; it is the faster way to get a weight
    gradient on CPU (``torch.nn.grad.conv2d_weight`` is slow), so bench.py's ``cpu_baseline`` times this one."""
    leaves = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in params.items())
    logits, _ = forward(spec, leaves, buffers, x, q, update_bn, train=True)
    loss, correct, _ = cross_entropy_fwd_bwd(logits, y, getattr(spec, "label_smoothing", 0.0), getattr(spec, "only_incorrect", False))
    grads = torch.autograd.grad(loss, list(leaves.values()))
    return [g.detach() for g in grads], loss.detach(), correct


def sqnorm(tensors):
    """``torch.stack([g.pow(2).sum() for g in grads]).sum()`` -- reference training.py:162, modules.py:223."""
    return torch.stack([t.pow(2).sum() for t in tensors]).sum()


# ----------------------------------------------------------------------------------------------------------------------
# GradRegularizer (reference fullbatch/models/modules.py:136-348)
# ----------------------------------------------------------------------------------------------------------------------
IMPLEMENTATIONS = (
    "autograd-pen", "autograd", "central-differences", "complex-step", "forward-differences", "forward-differences-legacy",
)


def gradreg(spec, params, buffers, grads, x, y, lr, block_strength, eps, implementation, q=identity, acc_strength=0.0, pre_grads=None,
            chunk_gradient=chunk_gradient):
    """In-place modification of ``grads`` for one chunk; mirrors modules.py:211-241 / 243-264 / 266-300.

    Parameters are perturbed in place and restored exactly like the reference (clone/copy, or subtract for legacy).
    BN running statistics are updated again by every extra forward.  ``pre_grads`` (the full gradient of the
    acc_strength pre-pass, training.py:128-142) joins the finite-difference direction: v = block_strength*g + acc_strength*pre
    (modules.py:217-221, 273-275); the legacy variant disregards it (modules.py:243-245).
    """
    if block_strength == 0 and acc_strength == 0:
        return grads  # _pass, modules.py:150-152,177-178
    if implementation not in IMPLEMENTATIONS:
        raise ValueError(f"Invalid spec. given for regularizer implementation: {implementation}")
    plist = list(params.values())
    cf = lr / 4  # modules.py:214
    if implementation == "forward-differences":
        original = [p.clone() for p in plist]
        vec = [g * block_strength for g in grads]
        if pre_grads is not None:
            for v, pg in zip(vec, pre_grads):
                v.add_(pg, alpha=acc_strength)
        eps_n = eps / sqnorm(vec).sqrt()
        for p, v in zip(plist, vec):
            p.add_(v, alpha=float(eps_n))
        off, _, _ = chunk_gradient(spec, params, buffers, x, y, q)
        for o, g in zip(off, grads):
            o.sub_(g).div_(eps_n)
        for p, o in zip(plist, original):
            p.copy_(o)
        for g, o in zip(grads, off):
            g.add_(o, alpha=cf)
    elif implementation == "forward-differences-legacy":
        cf = lr / 4 * block_strength
        eps_n = eps / sqnorm(grads).sqrt()
        for p, g in zip(plist, grads):
            p.add_(g, alpha=float(eps_n))
        off, _, _ = chunk_gradient(spec, params, buffers, x, y, q)
        for o, g in zip(off, grads):
            o.sub_(g).div_(eps_n)
        for p, g in zip(plist, grads):
            p.sub_(g, alpha=float(eps_n))
        for g, o in zip(grads, off):
            g.add_(o, alpha=cf)
    elif implementation == "central-differences":
        original = [p.clone() for p in plist]
        vec = [g * block_strength for g in grads]
        if pre_grads is not None:
            for v, pg in zip(vec, pre_grads):
                v.add_(pg, alpha=acc_strength)
        eps_n = eps / sqnorm(vec).sqrt()
        for p, v in zip(plist, vec):
            p.add_(v, alpha=float(0.5 * eps_n))
        plus, _, _ = chunk_gradient(spec, params, buffers, x, y, q)
        for p, v in zip(plist, vec):
            p.sub_(v, alpha=float(eps_n))
        minus, _, _ = chunk_gradient(spec, params, buffers, x, y, q)
        vhp = [(a - b) / eps_n for a, b in zip(plus, minus)]
        for p, o in zip(plist, original):
            p.copy_(o)
        for g, o in zip(grads, vhp):
            g.add_(o, alpha=cf)
    else:
        raise NotImplementedError(f"{implementation} needs double backward / complex autograd; out of scope (SURVEY 2.1)")
    return grads


# ----------------------------------------------------------------------------------------------------------------------
# LR schedule (reference optimizers.py:69-93, additional_optimizers/scheduler.py:32-91) -- same recursion as torch
# ----------------------------------------------------------------------------------------------------------------------
class LRSchedule:
    """Warm-up wrapper around torch's chainable CosineAnnealingLR / constant; reproduces the float sequence."""

    def __init__(self, base_lr, scheduler, steps, warmup):
        self.base_lr, self.warmup = base_lr, warmup
        if scheduler == "cosine-decay":
            self.t_max, self.eta_min = steps, 0.0
        elif scheduler == "cosine-decay-floored":
            self.t_max, self.eta_min = steps, base_lr / 25
        elif scheduler == "cosine-4000":
            self.t_max, self.eta_min = 4000, 0.0
        elif scheduler in ("", " ", None):
            self.t_max, self.eta_min = None, 0.0
        else:
            raise ValueError(f"Invalid scheduler {scheduler} provided.")
        self.last_epoch = 0  # warm-up scheduler counter
        self.after_epoch = 0  # wrapped scheduler counter
        self.finished = False
        self.after_lr = base_lr  # the wrapped scheduler initialises the group lr to base_lr
        self.lr = base_lr * (0.0 / warmup) if warmup > 0 else base_lr

    def _after_step(self):
        self.after_epoch += 1
        if self.t_max is None:
            return
        t, lr = self.after_epoch, self.after_lr
        if (t - 1 - self.t_max) % (2 * self.t_max) == 0:
            lr = lr + (self.base_lr - self.eta_min) * (1 - math.cos(math.pi / self.t_max)) / 2
        else:
            lr = (1 + math.cos(math.pi * t / self.t_max)) / (1 + math.cos(math.pi * (t - 1) / self.t_max)) * (
                lr - self.eta_min
            ) + self.eta_min
        self.after_lr = lr

    def step(self):
        if self.warmup <= 0:
            self._after_step()
            self.lr = self.after_lr
            return
        if self.finished:
            self._after_step()
            self.lr = self.after_lr
            return
        self.last_epoch += 1
        if self.last_epoch > self.warmup:
            self.finished = True
            self.lr = self.after_lr
        else:
            self.lr = self.base_lr * (float(self.last_epoch) / self.warmup)


# ----------------------------------------------------------------------------------------------------------------------
# One full-batch step (reference training.py:121-239) and a small training loop
# ----------------------------------------------------------------------------------------------------------------------
def full_batch_step(spec, params, buffers, momentum, X, Y, hyp, lr, stats, chunk, q=identity, chunk_range=None):
    """One optimizer step = ``optimizer.step(gradient_evaluation)`` (reference training.py:226-237).

    ``hyp``: dict(weight_decay, momentum, nesterov, dampening, block_strength, eps, implementation, grad_clip[, acc_strength,
    optim_modification=dict(name=...)]).  ``chunk``: images per chunk.  Returns the averaged (and clipped) gradient list the update used.

    ``optim_modification`` (reference optimizers.py:57-67):
      * none: torch SGD evaluates the closure once, then steps.
      * SAM (additional_optimizers/sam.py:84-92): closure -> e = rho * g / (|g| + 1e-12) (g already clipped by the closure) -> p += e ->
        closure again (stats recorded a second time, BN buffers updated again) -> p -= e -> SGD step with the second gradient.
      * LARS / LARC (additional_optimizers/lars.py:61-94): the wrapper rescales ``p.grad`` -- the gradients of the PREVIOUS step -- zeroes
        the group weight decay, and only then calls ``SGD.step(closure)``; the closure assigns fresh ``p.grad`` tensors
        (training.py:183-184), so the rescaling never reaches the update and the step is plain SGD WITHOUT weight decay.
    """
    mod = (hyp.get("optim_modification") or {}).get("name", "none")

    def closure():
        return _gradient_evaluation(spec, params, buffers, X, Y, hyp, lr, stats, chunk, q, chunk_range)

    if mod == "SAM":
        g1 = closure()
        grad_norm = torch.norm(torch.stack([g.norm(p=2) for g in g1]), p=2)          # sam.py:94-104
        scale = hyp["optim_modification"]["rho"] / (grad_norm + 1e-12)
        e_w = [g * scale for g in g1]
        for p, e in zip(params.values(), e_w):
            p.add_(e)
        avg = closure()
        for p, e in zip(params.values(), e_w):
            p.sub_(e)
        sgd_step(params, avg, momentum, lr, hyp)
    elif mod in ("LARS", "LARC"):
        avg = closure()                       # replaces whatever the wrapper did to the stale gradients
        sgd_step(params, avg, momentum, lr, dict(hyp, weight_decay=0.0))
    elif mod == "none":
        avg = closure()
        sgd_step(params, avg, momentum, lr, hyp)
    else:
        raise ValueError(mod)
    return avg


def _gradient_evaluation(spec, params, buffers, X, Y, hyp, lr, stats, chunk, q=identity, chunk_range=None):
    """The closure (reference training.py:217-225): ``_accumulate_full_gradient`` + ``_record_stats`` + ``_modify_gradient_params``."""
    names = list(params)
    n_chunks = X.shape[0] // chunk  # drop_last=True, reference data_preparation.py:68
    acc = hyp.get("acc_strength", 0.0)
    pre = None
    if acc != 0:  # pre-pass, training.py:128-142: plain full gradient (running mean over WHOLE blocks of data.batch_size images --
        # BN batches of their own when the main loop cuts blocks into sub_batch chunks); a train-mode pass of its own
        pre = [torch.zeros_like(p) for p in params.values()]
        block = hyp.get("block", chunk)
        n_blocks = X.shape[0] // block
        block_ids = range(n_blocks) if chunk_range is None else sorted({k * chunk // block for k in chunk_range})
        for counter, b in enumerate(block_ids):
            g0, _, _ = chunk_gradient(spec, params, buffers, X[b * block:(b + 1) * block], Y[b * block:(b + 1) * block], q)
            if hyp.get("batch_clip") is not None:  # training.py:138-139
                clip_gradient_list(g0, hyp["batch_clip"], hyp.get("grad_clip_norm", 2))
            for a, g in zip(pre, g0):
                g.sub_(a)
                a.add_(g, alpha=1 / (counter + 1))
    avg = [torch.zeros_like(p) for p in params.values()]
    grad_norms = torch.zeros(n_chunks, dtype=avg[0].dtype, device=avg[0].device)
    step_loss, step_preds, datapoints, clipped_batches = 0.0, 0.0, 0, 0
    ks = range(n_chunks) if chunk_range is None else chunk_range
    for counter, k in enumerate(ks):
        xk, yk = X[k * chunk:(k + 1) * chunk], Y[k * chunk:(k + 1) * chunk]
        datapoints += chunk
        grads, loss, correct = chunk_gradient(spec, params, buffers, xk, yk, q)
        grad_norms[k] = sqnorm(grads)  # training.py:162
        grads = gradreg(spec, params, buffers, grads, xk, yk, lr, hyp["block_strength"], hyp["eps"],
                        hyp["implementation"], q, acc, pre)  # training.py:163
        if hyp.get("batch_clip") is not None:  # training.py:166-167: the REGULARISED chunk gradient is clipped before it is averaged
            clipped_batches += clip_gradient_list(grads, hyp["batch_clip"], hyp.get("grad_clip_norm", 2))
        for a, g in zip(avg, grads):  # _stable_mean_accumulation, training.py:45-47
            g.sub_(a)
            a.add_(g, alpha=1 / (counter + 1))
        step_loss = step_loss + loss
        step_preds = step_preds + correct
    # _record_stats, training.py:85-119 (num_blocks == n_chunks for one chunk per block)
    for idx, entry in enumerate(grad_norms.sqrt().tolist()):
        stats[f"grad_norm_train_{idx}"].append(entry)
    param_norm = sum(p.pow(2).sum() for p in params.values())
    full_grad_norm = grad_norms.mean()
    full_loss = step_loss / n_chunks + 0.5 * hyp["weight_decay"] * param_norm
    if hyp["block_strength"] != 0:
        full_loss = full_loss + lr / 4 * hyp["block_strength"] * full_grad_norm
    if acc != 0:  # training.py:98-101
        full_loss = full_loss + lr / 4 * acc * sqnorm(pre)
    stats["train_loss"].append(float(step_loss) / n_chunks)
    stats["train_acc"].append(float(step_preds) / datapoints)
    stats["param_norm"].append(float(param_norm))
    stats["grad_norm"].append(float(full_grad_norm.sqrt()))
    stats["full_loss"].append(float(full_loss))
    if hyp.get("batch_clip") is not None:  # what training.py:117-118 means to record (the reference's line raises NameError: the counter is
        stats["clipped_batches"].append(clipped_batches)  # a local of _accumulate_full_gradient, invisible inside _record_stats)
    # _modify_gradient_params, training.py:187-211: external norm bias, then the clip
    nb = hyp.get("norm_bias") or {}
    if nb.get("strength", 0.0) > 0.0:  # training.py:188-196
        param_norm_l2 = sum(p.pow(2).sum() for p in params.values())
        if nb["norm_type"] == 1:
            diff_value_sign = (param_norm_l2 - nb["bias"] ** 2).sign()
            for g in avg:
                g.add_(nb["strength"] * diff_value_sign)
        else:
            factor = 2 * (param_norm_l2 - nb["bias"] ** 2)
            for g, p in zip(avg, params.values()):
                g.add_(nb["strength"] * factor * p)
    if hyp.get("grad_clip") is not None:
        if float(hyp.get("grad_clip_norm", 2)) == float("inf"):  # training.py:199-200
            grad_norm = max(g.abs().max() for g in avg)
        else:  # training.py:201-204: the p-norm of the per-tensor p-norms
            pn = float(hyp.get("grad_clip_norm", 2))
            grad_norm = torch.norm(torch.stack([torch.norm(g, pn) for g in avg]), pn)
        stats["preclip_gradnorm"].append(float(grad_norm))
        if grad_norm > hyp["grad_clip"]:
            for g in avg:
                g.mul_(hyp["grad_clip"] / (grad_norm + 1e-6))
            stats["clipped_step"].append(1)
        else:
            stats["clipped_step"].append(0)
    gn = hyp.get("grad_noise") or {}
    if gn.get("additive") is not None:  # training.py:212-213, Langevin-type noise; one randn_like per parameter, in parameter order
        for g in avg:
            g.add_(gn["additive"] * torch.randn_like(g))
    if gn.get("multiplicative") is not None:  # training.py:214-215
        for g in avg:
            g.mul_(1 + gn["multiplicative"] * torch.randn_like(g))
    return avg


def clip_gradient_list(grads, scaled_clip, norm_type=2, eps=1e-6):
    """``_clip_gradient_list`` (reference training/utils.py:4-19): in-place clip of a gradient list to ``scaled_clip`` in the p-norm of the
    per-tensor p-norms (or the max-abs for inf); returns 1 if it clipped."""
    if float(norm_type) == float("inf"):
        grad_norm = max(g.abs().max() for g in grads)
    else:
        grad_norm = torch.norm(torch.stack([torch.norm(g, float(norm_type)) for g in grads]), float(norm_type))
    if grad_norm > scaled_clip:
        for g in grads:
            g.mul_(scaled_clip / (grad_norm + eps))
        return 1
    return 0


def sgd_step(params, grads, momentum, lr, hyp):
    """torch.optim.SGD semantics (weight decay, momentum w/ first-step buffer = d, dampening, Nesterov)."""
    mu, wd_all, damp = hyp["momentum"], hyp["weight_decay"], hyp.get("dampening", 0.0)
    for i, ((name, p), g) in enumerate(zip(params.items(), grads)):
        # only_linear_layers_weight_decay (reference optimizers.py:14-21): one param group per tensor, no decay on "bias"/"gain" names
        wd = 0.0 if (hyp.get("only_linear_layers_weight_decay") and re.findall("(bias|gain)|skip_gain", name)) else wd_all
        d = g.add(p, alpha=wd) if wd != 0 else g.clone()
        if mu != 0:
            if momentum[i] is None:
                momentum[i] = d.clone()
            else:
                momentum[i].mul_(mu).add_(d, alpha=1 - damp)
            d = d.add(momentum[i], alpha=mu) if hyp.get("nesterov", True) else momentum[i]
        p.add_(d, alpha=-lr)


def evaluate(spec, params, buffers, X, Y, batch=128, q=identity, test_time_flips=False):
    """Restates ``evaluate`` (reference training.py:343-388) for one process.  ``test_time_flips`` (training.py:370-373): the SUM of
    the softmax outputs of the image and of its horizontal mirror goes into the loss function and the argmax in place of logits."""
    step_loss, step_preds, datapoints = 0.0, 0.0, 0
    for i in range(0, X.shape[0], batch):
        xb, yb = X[i:i + batch], Y[i:i + batch]
        logits, _ = forward(spec, params, buffers, xb, q, update_bn=False, train=False)
        if test_time_flips:
            mirrored, _ = forward(spec, params, buffers, torch.flip(xb, [3]), q, update_bn=False, train=False)
            logits = logits.softmax(dim=1) + mirrored.softmax(dim=1)
        loss, correct, _ = cross_entropy_fwd_bwd(logits, yb)
        step_loss += float(loss) * yb.shape[0]
        step_preds += float(correct)
        datapoints += yb.shape[0]
    return step_loss / datapoints, step_preds / datapoints


def train(spec, state, X, Y, hyp, steps, chunk, scheduler="cosine-decay", warmup=0, q=identity, Xv=None, Yv=None,
          validate_every=100, order_fn=None, before_eval=None):
    """Small driver mirroring reference training.py:217-239 + 296-298; mutates ``state`` in place; returns stats.

    ``order_fn(step)``: sample indices of this step's pass over the data in loader order (a shuffling train loader: the reference
    iterates its DataLoader anew every step, training.py:145-147); ``before_eval()``: called before every validation pass (lets a
    test advance a generator the way the reference's validation loader does)."""
    params, buffers = split_state(state)
    momentum = [None] * len(params)
    # get_loss_fn (reference training.py:391-413): the training loss; evaluate() always uses plain cross entropy (training.py:345)
    spec.label_smoothing = float(hyp.get("label_smoothing") or 0.0)
    spec.only_incorrect = hyp.get("loss_modification") == "incorrect-xent"
    sched = LRSchedule(hyp["lr"], scheduler, steps, warmup)
    stats = defaultdict(list)
    ema = hyp.get("evaluate_ema", False)
    if ema:  # training.py:72-73: a deep copy of the model at the start of training
        ema_params = {k: v.clone() for k, v in params.items()}
        ema_buffers = {k: v.clone() for k, v in buffers.items()}
    for step in range(steps):
        Xs, Ys = X, Y
        if order_fn is not None:
            idx = order_fn(step)
            Xs, Ys = X[idx], Y[idx]
        full_batch_step(spec, params, buffers, momentum, Xs, Ys, hyp, sched.lr, stats, chunk, q)
        stats["lr"].append(sched.lr)
        sched.step()
        eval_params, eval_buffers = params, buffers
        if ema:  # _update_ema, training/utils.py:22-29: parameters AND buffers (the long num_batches_tracked truncates on copy_)
            m = hyp["eval_ema_momentum"]
            for src, dst in ((params, ema_params), (buffers, ema_buffers)):
                for k in src:
                    dst[k].copy_(m * dst[k] + (1 - m) * src[k])
            eval_params, eval_buffers = ema_params, ema_buffers
        if Xv is not None and (step % validate_every == 0 or step + 1 >= steps):
            if before_eval is not None:
                before_eval()
            vl, va = evaluate(spec, eval_params, eval_buffers, Xv, Yv, q=q, test_time_flips=hyp.get("test_time_flips", False))
            stats["valid_loss"].append(vl)
            stats["valid_acc"].append(va)
    stats["_momentum"] = momentum
    return stats
