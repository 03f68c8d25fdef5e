"""Drop-in ``GradRegularizer`` (reference ``fullbatch/models/modules.py:136-348``) backed by the HIP engine.

Same constructor, same ``__call__(grads, inputs, labels, pre_grads) -> grads`` (in-place modification), same
``create_graph`` attribute and the same six ``implementation`` strings.  The finite-difference variants
(``forward-differences``, ``forward-differences-legacy``, ``central-differences``) are implemented: the extra
forward/backward passes at theta +/- eps_n*v run through ``libfbengine.so`` with fp32 storage (perturbations are ~1e-6 per
weight, below bf16 resolution) with exact fp32 products (``bf16x6``: three bf16 pieces per operand; ``FB_F32_SPLIT=f16x2`` selects the
training loop's faster 22-bit arithmetic).  The caller's ``grads`` come from ITS autograd (other kernels, other summation order), so the forward-difference
quotient never subtracts them from an engine gradient -- that would divide uncorrelated rounding noise by eps_n: the base gradient at theta
is evaluated once more by the engine, in the arithmetic and order of the perturbed pass, and vhp = (g_engine(theta + eps_n v) -
g_engine(theta)) / eps_n is added to the caller's gradient (the training loop, where both passes are the engine's anyway, does the same).
The autograd-based variants need double backward and raise ``NotImplementedError``; ``complex-step`` is declared non-working by the
reference itself.

The model must be a ``fullbatchtraining_amd.models.ResNet`` on a HIP device.  Like the reference, BatchNorm running
statistics are updated again by each extra forward pass; parameters are left untouched (the reference perturbs
them in place and restores them -- the engine perturbs its own copy).
"""
import torch

from .engine import BN_MOMENTUM, Engine, stem_patches
from .lib import call

_IMPLEMENTATIONS = ("autograd-pen", "autograd", "central-differences", "complex-step", "forward-differences",
                    "forward-differences-legacy")


class GradRegularizer:
    """Modify given iterable of gradients outside of autograd."""

    def __init__(self, model, optimizer, loss_fn, norm=2, block_strength=0.1, acc_strength=0.0, eps=1e-2,
                 implementation="finite_diff", mixed_precision=False):
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.norm, self.block_strength, self.acc_strength, self.eps = norm, block_strength, acc_strength, eps
        self.mixed_precision = mixed_precision
        self.implementation = implementation
        self._engines = {}
        if self.block_strength == 0 and self.acc_strength == 0:
            self.forward, self.create_graph = self._pass, False
        elif implementation in ("autograd-pen", "autograd"):
            self.forward, self.create_graph = self._needs_double_backward, True
        elif implementation == "complex-step":
            self.forward, self.create_graph = self._needs_double_backward, False
        elif implementation in ("central-differences", "forward-differences", "forward-differences-legacy"):
            self.forward, self.create_graph = self._finite_differences, False
        else:
            raise ValueError(f"Invalid spec. given for regularizer implementation: {implementation}")
        if norm != 2:
            raise NotImplementedError("only the squared 2-norm penalty is implemented")

    def _pass(self, grads, inputs, labels, pre_grads):
        return grads

    def _needs_double_backward(self, grads, inputs, labels, pre_grads):
        raise NotImplementedError(f"grad_reg.implementation={self.implementation!r} relies on double backward / complex autograd, "
                                  "which the hand-scheduled engine does not provide; use a finite-difference implementation")

    def _engine(self, inputs):
        key = (inputs.shape[0], inputs.shape[-1], inputs.device)
        if key not in self._engines:
            sets = 2 if self.implementation == "central-differences" else 1
            # reference precision: the object is the compatibility surface, not the hot path -- exact fp32 products (bf16x6) unless FB_F32_SPLIT says otherwise
            self._engines[key] = Engine(self.model, inputs.shape[-1], inputs.shape[0], 1, compute_dtype=torch.float32,
                                        device=inputs.device, fd_sets=sets, f32_split="bf16x6")
            loss_fn = self.loss_fn      # the perturbed passes use the caller's loss (reference modules.py:228-230): CE / smoothing / incorrect-xent
            if not (isinstance(loss_fn, torch.nn.CrossEntropyLoss) or hasattr(loss_fn, "smoothing")):
                raise NotImplementedError(f"GradRegularizer: loss function {type(loss_fn).__name__} is not implemented by the head kernel")
            self._engines[key].label_smoothing = getattr(loss_fn, "smoothing", 0.0)
            self._engines[key].only_incorrect = getattr(loss_fn, "only_incorrect", False)
        return self._engines[key]

    @torch.no_grad()
    def _finite_differences(self, grads, inputs, labels, pre_grads):
        eng = self._engine(inputs)
        P = eng.plan.P
        eng.load_from_model(self.model)
        eng.g[0].copy_(eng.flatten([g.detach() for g in grads]))
        lr = self.optimizer.param_groups[0]["lr"]
        legacy = self.implementation == "forward-differences-legacy"
        central = self.implementation == "central-differences"
        s = 1.0 if legacy else float(self.block_strength)
        cf = lr / 4 * (self.block_strength if legacy else 1.0)
        patches = stem_patches(inputs.float(), eng.plan.stem, torch.float32)
        labels = labels.to(dtype=torch.long)
        # direction v = s*g + acc_strength*pre_grads (modules.py:217-221); the legacy variant disregards pre_grads (:243-245)
        vpre, vacc = None, 0.0
        if pre_grads is not None and not legacy:
            pre = eng.flatten([g.detach() for g in pre_grads]).to(eng.device)
            vpre, vacc = pre.data_ptr(), float(self.acc_strength)
        call("fb_mt_sqnorm", eng.g.data_ptr(), P, 1, P, s, vpre, vacc, eng.vnorm2.data_ptr(), eng.mt_ws.data_ptr())
        passes = [(0.5, 0), (-0.5, 1)] if central else [(1.0, 0)]
        if not central:          # the base gradient at theta in the engine's own arithmetic (statistics of this pass are not an extra BN update)
            if getattr(eng, "g_base", None) is None:
                eng.g_base = torch.zeros_like(eng.g)
            eng.prep_weights(eng.theta, 1)
            eng.group_gradient(patches, labels, 1, eng.g_base, 1, eng.theta, 0)
        for sign, slot in passes:
            call("fb_mt_fd_perturb", eng.theta.data_ptr(), eng.g.data_ptr(), P, 1, P, s, float(self.eps), sign, eng.vnorm2.data_ptr(),
                 eng.eps_n.data_ptr(), vpre, vacc, eng.theta_k.data_ptr())
            eng.prep_weights(eng.theta_k, 1, per_chunk=True)
            eng.group_gradient(patches, labels, 1, eng.g_fd[slot], 2, eng.theta_k, 1 + slot)
        eng.avg.zero_()
        gb = eng.g_fd[1] if central else eng.g_base
        call("fb_mt_fd_combine_accumulate", eng.avg.data_ptr(), eng.g.data_ptr(), eng.g_fd[0].data_ptr(), gb.data_ptr(), P, 1, P,
             eng.eps_n.data_ptr(), cf, 0)
        # running statistics: one more EMA update per extra forward pass, in pass order
        n_extra = len(passes)
        call("fb_bn_running_update", eng.running_mean.data_ptr(), eng.running_var.data_ptr(), eng.mean_tab[1].data_ptr(),
             eng.var_tab[1].data_ptr(), n_extra, eng.G * eng.plan.ch_total, eng.unbias.data_ptr(), 1, eng.plan.ch_total, BN_MOMENTUM)
        eng.num_batches_tracked += n_extra
        eng.store_buffers_to_model(self.model)
        for g, new in zip(grads, eng.unflatten_list(eng.avg)):
            g.copy_(new.to(g.dtype))
        return grads

    def __call__(self, *args):
        return self.forward(*args)
