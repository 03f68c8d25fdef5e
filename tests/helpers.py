"""Shared test helpers: synthetic data (same generator recipe as tests/golden/make_golden.py) and summaries."""
import numpy as np
import torch

SAMPLE_STRIDE = 997
NOISE_SEED = 4242      # tests/golden/make_golden.py seeds the default generator with it right before train() of the noise scenarios


def make_data(n, pixels=32, classes=10, seed=1234):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, pixels, pixels, generator=gen)
    y = torch.randint(0, classes, (n,), generator=gen)
    return x, y


def summarise(tensors):
    flat = torch.cat([t.detach().reshape(-1).double() for t in tensors])
    per = np.array([[float(t.double().sum()), float(t.double().pow(2).sum()), float(t.abs().max())] for t in tensors])
    return per, flat[::SAMPLE_STRIDE].numpy()


def hyp_from_cfg(cfg):
    """Flatten the cfg.hyp keys the oracle's step consumes."""
    o = cfg.hyp.optim
    return dict(lr=o.lr, weight_decay=o.weight_decay, momentum=o.momentum, nesterov=o.nesterov, dampening=o.dampening,
                block_strength=cfg.hyp.grad_reg.block_strength, eps=cfg.hyp.grad_reg.eps,
                implementation=cfg.hyp.grad_reg.implementation, grad_clip=cfg.hyp.grad_clip,
                acc_strength=cfg.hyp.grad_reg.acc_strength, optim_modification=dict(cfg.hyp.optim_modification),
                grad_clip_norm=cfg.hyp.grad_clip_norm, norm_bias=dict(cfg.hyp.norm_bias), evaluate_ema=cfg.hyp.evaluate_ema,
                eval_ema_momentum=cfg.hyp.eval_ema_momentum, test_time_flips=cfg.hyp.test_time_flips,
                only_linear_layers_weight_decay=cfg.hyp.only_linear_layers_weight_decay, label_smoothing=cfg.hyp.label_smoothing,
                loss_modification=cfg.hyp.loss_modification, grad_noise=dict(cfg.hyp.grad_noise), block=cfg.data.batch_size,
                batch_clip=cfg.hyp.batch_clip)


def torch_bf16_chunk_grads(model, x, y, chunk, device="cpu"):
    """The INDEPENDENT bf16 yardstick: torch's own ``autocast(bfloat16)`` evaluation of the same model on the same chunks -- the reference's
    ``_compute_batched_gradient`` under ``impl.mixed_precision`` (fullbatch/training/training.py:76-83: forward + CrossEntropyLoss inside
    autocast, ``torch.autograd.grad`` outside).  Plain torch: it touches neither ``libfbengine`` nor ``oracle/``.  Returns
    [(gradient list in ``model.parameters()`` order (fp32, host), loss)] per chunk."""
    import copy

    dev = torch.device(device)
    m = copy.deepcopy(model).to(dev).float().train()
    params = list(m.parameters())
    out = []
    for k in range(x.shape[0] // chunk):
        xb, yb = x[k * chunk:(k + 1) * chunk].to(dev), y[k * chunk:(k + 1) * chunk].to(dev)
        with torch.autocast(device_type=dev.type, dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(m(xb), yb)
        grads = torch.autograd.grad(loss, params)
        out.append(([g.detach().float().cpu() for g in grads], float(loss.detach())))
    return out


def flat64(tensors):
    return torch.cat([t.detach().reshape(-1).double().cpu() for t in tensors])


def err_cos(a, truth):
    """(relative L2 distance, cosine) of flat float64 vectors."""
    return float((a - truth).norm() / truth.norm()), float((a * truth).sum() / (a.norm() * truth.norm()))


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def shuffling_loaders(x, y, batch):
    """The loaders tests/golden/make_golden.py hands to the reference in the "shuffle" scenarios: RandomSampler train loader and a
    validation loader that share one generator seeded with 0."""
    ds = torch.utils.data.TensorDataset(x, y)
    own = torch.Generator().manual_seed(0)
    train = torch.utils.data.DataLoader(ds, batch_size=min(batch, len(ds)), shuffle=True, drop_last=True, generator=own)
    valid = torch.utils.data.DataLoader(ds, batch_size=min(batch, len(ds)), shuffle=False, drop_last=False, generator=own)
    return train, valid


def loader_pass_indices(loader):
    """Sample indices of one pass over ``loader`` in its order, consuming its generator exactly like ``for batch in loader`` does
    (torch DataLoader: the iterator draws a base seed first, then the sampler draws its permutation)."""
    torch.empty((), dtype=torch.int64).random_(generator=loader.generator)
    return torch.tensor([i for batch in loader.batch_sampler for i in batch], dtype=torch.long)


# ----------------------------------------------------------------------------------------------------------------------------------
# A dict-backed stand-in for the part of the third-party `lmdb` API the reference's LMDB dataset code uses (fullbatch/data/
# lmdb_datasets.py: open / begin / put / get / commit / cursor first-key-value-set_key-next).  tests/golden/make_golden.py --r2 hands it
# to the REFERENCE's writer and reader (the `lmdb` package is not installed in the build image); the tests rebuild a store from the
# committed key/value fixture.  Keys iterate in byte order, like LMDB's B+tree.
class DictLMDB:
    _stores = {}

    class _Cursor:
        def __init__(self, store):
            self.store, self.keys, self.pos = store, sorted(store), 0

        def first(self):
            self.pos = 0
            return bool(self.keys)

        def key(self):
            return self.keys[self.pos] if self.pos < len(self.keys) else b""

        def value(self):
            return self.store[self.keys[self.pos]] if self.pos < len(self.keys) else b""

        def set_key(self, key):
            import bisect
            i = bisect.bisect_left(self.keys, key)
            if i < len(self.keys) and self.keys[i] == key:
                self.pos = i
                return True
            return False

        def next(self):
            self.pos += 1
            return self.pos < len(self.keys)

    class _Txn:
        def __init__(self, store):
            self.store = store

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

        def put(self, key, value):
            self.store[bytes(key)] = bytes(value)
            return True

        def get(self, key, default=None):
            return self.store.get(bytes(key), default)

        def commit(self):
            pass

        def cursor(self):
            return DictLMDB._Cursor(self.store)

    def __init__(self, store=None):
        self.store = {} if store is None else store

    def begin(self, write=False, **kwargs):
        return DictLMDB._Txn(self.store)

    @classmethod
    def open(cls, path, **kwargs):
        return cls(cls._stores.setdefault(str(path), {}))


# ----------------------------------------------------------------------------------------------------------------------------------
PG_TIMEOUT_S = 120       # every process group of the tests: a collective that never completes fails after two minutes, not thirty


def pg_timeout():
    import datetime
    return datetime.timedelta(seconds=PG_TIMEOUT_S)


def spawn_bounded(fn, args, nprocs, timeout=150.0):
    """``torch.multiprocessing.spawn(fn, args, nprocs)`` with a wall-clock cap: the ranks are fresh `spawn` children (never a re-exec of a
    process that touched the GPU); the parent polls them, and when the cap passes it KILLS every child that is still alive and fails the
    calling test.  An exception in a rank surfaces as torch's ProcessRaisedException, as with ``join=True``."""
    import time

    import torch.multiprocessing as mp

    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    deadline = time.monotonic() + timeout
    try:
        while not ctx.join(timeout=2.0):
            if time.monotonic() > deadline:
                raise TimeoutError(f"{getattr(fn, '__name__', fn)} with {nprocs} rank(s) still running after {timeout:.0f} s: ranks killed")
    finally:
        for proc in ctx.processes:
            if proc.is_alive():
                proc.kill()
        for proc in ctx.processes:
            proc.join(10)


# ----------------------------------------------------------------------------------------------------------------------------------
# Where the float64 oracle's tensors live in the `-m gpu` tests.  oracle/fb_oracle.py is a restatement in plain torch ops and does not
# care: on the host cores of a GPU box one float64 chunk gradient of ResNet-18 (128 images, 32 px) takes 29 s, with its tensors on the
# device 0.5 s, and the two agree to 2e-14 (tools/scratch/oracle_on_gpu.py; asserted by tests/test_gpu_engine.py::
# test_oracle_on_the_device_equals_the_oracle_on_the_host).  The convolutions then run in torch's own GPU kernels -- independent of
# libfbengine either way.  FB_ORACLE_DEVICE=cpu puts it back on the host; the `-m "not gpu"` tests always run it there.
def oracle_device():
    import os
    want = os.environ.get("FB_ORACLE_DEVICE", "cuda")
    return torch.device(want if (want == "cpu" or torch.cuda.is_available()) else "cpu")


def oracle_state(model, dtype=torch.float64, device=None):
    """(params, buffers) of the oracle from a parameter container, in ``dtype`` on the oracle's device."""
    from oracle import fb_oracle as orc
    device = oracle_device() if device is None else device
    state = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.detach().clone()).to(device) for k, v in model.state_dict().items()}
    return orc.split_state(state)


def to_oracle(*tensors, dtype=torch.float64):
    dev = oracle_device()
    out = tuple(t.to(dev, dtype) if t.is_floating_point() else t.to(dev) for t in tensors)
    return out[0] if len(out) == 1 else out
