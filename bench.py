#!/usr/bin/env python
"""Benchmark of the full-batch GD hot path: ResNet-18 / CIFAR-10-shaped synthetic data, one optimizer step = gradient of
ALL chunks (390 x 128 = 49 920 images, the reference's drop_last behaviour) + clip + Nesterov-SGD update.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line (rank 0, < 4 KB; tables, notes and sources go to gpurun_out/bench_detail.json): whole-job images/s (value), steps/s,
ms/step, plus
  roofline     : dominant MFMA kernel class -- algorithmic FLOP per launch / mean launch duration measured with HIP events on
                 the launch stream inside the timed region (libfbengine's fb_profile_*), against the dense bf16 MFMA peak
  cpu_baseline : the CPU oracle (restatement of the reference path, oracle/fb_oracle.py) timed on this box's host cores on a
                 bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3     # f32-input MFMA
PEAK_HBM_GBS = 8000.0       # HBM3E (spec; 6290 achievable, MI355X_MICROARCH.md)
CHUNK = 128
N_IMAGES = 50_000


def conv_flops(plan, n_img):
    """Algorithmic (direct-convolution, real channel counts) FLOP of the conv launches of one fwd+bwd over n_img images,
    per kernel class.  Matches SURVEY/BASELINE: 1.1108 GFLOP/img fwd, dgrad for all convs but the stem, wgrad for all."""
    fwd = dgrad = wgrad = 0
    for L in plan.layers:
        macs = n_img * L.hout * L.wout * L.cout * L.taps * L.cin_real
        fwd += 2 * macs
        wgrad += 2 * macs
        if L is not plan.stem:
            dgrad += 2 * macs
    return {"igemm_fwd": fwd, "igemm_dgrad": dgrad, "wgrad": wgrad}


def cpu_baseline(budget_s=24.0):
    """The CPU restatement of the reference path (oracle/fb_oracle.py, pinned to reference runs in tests/) timed on this box's host
    cores: fp32, chunks of 128 images, gradient through torch autograd (the way the reference computes it, training.py:76-83 -- the
    explicit-backward form is slower on CPU).  Thread counts 8 / 16 / 32 are probed on one chunk each, the best one is then
    timed on a bounded sample with the regulariser off and on.  SURVEY section 6 measured the REAL reference in the build container at
    ~122 images/s (8 threads, grad_reg off) / ~50 images/s (on)."""
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.models import construct_model
    from oracle import fb_oracle as orc

    torch.manual_seed(1)
    model = construct_model(compose([]).model, 3, 10)
    params, buffers = orc.split_state({k: v.clone() for k, v in model.state_dict().items()})
    spec = orc.Spec(18)
    gen = torch.Generator().manual_seed(1234)
    n_max = 24
    x = torch.randn(n_max * CHUNK, 3, 32, 32, generator=gen)
    y = torch.randint(0, 10, (n_max * CHUNK,), generator=gen)
    grad = orc.chunk_gradient_autograd

    def one(k, reg):
        xs, ys = x[k * CHUNK:(k + 1) * CHUNK], y[k * CHUNK:(k + 1) * CHUNK]
        g, _, _ = grad(spec, params, buffers, xs, ys)
        if reg:
            orc.gradreg(spec, params, buffers, g, xs, ys, 0.1, 0.5, 1e-2, "forward-differences", chunk_gradient=grad)

    avail = os.cpu_count() or 8
    try:                                                # physical cores of the host (BASELINE.md section 4 asks for the count beside the number)
        import psutil
        physical = psutil.cpu_count(logical=False)
    except Exception:
        physical = None
    probe = {}
    saved = torch.get_num_threads()
    for t in sorted({min(c, avail) for c in (8, 16, 32)}):          # (more threads only lose: 64: 64 img/s, 256: 1 img/s on the r3 box -- minutes of probing)
        torch.set_num_threads(t)
        one(0, False)                                   # warm-up at this thread count
        t0 = time.perf_counter()
        one(1, False)
        probe[t] = CHUNK / (time.perf_counter() - t0)
    best = max(probe, key=probe.get)
    torch.set_num_threads(best)
    out = {}
    for reg in (False, True):
        t0, k = time.perf_counter(), 0
        while k < n_max and (k < 2 or time.perf_counter() - t0 < budget_s / 2):
            one(k, reg)
            k += 1
        dt = time.perf_counter() - t0
        out[reg] = (k, dt)
    torch.set_num_threads(saved)
    (k0, d0), (k1, d1) = out[False], out[True]
    return {"value": round(k0 * CHUNK / d0, 2), "unit": "images/s", "cores": best, "kind": "port", "host_cpus": physical, "host_threads": avail,
            "sample_short": f"{k0} chunks x {CHUNK} images fwd+bwd fp32 oracle (torch autograd), {d0:.1f} s",
            "sample": f"{k0} chunks of {CHUNK} images, fwd+bwd via torch autograd of the fp32 oracle forward, grad_reg off, {d0:.1f} s "
                      f"(one full step = 390 chunks ~ {390 * d0 / k0:.0f} s); thread sweep on one chunk each: "
                      + ", ".join(f"{t}: {v:.0f} img/s" for t, v in probe.items()),
            "grad_reg_on": {"value": round(k1 * CHUNK / d1, 2), "unit": "images/s",
                            "sample": f"{k1} chunks, forward-differences block_strength 0.5, {d1:.1f} s (one step ~ {390 * d1 / k1:.0f} s)"},
            "reference_in_build_container": "SURVEY.md section 6: the real reference, 8 threads: ~122 images/s (off), ~50 images/s (on)"}


class PowerSampler:
    """Board power and shader clock of this process's GPU over the timed region, read by a HOST thread from the amdgpu hwmon files (power1_input in
    microwatts, freq1_input in Hz, power1_cap) every 0.2 s -- no device call, nothing queued.  Why it is in the line: the headline step runs AT the
    board's power cap (round 5: 1364-1384 W of 1400 W, shader clock 2.12-2.20 GHz instead of 2.4), which is why kernels that are faster alone on the
    device do not shorten the step.  Unreadable files (another driver layout, no permission): ``result()`` is None."""

    def __init__(self, device_index):
        import glob
        import threading
        self.dir, self.samples, self._stop, self._thread = None, [], threading.Event(), None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            found = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            if found and os.path.exists(os.path.join(found[0], "power1_input")):
                self.dir = found[0]
        except Exception:
            self.dir = None
        self._threading = threading

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as handle:
                return float(handle.read().strip())
        except Exception:
            return None

    def _run(self):
        while not self._stop.wait(0.2):
            self.samples.append((self._read("power1_input"), self._read("freq1_input")))

    def start(self):
        if self.dir is not None:
            self._thread = self._threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def result(self):
        if self._thread is None:
            return None
        self._stop.set()
        self._thread.join()
        watts = [p / 1e6 for p, _ in self.samples if p]
        mhz = [f / 1e6 for _, f in self.samples if f]
        cap = self._read("power1_cap")
        if not watts:
            return None
        return {"avg_w": round(sum(watts) / len(watts)), "max_w": round(max(watts)), "cap_w": round(cap / 1e6) if cap else None,
                "sclk_mhz": round(sum(mhz) / len(mhz)) if mhz else None, "sclk_nominal_mhz": 2400, "samples": len(watts), "source": "amdgpu hwmon, host thread"}


def _timed_steps(trainer, n_steps, warm):
    for _ in range(warm):
        trainer.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        trainer.step()
    trainer.flush_stats()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_steps


def _side_trainer(args, device, X, Y, overrides, name, env=None):
    """A second trainer on the same resident synthetic dataset (another configuration of BASELINE.json, timed in the same run)."""
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.models import construct_model
    from fullbatchtraining_amd.training import FullBatchTrainer

    saved = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        cfg = compose(overrides + [f"impl.engine.chunk_group={args.chunk_group}"] + ([] if any("device_augment=True" in o for o in overrides) else ["data.augmentations_train="]),
                      original_cwd=os.path.join(ROOT, "gpurun_out"), name=name)
        torch.manual_seed(1)
        model = construct_model(cfg.model, 3, 10)
        setup = dict(device=device, dtype=torch.float, memory_format=torch.contiguous_format)
        return FullBatchTrainer(model, (X, Y), None, setup, cfg)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _mean_gradient(trainer, n_chunks, block_strength=0.0):
    """Mean (regularised) gradient of the first ``n_chunks`` chunks at the trainer's current parameters; BN running statistics restored."""
    eng = trainer.engine
    rm, rv, nbt = eng.running_mean.clone(), eng.running_var.clone(), eng.num_batches_tracked
    rows = n_chunks * trainer.chunk_pad
    eng.full_gradient(trainer.patches[:rows], trainer.labels[:rows], 0.1, block_strength=block_strength)
    eng.running_mean.copy_(rm), eng.running_var.copy_(rv)
    eng.num_batches_tracked = nbt
    return eng.avg.double().clone()


def _rel(a, b, chunks=None):
    """Relative L2 distance and cosine; with ``chunks``: the bound the bf16 path is held to on a mean over that many chunks.  A single chunk
    gradient moves by ~0.2-0.3 under bf16 storage (ReLU-mask flips of 2^-9-rounded pre-activations: noise, not bias), the mean of K chunks
    by that / sqrt(K) (tests/test_gpu_bf16_parity.py asserts the decay): 0.5 / sqrt(K) is the line a biased kernel would cross."""
    out = {"rel_l2": round(float((a - b).norm() / b.norm()), 5), "cosine": round(float((a * b).sum() / (a.norm() * b.norm())), 6)}
    if chunks is not None:
        out["bound"] = round(0.5 / chunks ** 0.5, 5)
        out["within_bound"] = bool(out["rel_l2"] <= out["bound"])
    return out


def side_configs(args, device, X, Y, main_trainer):
    """Timed in the same run as the headline line (N = 1), on the same resident dataset:
      configs.k400            all 50 000 images as 400 chunks of 125 (data.batch_size = 125: the number BASELINE's metric is quoted on; stored padded to 128 images per chunk)
      configs.k250            all 50 000 images as 250 chunks of 200 (data.batch_size = 200: the smallest-padding form of "bs = 50 000" -- none at all)
      configs.gradreg         BASELINE config 3 (GradRegularizer block_strength 0.5, forward differences) at REFERENCE precision: fp32 storage,
                              every convolution product exact to 2^-23 (three bf16 pieces per operand, six MFMAs: "bf16x6")
      configs.gradreg_f16x2   the same step in the arithmetic the engine uses by default for the regulariser: operands carried as two scaled
                              fp16 pieces (22 significand bits), three MFMAs per product
      parity                  bf16 (the arithmetic behind `value`) vs fp32 on the mean gradient of ALL chunks of the step, and f16x2 vs
                              bf16x6 on the regularised mean gradient of 16 chunks"""
    import gc

    out = {"configs": {}, "parity": {}}
    flop_img = 3328997376
    # ---- all 50 000 images: 400 chunks of 125 ----
    tr = _side_trainer(args, device, X, Y, ["hyp=fb1", "hyp.warmup=0", "hyp.steps=12", "impl.mixed_precision=True", "data.batch_size=125", "hyp.sub_batch=125"], "bench_k400")
    dt = _timed_steps(tr, 5, 2)                     # (the literal "bs = 50 000" of the metric: timed like a headline, not like a footnote)
    out["configs"]["k400"] = {"workload": f"ResNet-18 CIFAR-10 full-batch GD step over ALL {tr.datapoints} images: {tr.n_chunks} chunks x {tr.chunk} (stored padded to "
                                          f"{tr.chunk_pad} images per chunk), bf16, grad_reg off", "ms_per_step": round(1000 * dt, 2), "value": round(tr.datapoints / dt, 1),
                              "unit": "images/s", "steps": 5, "warmup": 2, "dtype": "bf16", "step_mfma_frac": round(flop_img * tr.datapoints / dt / (PEAK_BF16_TFLOPS * 1e12), 4),
                              "train_loss_last": tr.stats["train_loss"][-1]}
    del tr
    gc.collect(), torch.cuda.empty_cache()
    # ---- all 50 000 images WITHOUT padding: 250 chunks of 200 (a chunk size whose pixels fill whole 128-pixel statistics blocks on every map: 200 x 16 = 25 blocks at 4x4) ----
    tr = _side_trainer(args, device, X, Y, ["hyp=fb1", "hyp.warmup=0", "hyp.steps=12", "impl.mixed_precision=True", "data.batch_size=200", "hyp.sub_batch=200"], "bench_k250")
    dt = _timed_steps(tr, 5, 2)
    out["configs"]["k250"] = {"workload": f"ResNet-18 CIFAR-10 full-batch GD step over ALL {tr.datapoints} images: {tr.n_chunks} chunks x {tr.chunk} (data.batch_size = 200: no padding, "
                                          f"stored chunk {tr.chunk_pad}), bf16, grad_reg off", "ms_per_step": round(1000 * dt, 2), "value": round(tr.datapoints / dt, 1),
                              "unit": "images/s", "steps": 5, "warmup": 2, "dtype": "bf16", "step_mfma_frac": round(flop_img * tr.datapoints / dt / (PEAK_BF16_TFLOPS * 1e12), 4),
                              "chunk_group": tr.engine.G, "train_loss_last": tr.stats["train_loss"][-1]}
    assert tr.chunk_pad == tr.chunk                         # (no padding images)
    del tr
    gc.collect(), torch.cuda.empty_cache()
    # ---- the headline workload with the stem's patch gather INSIDE the step: on-device RandomCrop(32, 4) + RandomHorizontalFlip of the resident
    # un-augmented images every step (config/data/CIFAR10.yaml:11-13; the static headline gathers the patches once, outside the timed region) ----
    tr = _side_trainer(args, device, X, Y, ["hyp=fb1", "hyp.warmup=0", "hyp.steps=8", "impl.mixed_precision=True", "impl.engine.device_augment=True"], "bench_augmented")
    dt = _timed_steps(tr, 3, 1)
    out["configs"]["augmented"] = {"workload": f"the headline step with the dataset augmented on the device every step (RandomCrop + flip inside fb_stem_patches: the "
                                               f"3.3 GB patch gather is part of the step), {tr.n_chunks} chunks x {tr.chunk}, bf16", "ms_per_step": round(1000 * dt, 2),
                                   "value": round(tr.datapoints / dt, 1), "unit": "images/s", "steps": 3, "warmup": 1, "dtype": "bf16", "train_loss_last": tr.stats["train_loss"][-1]}
    del tr
    gc.collect(), torch.cuda.empty_cache()
    # ---- config 3 in both arithmetic modes ----
    e16 = main_trainer.engine
    grads = {}
    for mode, label, dtype_label in (("bf16x6", "gradreg", "f32"), ("f16x2", "gradreg_f16x2", "f32-22bit")):
        tr = _side_trainer(args, device, X, Y, ["hyp=gradreg", "hyp.warmup=0", "hyp.steps=8", "hyp.grad_reg.block_strength=0.5", "impl.mixed_precision=False"],
                           "bench_" + label, env={"FB_F32_SPLIT": mode})
        assert tr.engine.f32_split == mode
        dt = _timed_steps(tr, 3, 1)
        flop = 2 * flop_img * tr.datapoints
        per_product = 6 if mode == "bf16x6" else 3
        out["configs"][label] = {
            "workload": f"ResNet-18 CIFAR-10 full-batch GD step + GradRegularizer block_strength=0.5 (forward differences, eps 1e-2), {tr.n_chunks} chunks x {tr.chunk}, "
                        "fp32 storage; " + ("every fp32 operand as three bf16 pieces, six MFMAs per product: products exact to 2^-23 (the reference runs these passes in fp32)"
                                            if mode == "bf16x6" else "every fp32 operand as two scaled fp16 pieces (22 significand bits, one power-of-two scale per chunk and "
                                            "tensor), three MFMAs per product -- narrower than the reference's fp32: see parity.f16x2_vs_bf16x6"),
            "arithmetic": mode, "ms_per_step": round(1000 * dt, 1), "value": round(tr.datapoints / dt, 1), "unit": "images/s", "steps": 3, "warmup": 1, "dtype": dtype_label,
            "train_loss_last": tr.stats["train_loss"][-1],
            "roofline": {"bound": "mfma", "unit": "TFLOP/s of 16-bit MFMA work", "mfma_per_product": per_product, "achieved": round(per_product * flop / dt / 1e12, 1),
                         "peak": PEAK_BF16_TFLOPS, "frac": round(per_product * flop / dt / 1e12 / PEAK_BF16_TFLOPS, 4), "algorithmic_tflops": round(flop / dt / 1e12, 1)}}
        eng = tr.engine
        eng.theta.copy_(e16.theta), eng.running_mean.copy_(e16.running_mean), eng.running_var.copy_(e16.running_var)
        grads[mode] = _mean_gradient(tr, 16, block_strength=0.5)
        # (the regularised passes too: production schedule vs one stream, bit for bit)
        saved_stream, saved_lists = eng.wstream, eng.cmdlists
        eng.wstream, eng.cmdlists = None, {}
        try:
            g_one = _mean_gradient(tr, 16, block_strength=0.5)
        finally:
            eng.wstream, eng.cmdlists = saved_stream, saved_lists
        out["parity"].setdefault("two_streams_vs_one_regularised", {})[mode] = dict(
            bit_identical=bool(torch.equal(grads[mode], g_one)), chunks=16, max_abs_diff=float((grads[mode] - g_one).abs().max()))
        if mode == "bf16x6":
            # the fp32 (exact-product) gradient of ALL chunks at the benchmark's parameters, against the bf16 path that produced `value`
            t0 = time.perf_counter()
            g32 = _mean_gradient(tr, tr.n_chunks)
            torch.cuda.synchronize()
            t32 = time.perf_counter() - t0
            g16 = _mean_gradient(main_trainer, main_trainer.n_chunks)
            out["parity"]["bf16_vs_f32"] = dict(_rel(g16, g32, tr.n_chunks), chunks=tr.n_chunks, f32_gradient_ms=round(1000 * t32, 1),
                                                note="MEAN gradient of all chunks of the step (what the update consumes), bf16 engine vs fp32 engine (bf16x6) at "
                                                     "the benchmark's current parameters; single chunks differ by ~0.2 (ReLU-mask flips of 2^-9-rounded pre-activations: "
                                                     "noise, not bias -- tests/test_gpu_bf16_parity.py asserts the 1/sqrt(K) decay)")
            if not out["parity"]["bf16_vs_f32"]["within_bound"]:         # a throughput number from a path that left its parity bound is not a result
                raise RuntimeError(f"bf16 mean gradient outside its bound: {out['parity']['bf16_vs_f32']}")
            # the production schedule (second stream, replayed command lists) against the same launches in ONE stream: bit-identical or broken
            # (the check that would have shown round 3's store-data hazard, csrc/common.h store_b128_guard, in the driver's own line)
            me = main_trainer.engine
            saved_stream, saved_lists = me.wstream, me.cmdlists
            me.wstream, me.cmdlists = None, {}
            try:
                g_one = _mean_gradient(main_trainer, main_trainer.n_chunks)
            finally:
                me.wstream, me.cmdlists = saved_stream, saved_lists
            again = [_mean_gradient(main_trainer, main_trainer.n_chunks) for _ in range(2)]
            out["parity"]["two_streams_vs_one"] = dict(bit_identical=bool(torch.equal(g16, g_one) and all(torch.equal(g, g_one) for g in again)),
                                                       evaluations=3, chunks=main_trainer.n_chunks,
                                                       max_abs_diff=float(max((g - g_one).abs().max() for g in [g16] + again)))
            g16_16, g32_16 = _mean_gradient(main_trainer, 16), _mean_gradient(tr, 16)
            out["parity"]["bf16_vs_f32_16_chunks"] = dict(_rel(g16_16, g32_16, 16), chunks=16)
        del tr, eng
        gc.collect(), torch.cuda.empty_cache()
    out["parity"]["f16x2_vs_bf16x6"] = dict(_rel(grads["f16x2"], grads["bf16x6"]), chunks=16,
                                            note="regularised (forward differences, block_strength 0.5) mean gradient of 16 chunks of 128 at 32 px, same parameters: "
                                                 "22-bit operands vs exact fp32 products")
    return out


def r152_configs(args, device):
    """BASELINE config 5's model on ONE GPU, timed in the driver's own run (the main trainer has been released: these steps take 210 GB of the device):
      configs.r152                 ResNet-152, 'standard' stem, 224 x 224 synthetic inputs, 16 chunks x 128 = 2048 images per step, bf16, grad_reg off
      configs.r152_gradreg         the configuration AS BASELINE STATES IT (with the GradRegularizer, block_strength 0.5, forward differences): 8 chunks x 128 = 1024 images
                                   per step, fp32 storage, six bf16 MFMAs per product (reference precision)
      configs.r152_gradreg_f16x2   the same step in the opt-in 22-bit arithmetic (three fp16 MFMAs per product)"""
    import gc

    gen = torch.Generator().manual_seed(4321)
    X = torch.randn(2048, 3, 224, 224, generator=gen)
    Y = torch.randint(0, 10, (2048,), generator=gen)
    model_over = ["model=resnet152", "model.stem=standard", "data.pixels=224"]
    out = {}
    runs = (("r152", 2048, ["hyp=fb1", "hyp.warmup=0", "hyp.steps=12", "impl.mixed_precision=True"], None, 4, 1, "bf16", 1),
            ("r152_gradreg", 1024, ["hyp=gradreg", "hyp.warmup=0", "hyp.steps=12", "hyp.grad_reg.block_strength=0.5", "impl.mixed_precision=False"], "bf16x6", 3, 1, "f32", 6),
            ("r152_gradreg_f16x2", 1024, ["hyp=gradreg", "hyp.warmup=0", "hyp.steps=12", "hyp.grad_reg.block_strength=0.5", "impl.mixed_precision=False"], "f16x2", 3, 1, "f32-22bit", 3))
    for label, n_img, over, mode, steps, warm, dtype_label, per_product in runs:
        tr = _side_trainer(args, device, X[:n_img], Y[:n_img], over + model_over, "bench_" + label, env={"FB_F32_SPLIT": mode} if mode else None)
        eng = tr.engine
        if tr.patches is not None:
            eng.choose_schedule(tr.patches, tr.labels, tr.shard.count)        # (the stream choice of a wide Bottleneck net: outside the timed steps)
        dt = _timed_steps(tr, steps, warm)
        flop_img = sum(conv_flops(eng.plan, 1).values())
        passes = 2 if mode else 1
        out[label] = {
            "workload": f"ResNet-152 ('standard' stem, 224 x 224 synthetic inputs, 10 classes) full-batch GD step, {tr.n_chunks} chunks x {tr.chunk} = {tr.datapoints} images/step"
                        + (", GradRegularizer block_strength=0.5 (forward differences), fp32 storage, " + ("six bf16" if mode == "bf16x6" else "three fp16") + " MFMAs per product" if mode else ", bf16, grad_reg off"),
            "ms_per_step": round(1000 * dt, 1), "value": round(tr.datapoints / dt, 1), "unit": "images/s", "steps": steps, "warmup": warm, "dtype": dtype_label,
            "chunk_group": eng.G, "streams": 1 if eng.wstream is None else 2, "train_loss_last": tr.stats["train_loss"][-1],
            "roofline": {"bound": "mfma", "unit": "TFLOP/s of 16-bit MFMA work", "mfma_per_product": per_product,
                         "achieved": round(per_product * passes * flop_img * tr.datapoints / dt / 1e12, 1), "peak": PEAK_BF16_TFLOPS,
                         "frac": round(per_product * passes * flop_img * tr.datapoints / dt / 1e12 / PEAK_BF16_TFLOPS, 4),
                         "algorithmic_tflops": round(passes * flop_img * tr.datapoints / dt / 1e12, 1)}}
        if mode:
            out[label]["arithmetic"] = mode
        del tr, eng
        gc.collect(), torch.cuda.empty_cache()
    return out


def _launch_work(cls, w, stem):
    """Algorithmic FLOP and bytes of one recorded launch from its shape words (include/fb_engine.h: fb_profile_read_launches)."""
    if cls in ("igemm_fwd", "igemm_dgrad", "wgrad"):
        n, hs, ws, cs, hd, wd, cd, r, stride, flags, kernel = w
        eb = 2 if ((flags >> 4) & 1 if cls != "wgrad" else (flags >> 16) & 1) else 4
        if cls == "igemm_fwd":
            macs, cin = n * hd * wd * cd * r * r * cs, cs
            byts = n * hs * ws * cs * eb + n * hd * wd * cd * eb
        elif cls == "igemm_dgrad":                         # src = dY [Hs,Ws,Cs] -> dst = dX [Hd,Wd,Cd]
            macs, cin = n * hs * ws * cs * r * r * cd, cd
            addend = flags & 3
            byts = n * hs * ws * cs * eb + n * hd * wd * cd * eb * (2 if addend == 1 else 1) + (n * hd * wd * cd * eb // 4 if addend == 2 else 0)
        else:                                              # x [Hs,Ws,Cs], dy [Hd,Wd,Cd]
            macs, cin = n * hd * wd * cd * r * r * cs, cs
            byts = n * hs * ws * cs * eb + n * hd * wd * cd * eb
        if r == 1 and cin == stem.cin_pad and (hd == stem.hout if cls != "igemm_dgrad" else False):
            macs = macs * stem.cin_real // stem.cin_pad    # the pre-gathered stem patches are zero-padded 27 -> 32 (147 -> 160): not algorithmic work
        shape = f"{cs}->{cd} k{r} s{stride} {hs}x{ws}->{hd}x{wd}" + (" +add" if cls == "igemm_dgrad" and flags & 3 else "")
        return 2.0 * macs, float(byts), shape, kernel
    px128, c, _, dtype, res, mask, dy_out, pooled = w[:8]
    eb = 2 if dtype == 1 else 4
    elems = px128 * 128 * c
    # (backward passes: `res` = 1 marks the two-BatchNorm form -- one more tensor read by the reduction, one more read + one more written by the apply step)
    passes = {"bn_apply": 2 + res + 0.25 * pooled, "bn_bwd_reduce": 2 + res, "bn_bwd_apply": 3 + dy_out + 2 * res, "bn_bwd_fused": 3 + dy_out}[cls]
    return 0.0, elems * eb * passes + (elems * eb / 16 if mask else 0), f"C{c} {px128 * 128} px", 0


def _kernel_table(launches, stem, n_steps):
    """Per (kernel, class, shape): launches, ms per step, algorithmic TFLOP/s and GB/s, fractions of both ceilings."""
    from fullbatchtraining_amd import lib
    rows = {}
    for cls, words, ms in launches:
        flop, byts, shape, kernel = _launch_work(cls, words, stem)
        name = lib.PROF_KERNELS.get(kernel, "?") if flop else {"bn_apply": "bn_apply_span_kernel", "bn_bwd_reduce": "bn_bwd_reduce_kernel", "bn_bwd_apply": "bn_bwd_apply_span_kernel",
                                                                   "bn_bwd_fused": "bn_bwd_fused_kernel"}[cls]
        r = rows.setdefault((name, cls, shape), [0, 0.0, 0.0, 0.0])
        r[0] += 1
        r[1] += ms
        r[2] += flop
        r[3] += byts
    table = []
    for (name, cls, shape), (n, ms, flop, byts) in rows.items():
        tf, gbs = flop / (ms * 1e-3) / 1e12, byts / (ms * 1e-3) / 1e9
        table.append({"kernel": name, "class": cls, "shape": shape, "launches_per_step": round(n / n_steps, 2), "ms_per_step": round(ms / n_steps, 3),
                      "avg_launch_us": round(1000 * ms / n, 1), "tflops": round(tf, 1), "gbs_algorithmic": round(gbs, 1),
                      "frac_mfma": round(tf / PEAK_BF16_TFLOPS, 4), "frac_hbm": round(gbs / PEAK_HBM_GBS, 4),
                      "flop_per_launch": flop / n, "bytes_per_launch": byts / n})
    table.sort(key=lambda r: -r["ms_per_step"])
    return table


def _by_kernel(table):
    agg = {}
    for r in table:
        a = agg.setdefault(r["kernel"], {"ms_per_step": 0.0, "launches_per_step": 0.0, "flop": 0.0, "bytes": 0.0})
        a["ms_per_step"] += r["ms_per_step"]
        a["launches_per_step"] += r["launches_per_step"]
        a["flop"] += r["flop_per_launch"] * r["launches_per_step"]
        a["bytes"] += r["bytes_per_launch"] * r["launches_per_step"]
    for k, a in agg.items():
        a["tflops"] = round(a["flop"] / (a["ms_per_step"] * 1e-3) / 1e12, 1)
        a["gbs_algorithmic"] = round(a["bytes"] / (a["ms_per_step"] * 1e-3) / 1e9, 1)
        a["frac_mfma"], a["frac_hbm"] = round(a["tflops"] / PEAK_BF16_TFLOPS, 4), round(a["gbs_algorithmic"] / PEAK_HBM_GBS, 4)
        a["avg_launch_us"] = round(1000 * a["ms_per_step"] / a["launches_per_step"], 1)
        a["flop_per_launch"], a["bytes_per_launch"] = a.pop("flop") / a["launches_per_step"], a.pop("bytes") / a["launches_per_step"]
        a["ms_per_step"], a["launches_per_step"] = round(a["ms_per_step"], 3), round(a["launches_per_step"], 2)
    return agg


def roofline_objects(args, trainer, launches, launches_iso, sec_per_step, world, headline, passes):
    """`roofline` (dominant MFMA kernel), `roofline_hbm` (dominant HBM-bound kernel), `hbm` (PMC traffic of the whole step) and the
    per-kernel / per-shape tables behind them.  Kernel = the __global__ function that served the launches (not a class of them)."""
    eng = trainer.engine
    stem = eng.plan.stem
    in_region = _kernel_table(launches, stem, args.steps)
    iso = _kernel_table(launches_iso, stem, 1) if launches_iso else None
    k_in, k_iso = _by_kernel(in_region), (_by_kernel(iso) if iso else None)
    basis = k_iso or k_in                                   # additive durations (one stream) decide which kernel dominates
    mfma = {k: v for k, v in basis.items() if v["flop_per_launch"] > 0}
    hbmk = {k: v for k, v in basis.items() if v["flop_per_launch"] == 0}
    dom = max(mfma, key=lambda k: mfma[k]["ms_per_step"])
    dom_h = max(hbmk, key=lambda k: hbmk[k]["ms_per_step"]) if hbmk else None
    profiled = trainer.dtype == torch.bfloat16 and args.grad_reg == 0 and world == 1 and eng.G == 98 and headline and args.chunk == CHUNK
    pmc = None
    if profiled:            # HBM bytes from the PMC passes (tools/pmc_bench.sh: FETCH_SIZE x2 + WRITE_SIZE, one counter per rocprofv3 pass, MI355X_MICROARCH.md)
        path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        with open(path) as handle:                          # a missing or unreadable evidence file is an error, not a null
            pmc = json.load(handle)
        if not pmc.get("kernels") or not pmc.get("total_bytes_per_step"):
            raise RuntimeError(f"{path}: no per-kernel PMC traffic in it (regenerate with tools/pmc_bench.sh)")

    def traffic_of(kernel):
        if pmc is None:
            return None
        hits = [v for k, v in pmc["kernels"].items() if kernel in k]
        n = sum(v["launches"] for v in hits)
        return sum(v["bytes_per_launch"] * v["launches"] for v in hits) / n if n else None

    def worst(rows, key):
        rows = [r for r in rows if r["ms_per_step"] >= 0.25]        # launches that matter (>= 0.1 % of the step)
        r = min(rows, key=lambda r: r[key])
        return {k: r[k] for k in ("kernel", "class", "shape", "ms_per_step", "avg_launch_us", "tflops", "gbs_algorithmic", "frac_mfma", "frac_hbm")}

    table = iso or in_region
    conv_rows = [r for r in table if r["flop_per_launch"] > 0]
    flop_img = 3328997376 if headline else sum(conv_flops(eng.plan, 1).values())
    # fp32 storage: a product costs 6 (bf16x6) / 3 (f16x2) 16-bit MFMAs -- `achieved` / `frac` are ALGORITHMIC FLOP over the bf16 peak, and
    # `mfma_per_product` says how many executed MFMAs stand behind one of them
    per_product = 1 if trainer.dtype == torch.bfloat16 else (3 if eng.f32_split == "f16x2" else 6)
    line = {"roofline": {
        "bound": "mfma", "kernel": dom, "achieved": k_in[dom]["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": k_in[dom]["frac_mfma"],
        "traffic": traffic_of(dom), "algorithmic_bytes_per_launch": k_in[dom]["bytes_per_launch"], "flop_per_launch": k_in[dom]["flop_per_launch"],
        "avg_launch_us": k_in[dom]["avg_launch_us"], "launches_per_step": k_in[dom]["launches_per_step"], "mfma_per_product": per_product,
        "frac_isolated": k_iso[dom]["frac_mfma"] if k_iso else None,
        "step_mfma_frac": round(flop_img * passes * trainer.datapoints / sec_per_step / world / (PEAK_BF16_TFLOPS * 1e12), 4)}}
    detail = {"roofline": {
        "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": ("profiles/hbm_traffic.json: " + pmc["command"]) if pmc else None,
        "measured": "HIP events on the launch stream around every launch, over a repetition of the timed steps in the production schedule (weight-gradient "
                    "kernels on a second stream beside these launches); `isolated` = one more step with one stream: every kernel alone on the device",
        "achieved_isolated": k_iso[dom]["tflops"] if k_iso else None,
        "per_kernel": {k: {f: v[f] for f in ("ms_per_step", "launches_per_step", "avg_launch_us", "tflops", "frac_mfma", "gbs_algorithmic", "frac_hbm")} for k, v in basis.items()},
        "per_kernel_in_region": {k: {f: v[f] for f in ("ms_per_step", "tflops", "frac_mfma")} for k, v in k_in.items() if v["flop_per_launch"] > 0},
        "worst_mfma_launch": worst(conv_rows, "frac_mfma"),
        "worst_launch_vs_both_ceilings": worst([dict(r, best=max(r["frac_mfma"], r["frac_hbm"])) for r in conv_rows], "best"),
        "per_shape": [{k: r[k] for k in ("kernel", "class", "shape", "launches_per_step", "ms_per_step", "avg_launch_us", "tflops", "gbs_algorithmic", "frac_mfma", "frac_hbm")}
                      for r in table]}}
    if dom_h is not None:
        line["roofline_hbm"] = {"bound": "hbm", "kernel": dom_h, "achieved": k_in[dom_h]["gbs_algorithmic"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": k_in[dom_h]["frac_hbm"], "traffic": traffic_of(dom_h), "algorithmic_bytes_per_launch": k_in[dom_h]["bytes_per_launch"],
                                "avg_launch_us": k_in[dom_h]["avg_launch_us"], "frac_of_achievable_6290": round(k_in[dom_h]["gbs_algorithmic"] / 6290.0, 4)}
        detail["roofline_hbm"] = {"achieved_isolated": k_iso[dom_h]["gbs_algorithmic"] if k_iso else None}
    if pmc is not None:
        total = pmc["total_bytes_per_step"]
        # (PMC bytes come from the committed rocprofv3 passes -- counters cannot be read from inside this process; the division by THIS run's step time is live)
        line["hbm"] = {"bytes_per_step": total, "tb_per_s": round(total / sec_per_step / 1e12, 3), "frac": round(total / sec_per_step / 1e9 / PEAK_HBM_GBS, 4),
                       "source": "profiles/hbm_traffic.json"}
        detail["hbm"] = {"source": "profiles/hbm_traffic.json: " + pmc["command"], "peak_tb_per_s": PEAK_HBM_GBS / 1e3,
                         "by_kernel_gb_per_step": {k: round(v["bytes_per_launch"] * v["launches"] / 1e9, 2) for k, v in sorted(pmc["kernels"].items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"])}}
    if profiled:            # MFMA utilisation from the hardware counters (tools/pmc_mfma.sh), quoted next to the algorithmic fraction when the file is there
        path = os.path.join(ROOT, "profiles", "mfma_util.json")
        if os.path.isfile(path):
            with open(path) as handle:
                mu = json.load(handle)
            hits = [v for k, v in mu["kernels"].items() if dom in k]
            w = sum(v["ms_per_step"] for v in hits)
            line["mfma_util_pmc"] = {"step": round(mu["mfma_util"], 4), "dominant_kernel": round(sum(v["mfma_util"] * v["ms_per_step"] for v in hits) / w, 4) if w else None,
                                     "executed_mfma_tflop_per_step": round(mu["executed_mfma_tflop_per_step"], 2), "source": "profiles/mfma_util.json"}
            detail["mfma_util_pmc"] = {"step_frac_of_peak_over_kernel_time": round(mu["frac_of_2p5_pflops_over_kernel_time"], 4),
                                       "algorithmic_tflop_per_step": round(flop_img * passes * trainer.datapoints / 1e12, 2),
                                       "source": "profiles/mfma_util.json: " + mu["command"],
                                       "note": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), kernels one at a time"}
    return line, detail


MAX_LINE_BYTES = 4096


def compact_side(side):
    """The numbers of side_configs() that go into the one JSON line (everything else: bench_detail.json)."""
    keep = ("value", "unit", "ms_per_step", "dtype", "steps", "arithmetic", "chunk_group", "error")
    configs = {}
    for name, c in side["configs"].items():
        configs[name] = {k: c[k] for k in keep if k in c}
        if "roofline" in c:
            configs[name]["roofline"] = {k: c["roofline"][k] for k in ("bound", "achieved", "peak", "frac", "mfma_per_product")}
    parity = {}
    for name, p in side["parity"].items():
        if "bit_identical" in p:
            parity[name] = p["bit_identical"]
        elif "rel_l2" in p:
            parity[name] = {k: p[k] for k in ("rel_l2", "cosine", "chunks", "bound", "within_bound") if k in p}
        else:
            parity[name] = {k: v["bit_identical"] for k, v in p.items()}
    return {"configs": configs, "parity": parity}


def compact_cpu_baseline(cpu):
    return {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "host_cpus": cpu.get("host_cpus"), "kind": cpu["kind"], "sample": cpu["sample_short"],
            "grad_reg_on": {"value": cpu["grad_reg_on"]["value"]}}


DETAIL_PATH = os.path.join("gpurun_out", "bench_detail.json")


def write_detail(detail):
    """Per-shape / per-kernel tables, notes and sources: a file next to the run, named in the line."""
    path = os.path.join(ROOT, DETAIL_PATH)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as handle:
        json.dump(detail, handle, indent=1)


def assemble_line(base, extra, roof=None, side=None, cpu=None):
    """(the ONE JSON line rank 0 prints, the detail document).  ``roof`` = roofline_objects(), ``side`` = side_configs(), ``cpu`` =
    cpu_baseline().  The driver parses the last stdout line out of a bounded tail: a line above MAX_LINE_BYTES is an error here, not there."""
    out, detail = dict(base), dict(base, **extra)
    if roof is not None:
        line_part, detail_part = roof
        out.update(line_part)
        for k, v in line_part.items():
            detail[k] = dict(v, **detail_part.get(k, {}))
    if side is not None:
        detail.update(side)
        out.update(compact_side(side))
    if cpu is not None:
        detail["cpu_baseline"] = cpu
        out["cpu_baseline"] = compact_cpu_baseline(cpu)
    out["detail"] = DETAIL_PATH
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > MAX_LINE_BYTES:
        raise RuntimeError(f"bench line is {len(line)} bytes (> {MAX_LINE_BYTES}); move fields to {DETAIL_PATH}")
    return line, detail


def self_launch(n_gpus):
    """Run this script as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>` in a child
    process and pass its output through.  Nothing in the calling process may have initialised the GPU."""
    import socket
    import subprocess

    share = os.environ.get("FB_BENCH_SHARE_DEVICE") == "1"
    visible = torch.cuda.device_count()                 # counting devices does not initialise HIP
    if visible < n_gpus and not share:
        print(f"bench.py --gpus {n_gpus}: only {visible} GPU(s) visible on this node", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL between the ranks (see the pool's driver notes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--grad-reg", type=float, default=0.0, help="block_strength of the finite-difference regulariser (config 3)")
    ap.add_argument("--chunk-group", type=int, default=98)
    ap.add_argument("--images", type=int, default=N_IMAGES)
    ap.add_argument("--chunk", type=int, default=CHUNK, help="data.batch_size = hyp.sub_batch; 125 = the all-50 000-images variant (400 chunks, "
                                                             "stored padded to 128 images per chunk)")
    ap.add_argument("--model", default="resnet18", help="other workloads than the headline one (e.g. BASELINE config 5: --model resnet152 "
                                                        "--stem standard --pixels 224 --images 1024 --grad-reg 0.5): not the benchmark line")
    ap.add_argument("--stem", default="CIFAR", choices=["CIFAR", "standard"])
    ap.add_argument("--pixels", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true", help="skip the grad_reg (BASELINE config 3) timing and the bf16 parity figure")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-r152", action="store_true", help="skip the ResNet-152 @224 lines (BASELINE config 5: configs.r152*)")
    ap.add_argument("--r152-child", action="store_true", help="(internal) run configs.r152* in this process and print them as one JSON object")
    ap.add_argument("--serialize", action="store_true",
                    help="run the weight-gradient kernels on the main stream and take the per-kernel HIP-event timings inside the "
                         "timed region (the protocol the rocprofv3 summaries under profiles/ are generated with)")
    args = ap.parse_args()

    if args.r152_child:
        torch.cuda.set_device(0)
        print(json.dumps(r152_configs(args, torch.device("cuda", 0))), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: this process has not touched the GPU yet; it starts N fresh ranks (one per GPU) as children of
        # torch.distributed.run, relays rank 0's JSON line and exits with the launcher's code (reference fullbatch/utils.py:33-45 is
        # the self-spawning launcher this replaces)
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus} inside a torch.distributed.run job of WORLD_SIZE={world}: the two must agree")
    # FB_BENCH_SHARE_DEVICE=1 (development only): all ranks on cuda:0 over gloo -- exercises the multi-rank orchestration on
    # a 1-GPU box (RCCL refuses two ranks on one device); the number it prints is not a scaling result
    share = os.environ.get("FB_BENCH_SHARE_DEVICE") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # FB_FORCE_DIST=1 with --gpus 1: the sharded step (reduce-scatter on the side stream under the last backward pass, shard-local clip + SGD,
    # all-gather) through RCCL with a process group of ONE rank -- what a 1-GPU box can measure of the exchange: does the collective's kernel
    # get onto the device while the persistent convolutions hold the CUs (`exchange` in the line; FB_CU_RESERVE leaves CUs free for it)
    force_dist = world == 1 and os.environ.get("FB_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # device timestamps around every bucket's reduce-scatter (parallel.BucketExchange): the `exchange` object of every multi-rank line
        os.environ["FB_EXCHANGE_TIMING"] = "1"
        if force_dist:
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device,
                                                 timeout=datetime.timedelta(seconds=300))
        elif share:
            torch.distributed.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
        else:
            # (a collective that never completes fails the run after five minutes instead of holding the node for the default ten-minute watchdog + retries)
            torch.distributed.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(seconds=300))

    from fullbatchtraining_amd import lib
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.models import construct_model
    from fullbatchtraining_amd.training import FullBatchTrainer

    if args.serialize:
        os.environ["FB_WGRAD_STREAM"] = "0"
    n_sched = args.steps + args.warmup + 1          # + the instrumented step after the timed region
    overrides = ["hyp=fb1", "hyp.warmup=0", f"hyp.steps={n_sched}", f"impl.engine.chunk_group={args.chunk_group}",
                 f"impl.mixed_precision={'True' if args.dtype == 'bf16' else 'False'}", "data.augmentations_train="]
    if args.grad_reg != 0:
        overrides += ["hyp=gradreg", "hyp.warmup=0", f"hyp.steps={n_sched}", f"hyp.grad_reg.block_strength={args.grad_reg}"]
    if args.chunk != CHUNK:
        overrides += [f"data.batch_size={args.chunk}", f"hyp.sub_batch={args.chunk}"]
    if world > 1 or force_dist:
        overrides += ["impl/setup=distributed"]
    headline = (args.model, args.stem, args.pixels) == ("resnet18", "CIFAR", 32)
    if not headline:
        overrides += [f"model={args.model}", f"model.stem={args.stem}", f"data.pixels={args.pixels}"]
    cfg = compose(overrides, original_cwd=os.path.join(ROOT, "gpurun_out"), name="bench")
    os.makedirs(cfg.original_cwd, exist_ok=True)

    # synthetic CIFAR-shaped data, generated identically on every rank
    gen = torch.Generator().manual_seed(1234)
    X = torch.randn(args.images, 3, args.pixels, args.pixels, generator=gen)
    Y = torch.randint(0, 10, (args.images,), generator=gen)
    torch.manual_seed(1)
    model = construct_model(cfg.model, 3, 10)
    setup = dict(device=device, dtype=torch.float, memory_format=torch.contiguous_format)
    trainer = FullBatchTrainer(model, (X, Y), None, setup, cfg)
    eng = trainer.engine

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step()
    if trainer.patches is not None:
        eng.choose_schedule(trainer.patches, trainer.labels, trainer.shard.count)      # (a no-op unless --warmup 0 left a wide Bottleneck net's stream choice pending)
    # ---- the timed region: EXACTLY --steps steps of the production schedule, no instrumentation ----
    power = PowerSampler(device.index if device.index is not None else 0) if rank == 0 else None
    sync()
    if power is not None:
        power.start()                # (a host thread reading hwmon files: nothing is queued on the device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step()
    trainer.flush_stats()            # the statistics of every timed step are read back and recorded inside the timed region
    sync()
    elapsed = time.perf_counter() - t0
    power = power.result() if power is not None else None
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    rank_ms = None
    if world > 1:
        every = [torch.zeros_like(t) for _ in range(world)]           # per-rank time of the same K steps: the spread the MAX hides
        torch.distributed.all_gather(every, t)
        rank_ms = [round(1000 * float(e[0]) / args.steps, 2) for e in every]
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t[0])
    # ---- outside the step: the stem's im2col patch gather of the static dataset (once, before the timed region) -- timed here on its own ----
    outside_ms = None
    if trainer.patches is not None and trainer.augment is None and trainer.shuffler is None:
        lo = trainer.shard.first * trainer.chunk
        mine = X[lo:lo + trainer.shard.count * trainer.chunk].to(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        trainer._gather_patches(mine)                                # (the same images into the same buffer: idempotent)
        e1.record()
        torch.cuda.synchronize()
        outside_ms = round(e0.elapsed_time(e1), 3)
        del mine
    enqueue_ms = round(1000 * sum(trainer.enqueue_times[-args.steps:]) / max(args.steps, 1), 2)
    loss_last = trainer.stats["train_loss"][-1]

    # ---- the same --steps steps again with HIP events (libfbengine's fb_profile_*) around every convolution and BatchNorm launch, on its
    # launch stream: the in-region kernel durations of the roofline object (its wall time is reported beside `ms_per_step`) ----
    launches = launches_iso = None
    elapsed_ev = None
    if not args.no_kernel_timing:                       # (every rank steps: the step is collective)
        lib.profile_enable(True, 1 << 17)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            trainer.step()
        trainer.flush_stats()
        sync()
        elapsed_ev = time.perf_counter() - t0
        launches = lib.profile_read_launches()
        lib.profile_read()
        if not args.serialize and world == 1:
            # In the production schedule the weight-gradient kernels run on their own stream beside dgrad / BN backward, so an event bracket
            # there times a kernel that shares the GPU.  One more step with that stream folded into the main one gives each kernel's
            # duration alone on the device ("isolated"): what the rocprofv3 --kernel-trace summaries under profiles/ show.
            saved_stream, saved_lists = eng.wstream, eng.cmdlists
            eng.wstream, eng.cmdlists = None, {}
            trainer.step()
            torch.cuda.synchronize()
            eng.wstream, eng.cmdlists = saved_stream, saved_lists
            launches_iso = lib.profile_read_launches()
            lib.profile_read()
        lib.profile_enable(False)
    if rank == 0:
        images_per_step = trainer.datapoints
        steps_per_sec = args.steps / elapsed
        passes = 1 if args.grad_reg == 0 else 2
        out = {
            "metric": "full-batch GD images/sec (ResNet-18 CIFAR-10 shaped, all chunks accumulated per step)" if headline else
                      f"full-batch GD images/sec ({args.model}, {args.stem} stem, {args.pixels}x{args.pixels} synthetic inputs; NOT the headline workload)",
            "value": round(images_per_step * steps_per_sec, 1), "unit": "images/s",
            "steps_per_sec": round(steps_per_sec, 4), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16" if trainer.dtype == torch.bfloat16 else ("f32" if eng.f32_split != "f16x2" else "f32-22bit"), "data": "synthetic",
            "config": {"workload": f"{'ResNet-18 CIFAR-10' if headline else args.model + ' ' + str(args.pixels) + 'px'} full-batch GD step, {trainer.n_chunks} chunks x {trainer.chunk} = {images_per_step} "
                                   f"images/step (drop_last), grad_reg block_strength={args.grad_reg}, fp32 master/accumulate",
                       "chunk_group": eng.G, "streams": 1 if eng.wstream is None else 2,
                       "parallelism": f"dp{world} (contiguous chunk ranges, reduce-scatter + all-gather)"},
            "train_loss_last": loss_last,
            # host time from the start of a step until its last kernel is queued (mean over the timed steps): launch overhead that the
            # GPU hides as long as it stays below ms_per_step
            "host_enqueue_ms_per_step": enqueue_ms,
            # work of the static-dataset protocol that happens ONCE, before the timed region (fb_stem_patches over this rank's images); with the
            # dataset augmented on the device it is inside every step: configs.augmented
            "outside_step_ms": outside_ms,
        }
        if rank_ms is not None:
            out["rank_ms_per_step"] = {"min": min(rank_ms), "max": max(rank_ms)}
        out["power"] = power
        extra = {"config": dict(out["config"], launches="native command lists (one host call per chunk group)" if eng.use_replay else "one ctypes call per launch (FB_REPLAY=0)"),
                 "outside_the_step": "the dataset is resident in HBM and the stem's im2col patches (fb_stem_patches, 3.3 GB bf16, ~3 ms) are gathered once before the "
                                     "timed region (static, un-augmented dataset); inside: weight prep, all chunk forward/backward passes, running mean, clip + SGD "
                                     "update, statistics read-back"}
        if (world > 1 or force_dist) and getattr(trainer, "_last_exchange", None) is not None:
            # rank 0's view of the last step's exchange: per bucket the reduce-scatter's duration, how long before the main stream needed it it had
            # started (lead_ms) and what was left exposed -- on every multi-rank run (and with FB_FORCE_DIST=1 on one rank)
            torch.cuda.synchronize()
            buckets = trainer._last_exchange.timing_summary()
            late = [b for b in buckets if b["lead_ms"] > 0]
            out["exchange"] = {"ranks": world, "backend": {"nccl": "rccl"}.get(torch.distributed.get_backend(), torch.distributed.get_backend()), "cu_reserve": int(os.environ.get("FB_CU_RESERVE", "0") or 0),
                               "buckets": buckets, "overlap_frac": late[0]["overlap_frac"] if late else 0.0,
                               "exposed_ms": round(sum(b["exposed_ms"] for b in buckets), 3)}
            if rank_ms is not None:
                extra["rank_ms_per_step_all"] = rank_ms
        roof = side = cpu = None
        if launches is not None:
            out["ms_per_step_with_kernel_events"] = round(1000 * elapsed_ev / args.steps, 2)
            roof = roofline_objects(args, trainer, launches, launches_iso, elapsed / args.steps, world, headline, passes)
        if world == 1 and headline and args.grad_reg == 0 and trainer.dtype == torch.bfloat16 and not args.no_side_configs and not force_dist:
            side = side_configs(args, device, X, Y, trainer)
            if args.images == N_IMAGES and not args.no_r152:
                # BASELINE config 5's model, timed by the same run: the headline trainer goes first (its 50 GB would cut the ResNet-152 chunk groups short)
                import gc
                import subprocess
                del trainer, eng
                gc.collect(), torch.cuda.empty_cache()
                # ... in a CHILD process (a fresh HIP context; this process keeps running and relays the result -- no exec): measured on one box, the same three
                # configurations run 3.5-4 % slower at the end of this process's allocation history (788 against 817-824 images/s with the regulariser) than in a
                # process of their own, before and after
                # (a failure of these extra configurations must not take the headline line with it: it is reported inside the line instead)
                try:
                    res = subprocess.run([sys.executable, os.path.abspath(__file__), "--r152-child", f"--chunk-group={args.chunk_group}"], capture_output=True, text=True, timeout=900)
                    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
                    if res.returncode != 0 or not lines:
                        raise RuntimeError(f"exit code {res.returncode}: {res.stderr.strip().splitlines()[-1][:200] if res.stderr.strip() else 'no output'}")
                    side["configs"].update(json.loads(lines[-1]))
                except Exception as exc:                      # noqa: BLE001
                    side["configs"]["r152_error"] = {"value": None, "unit": "images/s", "ms_per_step": None, "dtype": "bf16", "steps": 0, "error": f"{type(exc).__name__}: {exc}"[:300]}
                    print(f"bench.py: the ResNet-152 configurations failed: {exc}", file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
        line, detail = assemble_line(out, extra, roof, side, cpu)
        write_detail(detail)
        sys.stdout.flush()
        print(line, flush=True)
    if world > 1 or force_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
