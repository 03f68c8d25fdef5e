#!/usr/bin/env python
"""Benchmark of the full-batch GD hot path: ResNet-18 / CIFAR-10-shaped synthetic data, one optimizer step = gradient of
ALL chunks (390 x 128 = 49 920 images, the reference's drop_last behaviour) + clip + Nesterov-SGD update.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line (rank 0): whole-job images/s (value), steps/s, ms/step, plus
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
    explicit-backward form is slower on CPU).  Thread counts 8 / 16 / 32 / 64 / all are probed on one chunk each, the best one is then
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
    probe = {}
    saved = torch.get_num_threads()
    for t in sorted({min(c, avail) for c in (8, 16, 32, 64, avail)}):
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
    return {"value": round(k0 * CHUNK / d0, 2), "unit": "images/s", "cores": best, "kind": "port",
            "sample": f"{k0} chunks of {CHUNK} images, fwd+bwd via torch autograd of the fp32 oracle forward, grad_reg off, {d0:.1f} s "
                      f"(one full step = 390 chunks ~ {390 * d0 / k0:.0f} s); thread sweep on one chunk each: "
                      + ", ".join(f"{t}: {v:.0f} img/s" for t, v in probe.items()),
            "grad_reg_on": {"value": round(k1 * CHUNK / d1, 2), "unit": "images/s",
                            "sample": f"{k1} chunks, forward-differences block_strength 0.5, {d1:.1f} s (one step ~ {390 * d1 / k1:.0f} s)"},
            "reference_in_build_container": "SURVEY.md section 6: the real reference, 8 threads: ~122 images/s (off), ~50 images/s (on)"}


def side_configs(args, device, X, Y, main_trainer):
    """Timed in the same run as the headline line (N = 1): BASELINE config 3 (GradRegularizer block_strength 0.5, forward differences,
    fp32 passes) and the distance between the bf16 path that produced `value` and the fp32 path on the mean gradient of 16 chunks."""
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.models import construct_model
    from fullbatchtraining_amd.training import FullBatchTrainer

    n_steps, warm = 2, 1
    cfg = compose(["hyp=gradreg", "hyp.warmup=0", f"hyp.steps={n_steps + warm}", "hyp.grad_reg.block_strength=0.5",
                   f"impl.engine.chunk_group={args.chunk_group}", "impl.mixed_precision=False", "data.augmentations_train="],
                  original_cwd=os.path.join(ROOT, "gpurun_out"), name="bench_gradreg")
    torch.manual_seed(1)
    model = construct_model(cfg.model, 3, 10)
    setup = dict(device=device, dtype=torch.float, memory_format=torch.contiguous_format)
    tr = FullBatchTrainer(model, (X, Y), None, setup, cfg)
    for _ in range(warm):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        tr.step()
    tr.flush_stats()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_steps
    flop = 2 * 3328997376 * tr.datapoints
    mode = tr.engine.f32_split                              # "f16x2": two scaled fp16 pieces per operand, three MFMAs per product; "bf16x6": three bf16 pieces, six
    per_product = 3 if mode == "f16x2" else 6
    arithmetic = ("convolutions on the fp16 matrix pipe: every fp32 operand as two scaled fp16 pieces (22 significand bits, one power-of-two scale per chunk "
                  "and tensor), three MFMAs per product (f16x2)" if mode == "f16x2" else
                  "convolutions on the bf16 matrix pipe with an exact three-way split of every fp32 operand (bf16x6)")
    out = {"configs": {"gradreg": {
        "workload": f"ResNet-18 CIFAR-10 full-batch GD step + GradRegularizer block_strength=0.5 (forward differences, eps 1e-2), {tr.n_chunks} chunks x "
                    f"{tr.chunk}, fp32 storage; {arithmetic}",
        "ms_per_step": round(1000 * dt, 1), "value": round(tr.datapoints / dt, 1), "unit": "images/s", "steps": n_steps, "warmup": warm, "dtype": "f32",
        "tflops": round(flop / dt / 1e12, 1),
        "roofline": {"bound": "mfma", "peak": round(PEAK_BF16_TFLOPS / per_product, 1),
                     "unit": f"TFLOP/s (fp32-equivalent: {per_product} 16-bit MFMAs per fp32 product)",
                     "achieved": round(flop / dt / 1e12, 1), "frac": round(flop / dt / 1e12 / (PEAK_BF16_TFLOPS / per_product), 4),
                     "frac_of_f32_mfma_peak": round(flop / dt / 1e12 / PEAK_F32_TFLOPS, 4)}}}}
    # bf16 vs fp32 on the mean gradient of the first 16 chunks at the benchmark's parameters (tests/test_gpu_bf16_parity.py)
    K = 16
    e16, e32 = main_trainer.engine, tr.engine
    e32.theta.copy_(e16.theta), e32.running_mean.copy_(e16.running_mean), e32.running_var.copy_(e16.running_var)
    res = []
    for eng, t in ((e16, main_trainer), (e32, tr)):
        rm, rv, nbt = eng.running_mean.clone(), eng.running_var.clone(), eng.num_batches_tracked
        eng.full_gradient(t.patches[:K * t.chunk_pad], t.labels[:K * t.chunk_pad], 0.1)
        eng.running_mean.copy_(rm), eng.running_var.copy_(rv)
        eng.num_batches_tracked = nbt
        res.append(eng.avg.double().clone())
    a, b = res[1], res[0]
    out["parity"] = {"bf16_vs_f32_mean_gradient_rel_l2": round(float((a - b).norm() / a.norm()), 4),
                     "cosine": round(float((a * b).sum() / (a.norm() * b.norm())), 5), "chunks": K,
                     "note": "error of the bf16 path on the MEAN gradient of 16 chunks at the benchmark's current parameters; it is noise, not bias: "
                             "0.22 / 0.15 / 0.08 / 0.04 at K = 1 / 4 / 16 / 64 chunks (tests/test_gpu_bf16_parity.py), f32 path vs float64 oracle "
                             "2.5e-6..2e-3 (tests/test_gpu_engine.py)"}
    return out


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
    ap.add_argument("--serialize", action="store_true",
                    help="run the weight-gradient kernels on the main stream and take the per-kernel HIP-event timings inside the "
                         "timed region (the protocol the rocprofv3 summaries under profiles/ are generated with)")
    args = ap.parse_args()

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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=device)

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
    if world > 1:
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
    timing = not args.no_kernel_timing
    if timing:            # HIP events around every convolution launch, on its launch stream, INSIDE the timed region
        lib.profile_enable(True, 65536)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step()
    trainer.flush_stats()            # the statistics of every timed step are read back and recorded inside the timed region
    sync()
    elapsed = time.perf_counter() - t0
    prof, prof_iso = None, None
    if timing:
        prof = lib.profile_read()
        if not args.serialize:
            # In the timed region the weight-gradient kernels run on their own stream beside dgrad / BN backward, so an event bracket there
            # times a kernel that shares the GPU.  One more step of the same workload with that stream folded into the main one gives
            # the duration of each kernel alone on the device ("isolated"), which is what the rocprofv3 summaries in profiles/ show.
            saved, eng.wstream = eng.wstream, None
            trainer.step()
            sync()
            eng.wstream = saved
            prof_iso = lib.profile_read()
        lib.profile_enable(False)
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t[0])

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
            "vs_baseline": None, "dtype": "bf16" if trainer.dtype == torch.bfloat16 else "f32", "data": "synthetic",
            "config": {"workload": f"{'ResNet-18 CIFAR-10' if headline else args.model + ' ' + str(args.pixels) + 'px'} full-batch GD step, {trainer.n_chunks} chunks x {trainer.chunk} = {images_per_step} "
                                   f"images/step (drop_last), grad_reg block_strength={args.grad_reg}, fp32 master/accumulate",
                       "chunk_group": eng.G, "parallelism": f"dp{world} (contiguous chunk ranges, reduce-scatter + all-gather)"},
            "train_loss_last": trainer.stats["train_loss"][-1],
            # host time from the start of a step until its last kernel is queued (mean over the timed steps): launch overhead that the
            # GPU hides as long as it stays below ms_per_step
            "host_enqueue_ms_per_step": round(1000 * sum(trainer.enqueue_times[-args.steps:]) / max(args.steps, 1), 2),
            "outside_the_step": "the dataset is resident in HBM and the stem's im2col patches (fb_stem_patches, 3.3 GB bf16, ~3 ms) are "
                                "gathered once before the timed region (static, un-augmented dataset); inside: weight prep, all chunk "
                                "forward/backward passes, running mean, clip + SGD update, statistics read-back",
        }
        if prof is not None:
            peak = PEAK_BF16_TFLOPS if trainer.dtype == torch.bfloat16 else PEAK_F32_TFLOPS

            def per_class(table, n_steps):
                flops = conv_flops(eng.plan, trainer.shard.count * trainer.chunk * n_steps * passes)
                res = {}
                for k, (ms, launches, dropped) in table.items():
                    if launches:
                        scale = launches / max(launches + dropped, 1)
                        res[k] = {"ms_total": round(ms, 2), "launches": launches, "dropped": dropped, "avg_launch_us": round(1000 * ms / launches, 2),
                                  "tflops": round(flops[k] * scale / (ms * 1e-3) / 1e12, 1)}
                return res, flops

            kernels, flops = per_class(prof, args.steps)
            iso = per_class(prof_iso, 1)[0] if prof_iso is not None else None
            # dominant class = the one with the most kernel time when every kernel has the device to itself (inside the timed region the
            # weight-gradient launches of the second stream overlap the others, so the in-region durations are not additive)
            dom = max(iso or kernels, key=lambda k: (iso or kernels)[k]["ms_total"])
            # HBM bytes per launch of the same kernel class from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, one counter per rocprofv3
            # pass as MI355X_MICROARCH.md prescribes; tools/pmc_bench.sh writes the file, profiles/ keeps the copy behind the number)
            traffic, traffic_src = None, None
            try:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")) as handle:
                    pmc = json.load(handle)
                if trainer.dtype == torch.bfloat16 and args.grad_reg == 0 and world == 1 and eng.G == 98 and headline:   # the profiled configuration
                    traffic, traffic_src = pmc["classes"][dom]["bytes_per_launch"], "profiles/hbm_traffic.json: " + pmc["command"]
            except (OSError, KeyError, ValueError):
                pass
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": kernels[dom]["tflops"], "peak": peak, "unit": "TFLOP/s",
                               "frac": round(kernels[dom]["tflops"] / peak, 4), "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                               "flop_per_launch": flops[dom] / max(kernels[dom]["launches"] + kernels[dom]["dropped"], 1),
                               "avg_launch_us": kernels[dom]["avg_launch_us"], "kernels": kernels,
                               "measured": "HIP events on the launch stream around every launch of the class, inside the timed region"
                                           + (" (--serialize: one stream)" if args.serialize else " (production schedule: weight-gradient kernels "
                                              "run on a second stream beside these launches; `isolated` = the same from one extra step with one stream)"),
                               "achieved_isolated": iso[dom]["tflops"] if iso else None, "frac_isolated": round(iso[dom]["tflops"] / peak, 4) if iso else None,
                               "isolated": iso,
                               "step_mfma_frac": round((3328997376 if headline else sum(conv_flops(eng.plan, 1).values())) * passes * images_per_step * steps_per_sec / world / (peak * 1e12), 4)}
        if world == 1 and headline and args.grad_reg == 0 and trainer.dtype == torch.bfloat16 and not args.no_side_configs:
            out.update(side_configs(args, device, X, Y, trainer))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
