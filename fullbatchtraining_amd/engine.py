"""Host-side schedule of the full-batch gradient hot path on one MI355X.

This is the device counterpart of the reference's chunk loop
(``fullbatch/training/training.py:121-185`` with ``_compute_batched_gradient`` :76-83, ``GradRegularizer`` in
``fullbatch/models/modules.py:211-300`` and ``_stable_mean_accumulation`` :45-47): the network is a fixed DAG, so forward,
dgrad and wgrad are hand-scheduled as calls into ``libfbengine.so`` (no autograd).  ``G`` chunks ("chunk group") are
processed per launch; each chunk keeps its own BatchNorm statistics, loss mean and gradient exactly like the
reference's sequential loop, and the running mean over chunks is folded in the reference's order.

Memory (HBM): one flat fp32 arena per quantity -- theta (master parameters), momentum, averaged gradient, per-chunk
gradients ``g[G][P]`` (+ a second/third set and per-chunk perturbed parameters for the finite-difference passes) --
with conv weights in KRSC order ``[Cout][R*S][Cin]``; NHWC activations in the compute dtype (bf16 or f32).
"""
import math
import os
import time
import warnings

import torch

from . import lib
from .lib import call, _ptr

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _round_up(x, m):
    return (x + m - 1) // m * m


class ConvBN:
    """One convolution with the BatchNorm that follows it."""

    def __init__(self, conv_name, bn_name, cin_real, cout, k, stride, pad, hin, win, patches=False):
        self.conv_name, self.bn_name = conv_name, bn_name
        self.k_orig, self.stride_orig, self.pad_orig = k, stride, pad
        self.hout = (hin + 2 * pad - k) // stride + 1
        self.wout = (win + 2 * pad - k) // stride + 1
        self.patches = patches
        if patches:  # stem: pre-gathered k*k*cin patches, executed as a 1x1 convolution
            self.R = self.S = 1
            self.stride, self.pad = 1, 0
            self.cin_real = k * k * cin_real
            self.hin, self.win = self.hout, self.wout
        else:
            self.R = self.S = k
            self.stride, self.pad = stride, pad
            self.cin_real = cin_real
            self.hin, self.win = hin, win
        self.cin_pad = _round_up(self.cin_real, 32)
        self.cout = cout
        self.taps = self.R * self.S
        self.w_numel = cout * self.taps * self.cin_real
        self.wc_numel = cout * self.taps * self.cin_pad


class Block:
    def __init__(self, convs, shortcut, stride, cin, hin, win):
        self.convs, self.shortcut, self.stride, self.cin, self.hin, self.win = convs, shortcut, stride, cin, hin, win


def max_group(plan, chunk, dtype, device=None, reserve_bytes=0, use_free=True, fd_sets=0):
    """Largest chunk group: as many chunks as the device's memory holds the resident activations of (and a sanity limit of 2^35 bytes per tensor: the persistent
    kernels' tile counts are 32-bit), never less than the group whose biggest activation tensor (NHWC) stays below 2^31 bytes.  Every kernel bases its buffer
    descriptors at its own tile / K slice or addresses with 64-bit pointers -- bf16 since round 3, and the fp32 kernels as well (round 6: the limit that used to hold
    fp32 storage at 2^31 bytes per tensor was a leftover; ResNet-152 @224 with the regulariser: 8 chunks in one group, 3.06 GiB tensors, same losses bit for bit,
    +2.5 %; tests/test_gpu_engine.py::test_f32_chunk_group_beyond_2g_byte_tensors_equals_smaller_groups).  FB_BIG_GROUPS=0 keeps the 2^31 rule.
    ``reserve_bytes`` = what the caller will allocate beside the engine (the stem's pre-gathered patches of the rank's whole shard, resident images); ``fd_sets``: the
    finite-difference passes' per-chunk arenas (gradient sets, perturbed parameters, per-chunk weight copies) count per chunk too.
    ``use_free=False``: a DETERMINISTIC cap -- the device's total memory only, never what happens
    to be free of it now (other processes, allocator state): the caller derives the NOMINAL group from it, the number the weight gradients' K-slice counts are sized
    for, which has to be the same on every rank of a job and in every run (``FullBatchTrainer``: whole-problem reserve, MIN over the ranks)."""
    # (the stem's pre-gathered patches -- 7x7x3 -> 160 values per pixel for the ImageNet stem, the largest tensor by far -- do not count: the two
    # launches that read them are cut into chunk ranges below 2^31 bytes, Engine._stem_ranges; ResNet-152 @224: groups of 10 chunks instead of 4)
    per_image = max(max(L.hout * L.wout * L.cout, L.hin * L.win * (L.cin_pad if L is not plan.stem else 0)) for L in plan.layers)
    es = torch.empty((), dtype=dtype).element_size()
    cap31 = max(1, ((1 << 31) - 1) // (chunk * per_image * es))
    if os.environ.get("FB_BIG_GROUPS", "1") == "0":
        return cap31
    if dtype == torch.float32 and plan.kind == "basic" and os.environ.get("FB_BIG_GROUPS") != "2":
        # BasicBlock nets with fp32 storage keep the old size: measured (round 6, same box, BASELINE config 3: ResNet-18 with the regulariser) 2252 / 2263 ms per step with
        # groups of 56 chunks against 2278 / 2290 with groups of 98 -- the larger tensors buy nothing there (the launches are long already) and cost 1 %;
        # the Bottleneck nets' 4 -> 8 chunks are worth +2.5 %.  FB_BIG_GROUPS=2: the memory rule for every net (tests)
        return cap31
    # beyond the old limit as far as the resident activations of a group fit: every layer's conv output and post-BN activation (a block's last
    # BatchNorm writes the block output: counted once) with their ReLU masks, ~8 gradient buffers of the largest tensor -- against 90 % of the device
    # or what is free of it, less the caller's own tensors.  Round 5 measured ResNet-152 @224: 108.8 GB at 8 chunks of 128 images, 207.9 GB at 16
    # (12.4 GB per chunk + 9.5 GB; this estimate: 13.9 GB per chunk) -- and 16 chunks in ONE group are 2.5 % faster than two groups of 8 (half as many launches
    # for the same work; the old rule, a third of the device, stopped at 10).  fp32 storage with the regulariser (round 6): 211.4 GiB at 8 chunks (estimate 28.9 GB
    # per chunk).  FB_GROUP_MEM_FRAC overrides the 0.9.
    acts = sum(2 * L.hout * L.wout * L.cout for L in plan.layers)
    per_image_bytes = (acts * 17 // 16 + 8 * per_image) * es
    per_chunk_arenas = (1 + (fd_sets + 1 if fd_sets else 0)) * plan.P * 4 + (2 * plan.wc_total * es if fd_sets else 0)
    dev = torch.device(device if device is not None else "cuda")
    if torch.cuda.is_available():
        total = torch.cuda.get_device_properties(dev).total_memory          # (the engine's OWN device: ranks other than local rank 0, heterogeneous boxes)
        free = torch.cuda.mem_get_info(dev)[0]
    else:
        total = free = 288 << 30
    frac = float(os.environ.get("FB_GROUP_MEM_FRAC", "0.9"))
    budget = max(0, (min(int(total * frac), free) if use_free else int(total * frac)) - int(reserve_bytes))
    return max(cap31, min((1 << 35) // (chunk * per_image * es), budget // (chunk * per_image_bytes + per_chunk_arenas)))


def padded_chunk(plan, chunk):
    """Images per chunk AS STORED: the smallest size >= ``chunk`` whose pixels fill whole 128-pixel statistics blocks on every feature
    map of the plan (the conv epilogues and the BN kernels reduce statistics in 128-pixel blocks that must not straddle two chunks).
    128 stays 128; data.batch_size=125 (the all-50 000-images variant, 400 chunks) is stored as 128 images with 3 zero images."""
    mult = 1
    for L in plan.layers:
        mult = math.lcm(mult, 128 // math.gcd(128, L.hout * L.wout))
    return _round_up(chunk, mult)


class Plan:
    """Static layer plan + arena offsets derived from the parameter container (``fullbatchtraining_amd.models.ResNet``)."""

    def __init__(self, model, pixels, arena_align=64):
        """``arena_align``: the arena length is padded to a multiple of it (lcm(64, world size) so that every rank's shard of the
        reduce-scatter has the same length for any number of GPUs)."""
        self.kind, self.stem_kind, self.classes, self.channels = model.kind, model.stem_kind, model.classes, model.channels
        self.pixels = pixels
        named = dict(model.named_parameters())
        self.param_names = [k for k, _ in model.named_parameters()]
        self.param_shapes = {k: tuple(v.shape) for k, v in named.items()}
        # arena offsets (each tensor 16-byte aligned)
        self.offsets, off = {}, 0
        for name in self.param_names:
            self.offsets[name] = off
            off = _round_up(off + named[name].numel(), 4)
        self.P = _round_up(off, math.lcm(64, int(arena_align)))
        self.n_params = sum(v.numel() for v in named.values())
        # layers
        if self.stem_kind == "CIFAR":
            self.stem = ConvBN("stem.0", "stem.1", self.channels, 64, 3, 1, 1, pixels, pixels, patches=True)
            h = self.stem.hout
            self.stem_pool = False
        else:
            self.stem = ConvBN("stem.0", "stem.1", self.channels, 64, 7, 2, 3, pixels, pixels, patches=True)
            h = (self.stem.hout + 2 - 3) // 2 + 1
            self.stem_pool = True
        self.blocks = []
        cin = 64
        for si, stage in enumerate(model.layers):
            for bi, blk in enumerate(stage):
                p = f"layers.{si}.{bi}"
                stride = blk.stride
                planes = blk.conv1.out_channels
                if self.kind == "basic":
                    c1 = ConvBN(f"{p}.conv1", f"{p}.bn1", cin, planes, 3, stride, 1, h, h)
                    c2 = ConvBN(f"{p}.conv2", f"{p}.bn2", planes, planes, 3, 1, 1, c1.hout, c1.wout)
                    convs, cout = [c1, c2], planes
                else:
                    c1 = ConvBN(f"{p}.conv1", f"{p}.bn1", cin, planes, 1, 1, 0, h, h)
                    c2 = ConvBN(f"{p}.conv2", f"{p}.bn2", planes, planes, 3, stride, 1, h, h)
                    c3 = ConvBN(f"{p}.conv3", f"{p}.bn3", planes, planes * 4, 1, 1, 0, c2.hout, c2.wout)
                    convs, cout = [c1, c2, c3], planes * 4
                shortcut = None
                if blk.downsample is not None:
                    hs = h // stride
                    shortcut = ConvBN(f"{p}.downsample.1", f"{p}.downsample.2", cin, cout, 1, 1, 0, hs, hs)
                self.blocks.append(Block(convs, shortcut, stride, cin, h, h))
                cin, h = cout, convs[-1].hout
        self.feat, self.h_final = cin, h
        # "late bucket" of the arena: the parameters of the last stage and the classifier.  The backward pass finishes their gradients
        # first (it walks the blocks in reverse), so their share of the gradient exchange can start while it is still working on the
        # earlier stages (parallel.BucketExchange).  late_block = index of the first block of the last stage.
        self.late_block = sum(len(stage) for stage in list(model.layers)[:-1])
        self.late_offset = self.offsets[f"layers.{len(model.layers) - 1}.0.conv1.weight"]
        self.layers = [self.stem] + [c for b in self.blocks for c in b.convs + ([b.shortcut] if b.shortcut else [])]
        # execution-order independent tables: BN channel table follows state_dict order of BN modules
        order = {name: i for i, name in enumerate(self.param_names)}
        self.layers_by_param = sorted(self.layers, key=lambda L: order[f"{L.conv_name}.weight"])
        ch, wc = 0, 0
        for L in self.layers_by_param:
            L.w_off = self.offsets[f"{L.conv_name}.weight"]
            L.g_off = self.offsets[f"{L.bn_name}.weight"]
            L.b_off = self.offsets[f"{L.bn_name}.bias"]
            L.ch_off, L.wc_off = ch, wc
            ch += L.cout
            wc += L.wc_numel
        self.ch_total, self.wc_total = ch, _round_up(wc, 64)
        self.fcw_off, self.fcb_off = self.offsets["fc.weight"], self.offsets["fc.bias"]


def stem_patches(x_nchw, layer, dtype, aug=None, out=None):
    """[N,C,H,W] fp32 (device) -> [N,Ho,Wo,cin_pad] patches in tap-major (kh,kw,c) order matching the KRSC weights
    (`fb_stem_patches`).  ``aug = (crop_oy, crop_ox, flip, crop_pad, pad_value)``: per-image int8 device tensors for the on-device
    RandomCrop(H, crop_pad) / RandomHorizontalFlip (either may be None) and the per-channel value of a black pixel."""
    if x_nchw.device.type != "cuda":
        raise lib.EngineError("stem_patches: the patch gather runs on the GPU (no CPU path)")
    x = x_nchw.float().contiguous()
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty(n, layer.hout, layer.wout, layer.cin_pad, device=x.device, dtype=dtype)
    oy = ox = fl = pv = None
    crop_pad = 0
    if aug is not None:
        oy, ox, fl, crop_pad, pad_value = aug
        pv = (lib.c_float * c)(*[float(v) for v in pad_value]) if pad_value is not None else None
    call("fb_stem_patches", x.data_ptr(), out.data_ptr(), n, c, h, w, layer.k_orig, layer.stride_orig, layer.pad_orig, layer.cin_pad,
         _ptr(oy), _ptr(ox), _ptr(fl), crop_pad, pv, lib.dtype_code(dtype))
    return out


class _Events:
    """Cross-stream ordering through the library's event table (``fb_event_*``): the same events order eager launches and the launches
    of replayed command lists.  Eager code draws ids from a ring (re-recording an event whose waits have been issued is safe: a wait
    binds to the record that precedes it in host order); a recording gets ids of its own that stay with its list -- fresh ones, or the
    ids of lists the engine has dropped (``release``): library events live as long as the process, so they are reused, never leaked."""
    RING = 1024

    def __init__(self):
        self.ring, self.pos, self.free = [], 0, []

    def release(self, ids):
        self.free.extend(ids)

    def record(self, stream=None):
        if lib.recording():
            ev = self.free.pop() if self.free else lib.event_new()
            lib.current_recorder().events.append(ev)
        elif len(self.ring) < self.RING:
            ev = lib.event_new()
            self.ring.append(ev)
        else:
            ev = self.ring[self.pos]
            self.pos = (self.pos + 1) % self.RING
        lib.event_record(ev, stream)
        return ev

    def wait(self, ev, stream=None):
        lib.event_wait(ev, stream)


class _Pool:
    """Stream-ordered scratch reuse keyed by element count.  A buffer may be returned together with an event (``_Events`` id) of another
    stream that still reads it (the weight-gradient stream): the next user waits for that event before writing."""

    def __init__(self, device, dtype, on_reuse=None, events=None):
        self.device, self.dtype, self.free, self.on_reuse, self.events = device, dtype, {}, on_reuse, events

    def get(self, shape):
        numel = math.prod(shape)
        lst = self.free.get(numel)
        if lst:
            t, ev = lst.pop(0)                       # oldest first: its reader has most likely finished
            if ev is not None:
                self.events.wait(ev)
            if self.on_reuse is not None:
                self.on_reuse(t)
            return t.view(shape)
        return torch.empty(shape, device=self.device, dtype=self.dtype)

    def put(self, *tensors, event=None):
        for t in tensors:
            if t is not None:
                self.free.setdefault(t.numel(), []).append((t, event))


class Engine:
    AMAX_BLOCK, AMAX_SLOT_LIMIT = 256, 16384         # fp16x2 scale slots: rows per allocation, sanity limit on distinct operand buffers

    def __init__(self, model, pixels, chunk, max_groups, compute_dtype=torch.bfloat16, device="cuda", fd_sets=0, arena_align=64,
                 chunk_valid=None, nominal_group=None, f32_split=None):
        """``fd_sets``: extra per-chunk gradient/parameter sets for finite differences (0 none, 1 forward, 2 central).
        ``chunk``: images per statistics group AS STORED; ``chunk_valid`` (default ``chunk``): the real images of a chunk.  A chunk size
        whose pixels do not fill whole 128-pixel statistics blocks on every feature map (e.g. data.batch_size=125) is stored padded
        to ``padded_chunk(...)`` images: the padding images are zeros with label -1, ``fb_bn_apply`` keeps them exactly zero after every
        BatchNorm (so they add nothing to any statistic, weight gradient or loss) and all counts use ``chunk_valid``."""
        lib.load()
        if not torch.cuda.is_available():
            raise lib.EngineError("Engine needs an MI355X: no HIP device visible (there is no CPU path).")
        self.device = torch.device(device)
        self.dt = compute_dtype
        self.dtc = lib.dtype_code(compute_dtype)
        self.plan = Plan(model, pixels, arena_align)
        self.chunk, self.G = chunk, max_groups
        # the chunk group the weight-gradient K-slice counts are sized for: a property of the GLOBAL problem (all ranks and the 1-process run
        # pass the same number), never of the group this engine happens to run -- see _choose_split
        self.nominal_group = int(os.environ.get("FB_NOMINAL_GROUP", "0")) or int(nominal_group or max_groups)
        self.valid = chunk if chunk_valid is None else int(chunk_valid)
        if not 0 < self.valid <= chunk:
            raise lib.EngineError(f"chunk_valid={chunk_valid} outside (0, {chunk}]")
        for L in self.plan.layers:
            if (chunk * L.hout * L.wout) % 128 != 0:
                raise lib.EngineError(f"stored chunk size {chunk} x {L.hout}x{L.wout} feature map of {L.conv_name}: pixels per chunk must be "
                                      "a multiple of 128 (BN statistics are reduced in 128-pixel blocks that must not straddle chunks); "
                                      "pass chunk=padded_chunk(plan, n) and chunk_valid=n for other chunk sizes")
        P = self.plan.P
        f32 = dict(device=self.device, dtype=torch.float32)
        self.theta = torch.zeros(P, **f32)
        self.mom = torch.zeros(P, **f32)
        self.avg = torch.zeros(P, **f32)
        self.g = torch.zeros(self.G, P, **f32)
        self.fd_sets = fd_sets
        self.g_fd = [torch.zeros(self.G, P, **f32) for _ in range(fd_sets)]
        self.theta_k = torch.zeros(self.G, P, **f32) if fd_sets else None
        self.running_mean = torch.zeros(self.plan.ch_total, **f32)
        self.running_var = torch.ones(self.plan.ch_total, **f32)
        self.num_batches_tracked = 0
        self.first_step = True
        # compute-dtype weight copies: [0] one shared set (base pass), [1] one set per chunk (finite-difference passes)
        self.w_fwd = [torch.zeros(1, self.plan.wc_total, device=self.device, dtype=self.dt), None]
        self.w_dgrad = [torch.zeros(1, self.plan.wc_total, device=self.device, dtype=self.dt), None]
        if fd_sets:
            self.w_fwd[1] = torch.zeros(self.G, self.plan.wc_total, device=self.device, dtype=self.dt)
            self.w_dgrad[1] = torch.zeros(self.G, self.plan.wc_total, device=self.device, dtype=self.dt)
        self.n_passes = 1 + fd_sets
        self.mean_tab = torch.zeros(self.n_passes, self.G, self.plan.ch_total, **f32)
        self.var_tab = torch.zeros(self.n_passes, self.G, self.plan.ch_total, **f32)
        self._unbias_tabs = {}
        self.unbias = self._unbias_for(self.valid)
        # fp32 convolutions: "f16x2" (two scaled fp16 pieces per operand, three MFMAs per product; needs the largest magnitude of every
        # operand tensor: fb_absmax, cached per tensor below), "bf16x6" (three bf16 pieces, six MFMAs); FB_F32_EXACT=1 in the library: exact f32
        # Default: bf16x6 everywhere -- the reference runs the regulariser's passes in fp32 (modules.py:226-240) and bf16x6 keeps every product
        # exact to 2^-23.  f16x2 (22-bit operands) is an explicit opt-in: ``f32_split="f16x2"`` (the trainer passes impl.engine.fd_arithmetic) or
        # FB_F32_SPLIT=f16x2; the regulariser's own truncation error (3.5e-2 in fp32) hides its operand rounding (float64 oracle with 22-bit
        # operands: 3.7e-2), at 0.6x the step time
        self.f32_split = os.environ.get("FB_F32_SPLIT", f32_split or "bf16x6") if compute_dtype == torch.float32 else None
        if self.f32_split not in (None, "bf16x6", "f16x2"):
            raise lib.EngineError(f"fp32 arithmetic {self.f32_split!r}: bf16x6 or f16x2")
        self._alloc_activations()
        self.mt_ws = torch.zeros(lib.load().fb_ws_mt_floats(self.G), **f32)
        self.sq = torch.zeros(self.G, **f32)
        self.vnorm2 = torch.zeros(self.G, **f32)
        self.eps_n = torch.zeros(self.G, **f32)
        self.norms2 = torch.zeros(2, **f32)
        # one scale per CHUNK and tensor (a chunk's arithmetic must not depend on how chunks are batched or sharded): slots of G floats
        # Every tensor (buffer address) OWNS its slot for the life of the engine -- the activation, pool and dataset buffers are a fixed
        # set -- so a slot can never be handed to a second tensor while a kernel still reads the first one's scale through it (a
        # 256-entry ring did exactly that on ResNet-152: ~310 slots per forward + backward, fp16 overflow in the early weight gradients).
        # ``amax_map`` says whose slot holds the magnitudes of the tensor's CURRENT content.
        self.amax_blocks, self.amax_slots, self.amax_map, self.amax_handouts = [], {}, {}, 0
        for li, L in enumerate(self.plan.layers):
            L.li = li
        self.w_amax = [torch.zeros(len(self.plan.layers), self.G if k else 1, device=self.device, dtype=torch.float32) for k in range(2)]
        self.amax_ws = None
        if self.f32_split == "f16x2":                # scratch of the apply passes that track the magnitudes of their own output
            n_max = self.G * chunk
            need = max(lib.load().fb_ws_bn_amax_floats(n_max * L.hout * L.wout, L.cout, chunk * L.hout * L.wout) for L in self.plan.layers)
            self.amax_ws = torch.zeros(int(need), device=self.device, dtype=torch.float32)
        self.events = _Events()
        self.pool = _Pool(self.device, self.dt, on_reuse=lambda t: self.amax_map.pop(t.data_ptr(), None), events=self.events)
        # native launch executor (csrc/cmdlist.cpp): the launches of one chunk group's forward + backward (and of the weight preparation) are a
        # static sequence -- recorded the first time, replayed with one host call afterwards (FB_REPLAY=0: every launch through ctypes)
        self.use_replay = os.environ.get("FB_REPLAY", "1") != "0"
        self.cmdlists, self.replays, self.MAX_CMDLISTS = {}, 0, int(os.environ.get("FB_MAX_CMDLISTS", "256"))
        self.cmd_evictions, self.record_new, self.unrecorded_runs, self._warned_record_off, self._off_misses = 0, True, 0, False, 0
        self.masks = {}
        # FB_EXPERIMENTAL=1: the switches below that turn ON a form which lost its same-box A/B act only together with it (the library's do the same: csrc/runtime.cpp)
        exp = os.environ.get("FB_EXPERIMENTAL", "0") not in ("", "0")
        self.fuse_bwd_stat = exp and os.environ.get("FB_FUSED_BWD_STAT", "0") != "0"     # BN-backward reduction in the input-gradient epilogues: built, parity-tested,
        # measured SLOWER at step level (profiles/r2_notes.md: the separate HBM-bound reduction overlaps the weight-gradient stream) -> off
        self.bst_done = False
        # BatchNorm backward in one pass over (dout, x) with a chunk's operands held in registers by a resident cluster of workgroups
        # (csrc/bn_bwd_fused.hip): built, parity-tested, and measured SLOWER than reduce -> finalize -> apply (2.1 vs 1.27 ms on the 64-channel layer:
        # eight waves per CU pulling 128 KiB bursts reach 2.8 TB/s even without the waits -- profiles/r4_notes.md) -> off; FB_BN_BWD_FUSED=1 selects it
        self.bn_fused = exp and os.environ.get("FB_BN_BWD_FUSED", "0") == "1"
        # partial rows: (vectors of the tensor / 4096) x 2C floats, whatever the grouping -- sized for the largest layer of a full group
        es = torch.empty((), dtype=self.dt).element_size()
        self.bnf_ws = torch.empty(max((self.G * chunk * L.hout * L.wout * L.cout * es // 16 // 4096 * 2 + 4 * self.G) * L.cout for L in self.plan.layers) + 64,
                                  device=self.device, dtype=torch.float32)
        self.bnf_sync = torch.zeros(int(lib.load().fb_ws_bn_bwd_fused_ints(self.G)), device=self.device, dtype=torch.int32)
        # weight gradients depend on nothing downstream in the backward chain: they run on their own stream, overlapping the
        # HBM-bound BN backward kernels and the dgrad convolutions of the main stream (FB_WGRAD_STREAM=0: same stream)
        sw = os.environ.get("FB_WGRAD_STREAM")
        self.wstream = torch.cuda.Stream(device=self.device) if sw != "0" else None
        # Whether the second stream pays depends on what it would run beside: the 3 x 3 layers of a BasicBlock net leave room on a CU (ResNet-18 @32: -1..4 % step
        # time, ResNet-50 @32: -5 %), the big-tile 1 x 1 kernels of a wide Bottleneck net take a CU's whole LDS / register file and the two streams only get
        # in each other's way (ResNet-152 @224 bf16: ONE stream is 1.5 % faster; with the fp32 regulariser two are 1.3 % faster again; round 5, same box).
        # Both schedules give the same bits (test_two_stream_schedule_is_bit_identical_to_one_stream), so for such nets the first full_gradient call times one
        # chunk group both ways and keeps the faster one.  FB_WGRAD_STREAM=0 / 1 pins the choice.
        self.stream_autotune = sw not in ("0", "1") and any(L.R == 1 and L.cin_pad >= 512 for L in self.plan.layers)
        self.stream_times = None
        self._wgrad_event = None
        self.label_smoothing, self.only_incorrect = 0.0, False      # loss function of the head kernel (reference get_loss_fn)
        self.load_from_model(model)

    def _unbias_for(self, batch):
        """Bessel factors m/(m-1) per BN channel for BN batches of ``batch`` images (the running variance is the unbiased one)."""
        tab = self._unbias_tabs.get(batch)
        if tab is None:
            unbias = torch.ones(self.plan.ch_total)
            for L in self.plan.layers:
                m = batch * L.hout * L.wout
                unbias[L.ch_off:L.ch_off + L.cout] = m / (m - 1)
            tab = self._unbias_tabs[batch] = unbias.to(self.device)
        return tab

    # ------------------------------------------------------------------------------------------------------ buffers --
    def _alloc_activations(self):
        n, dev, dt = self.G * self.chunk, self.device, self.dt
        f32 = dict(device=dev, dtype=torch.float32)
        max_part, max_slab = 0, 0
        for L in self.plan.layers:
            L.x = torch.empty(n, L.hout, L.wout, L.cout, device=dev, dtype=dt)   # raw conv output (pre-BN)
            L.scale = torch.empty(self.G, L.cout, **f32)
            L.shift = torch.empty(self.G, L.cout, **f32)
            L.invstd = torch.empty(self.G, L.cout, **f32)
            L.coef = torch.empty(self.G, L.cout, 3, **f32)
            px = n * L.hout * L.wout
            L.split_k = self._choose_split(L)
            # scratch sizes from the library's own queries (conv statistics / BN-backward partial sums share one buffer)
            handle = lib.load()
            max_part = max(max_part, handle.fb_ws_bn_partial_floats(px, L.cout),
                           handle.fb_ws_conv_stat_floats(lib.C.byref(lib.ConvArgs(n_img=n, Hd=L.hout, Wd=L.wout, Cd=L.cout))))
            max_slab = max(max_slab, handle.fb_ws_wgrad_slab_floats(lib.C.byref(lib.WgradArgs(
                n_img=n, imgs_per_group=self.chunk, split_k=L.split_k, Cd=L.cout, R=L.R, S=L.S, Cs=L.cin_pad))))
        self.stat_ws = torch.empty(max_part, **f32)
        self.slab_ws = torch.empty(max_slab, **f32)
        # Chunk-chained weight gradients (fb_conv2d_wgrad_chain; full_gradient's plain bf16 path): for the 3x3 layers on 4x4 maps -- 7 of the 11 M
        # parameters of ResNet-18 -- the kernel leaves the SUM of the group's chunk gradients in ``gsum`` and every chunk's sum of squares in
        # ``L.chain_sq`` instead of one fp32 gradient per chunk in the arena (2.8 GB per group written there and read again by the running mean).
        self.chain_layers, self.chain_on = [], False
        # OPT-IN (FB_WGRAD_CHAIN=1): the running-mean pass drops from 4.0 to 1.7 ms/step, but the chained kernel -- one 8-wave workgroup per CU with
        # two accumulator sets -- takes 840 us where the per-chunk kernel with two independent 4-wave workgroups takes 692 (profiles/r4_notes.md)
        if dt == torch.bfloat16 and os.environ.get("FB_WGRAD_CHAIN", "0") == "1" and os.environ.get("FB_EXPERIMENTAL", "0") not in ("", "0"):
            for L in self.plan.layers:
                a = lib.WgradArgs(None, None, None, n, L.hin, L.win, L.cin_pad, L.hout, L.wout, L.cout, L.R, L.S, L.stride, L.pad, self.chunk, 1, self.dtc, 0)
                # (at most four: fb_mt_accumulate_skip leaves four ranges of the arena alone)
                if (len(self.chain_layers) < 4 and L.split_k == 1 and L.cin_pad == L.cin_real and L.w_off % 4 == 0
                        and bool(lib.load().fb_wgrad_chain_supported(lib.C.byref(a)))):
                    L.chain_tiles = (L.cout // 64) * (L.cin_pad // 64)
                    # one 8-wave workgroup per CU and chain; FB_WGRAD_CHAINS overrides the number of chains per tile
                    L.n_chains = int(os.environ.get("FB_WGRAD_CHAINS", "0")) or max(1, 512 // L.chain_tiles)
                    L.chain_sq = torch.zeros(self.G, L.chain_tiles * 8, **f32)
                    self.chain_layers.append(L)
        if self.chain_layers:
            self.chain_ws = torch.empty(max(min(L.n_chains, self.G) * L.cout * L.taps * L.cin_pad for L in self.chain_layers), **f32)
            self.gsum = torch.zeros(self.plan.P, **f32)
            # the arena ranges that stay per chunk: [0, P) minus the chained layers' weights
            cuts, lo = [], 0
            for L in sorted(self.chain_layers, key=lambda l: l.w_off):
                if L.w_off > lo:
                    cuts.append((lo, L.w_off))
                lo = L.w_off + L.cout * L.taps * L.cin_real
            if lo < self.plan.P:
                cuts.append((lo, self.plan.P))
            assert all(a % 4 == 0 for a, _ in cuts), cuts
            self.plain_segments = cuts
            # scratch rows for the per-chunk norms of the non-chained ranges: one per caller of _fold that may be in flight at the same time
            # (row 0: the main / weight-gradient stream, row 1: the side stream of the multi-GPU late bucket)
            self.sq_seg = torch.zeros(2, self.G, **f32)
        self.stem_out = torch.empty(n, self.plan.stem.hout, self.plan.stem.wout, 64, device=dev, dtype=dt)
        if self.plan.stem_pool:
            hp = (self.plan.stem.hout + 1) // 2
            self.stem_pooled = torch.empty(n, hp, hp, 64, device=dev, dtype=dt)
            # the argmax of every pooling window (one byte per pooled element): the backward pass reads it instead of recomputing 4 windows x 9 values per input
            # quad from the pre-pool tensor (fb_maxpool3s2_bwd_idx: 2.0 -> ~0.5 ms per 1024 images of the ImageNet stem, same bits); FB_MAXPOOL_IDX=0: recompute
            self.stem_pool_idx = torch.empty(n, hp, hp, 64, device=dev, dtype=torch.uint8) if os.environ.get("FB_MAXPOOL_IDX", "1") != "0" else None
        for b in self.plan.blocks:
            b.mids = [torch.empty_like(c.x) for c in b.convs[:-1]]              # post BN-ReLU activations inside the block
            b.out = torch.empty_like(b.convs[-1].x)                              # block output (post add + ReLU)
            b.pooled = None
            if b.shortcut is not None and b.stride == 2:
                b.pooled = torch.empty(n, b.hin // 2, b.win // 2, b.cin, device=dev, dtype=dt)
        self.feat = torch.empty(n, self.plan.feat, **f32)
        self.logits = torch.empty(n, self.plan.classes, **f32)
        self.dlogits = torch.empty(n, self.plan.classes, **f32)
        self.loss = torch.zeros(self.G, **f32)
        self.correct = torch.zeros(self.G, **f32)

    # K-slice counts are sized for ``self.nominal_group`` -- the chunk group of the whole problem on ONE GPU (the trainer passes
    # group_size(all chunks, chunk_group, cap), the same number on every rank) -- never for the group this engine runs: the number of slices
    # fixes the order in which a chunk's pixels are summed, and a chunk's gradient must not depend on how chunks are batched or sharded (the
    # 2-rank == 1-process tests).  A rank of an 8-GPU job (49 chunks, slices sized for 98) pays 1 % for it: 32.08 vs 31.76 ms/step.
    def _choose_split(self, L):
        """Split the pixel reduction of wgrad so that >= ~1000 workgroups exist; slices are multiples of the K-step."""
        bf16 = self.dt == torch.bfloat16
        G = self.nominal_group
        widths = ((4, 8, 16, 32) + ((14, 28, 56) if bf16 else ())) if L.stride == 1 else ((4, 8, 16) if (bf16 or self.f32_split == "f16x2") else ())
        if (L.R == 3 and L.stride in (1, 2) and L.pad == 1 and L.hout == L.wout and L.hin == L.stride * L.hout and L.wout in widths
                and L.cin_pad % 64 == 0 and L.cout % 64 == 0 and not (L.wout == 4 and self.chunk % 2)):
            # all-taps halo wgrad kernel: split-K over whole images (pairs for 4x4 maps); any split works (ragged last slice).
            # One wave of workgroups: as many K slices as fit the resident slots (2 workgroups per CU; the stride-2 4x4 kernel 1)
            tiles = (L.cout // 64) * (L.cin_pad // 64)
            slots = 256 if (L.stride == 2 and L.wout == 4) else 512
            unit = 2 if L.wout == 4 else 1
            return max(1, min(slots // (tiles * G), self.chunk // (2 * unit)))
        big = L.cin_pad % 128 == 0 and L.cout % 128 == 0 and (L.cin_pad >= 256 or L.cout >= 256 or L.R == 3)   # = conv_wgrad.hip
        tile = 128 if big else 64
        tiles = (L.cout // tile) * max(L.cin_pad // tile, 1) * L.taps
        px = self.chunk * L.hout * L.wout
        kstep = (32 if self.dt == torch.float32 else 64) if big else 128
        want = max(1, 1024 // max(tiles * G, 1))
        if L is self.plan.stem and L.cin_pad % 64 != 0 and len(self._stem_ranges(G)) > 1:
            # the stem on its pre-gathered patches is launched range by range (_stem_ranges: 4 chunks of the ImageNet stem), and bf16 patches of 160 values are ONE
            # 64 x 160 tile per slice (conv_wgrad.hip): sized by the group, a 16-chunk ResNet-152 group left it 128 workgroups (1.6 ms per launch at 1.8 TB/s)
            per_range = self._stem_ranges(G)[0][1]
            tiles = (L.cout // 64) * (1 if (bf16 and L.cin_pad % 160 == 0) else L.cin_pad // 32)
            want = max(1, 512 // max(tiles * per_range, 1))
        split = max(1, min(want, px // (kstep * 4)))
        while split > 1 and (split - 1) * (_round_up(-(-px // split), kstep)) >= px:
            split -= 1
        if self.dt == torch.float32 and L is not self.plan.stem and os.environ.get("FB_WGRAD_ROUNDS", "1") != "0":
            # fp32 storage (six MFMAs per product: these launches are a quarter of the regularised ResNet-152 step): all workgroups of a launch take the same time, so
            # a launch costs ceil(workgroups / resident slots) rounds -- "about 1000 workgroups" left the 256 -> 256 3x3 layers @14x14 of an 8-chunk group at 864 of
            # 2 x 512 slots (0.84).  Take the smallest K-slice count whose last round is (nearly) full.  (Round 6; conv_wgrad_kernel<f32s_tag, 2, 2, 1, 1, 4>: two
            # workgroups per CU by registers, and so has the 64 x 64-tile form since its waves are arranged 2 x 2.)
            slots = 2 * 256
            unit = max(tiles * G, 1)

            def ok(sp):
                return sp == 1 or (sp - 1) * (_round_up(-(-px // sp), kstep)) < px

            def eff(sp):
                n = unit * sp
                return n / (-(-n // slots) * slots)

            cands = [sp for sp in range(1, max(1, min(px // (kstep * 4), 4 * max(want, 1))) + 1) if ok(sp)]
            top = max(eff(sp) for sp in cands)
            split = min(sp for sp in cands if eff(sp) >= top - 0.02)
        return split

    # ------------------------------------------------------------------------------------------- parameter exchange --
    def _krsc(self, L, w):
        return w.permute(0, 2, 3, 1).reshape(-1)

    def load_from_model(self, model):
        """Mirror parameters/buffers of the container into the arena (KRSC permutation for conv weights)."""
        named = dict(model.named_parameters())
        flat = torch.zeros(self.plan.P)
        conv_names = {f"{L.conv_name}.weight" for L in self.plan.layers}
        for name in self.plan.param_names:
            w = named[name].detach().float().cpu()
            v = w.permute(0, 2, 3, 1).reshape(-1) if name in conv_names else w.reshape(-1)
            flat[self.plan.offsets[name]:self.plan.offsets[name] + v.numel()] = v
        self.theta.copy_(flat)
        bufs = dict(model.named_buffers())
        rm, rv = torch.zeros(self.plan.ch_total), torch.ones(self.plan.ch_total)
        for L in self.plan.layers:
            rm[L.ch_off:L.ch_off + L.cout] = bufs[f"{L.bn_name}.running_mean"].detach().float().cpu()
            rv[L.ch_off:L.ch_off + L.cout] = bufs[f"{L.bn_name}.running_var"].detach().float().cpu()
        self.running_mean.copy_(rm)
        self.running_var.copy_(rv)
        self.num_batches_tracked = int(bufs[f"{self.plan.stem.bn_name}.num_batches_tracked"])

    def _unflatten(self, flat, name):
        shape = self.plan.param_shapes[name]
        v = flat[self.plan.offsets[name]:self.plan.offsets[name] + math.prod(shape)]
        if len(shape) == 4:
            co, ci, kh, kw = shape
            return v.view(co, kh, kw, ci).permute(0, 3, 1, 2).contiguous()
        return v.view(shape).clone()

    def store_to_model(self, model, with_grad=False):
        """Write arena parameters / BN buffers (and optionally ``p.grad`` = averaged gradient) back into the container."""
        flat = self.theta.detach().cpu()
        gflat = self.avg.detach().cpu() if with_grad else None
        with torch.no_grad():
            for name, p in model.named_parameters():
                p.copy_(self._unflatten(flat, name).to(p.dtype))
                if with_grad:
                    p.grad = self._unflatten(gflat, name).to(p.dtype).to(p.device)
            bufs = dict(model.named_buffers())
            rm, rv = self.running_mean.cpu(), self.running_var.cpu()
            for L in self.plan.layers:
                bufs[f"{L.bn_name}.running_mean"].copy_(rm[L.ch_off:L.ch_off + L.cout])
                bufs[f"{L.bn_name}.running_var"].copy_(rv[L.ch_off:L.ch_off + L.cout])
                bufs[f"{L.bn_name}.num_batches_tracked"].fill_(self.num_batches_tracked)

    def flatten(self, tensors):
        """List of tensors in ``model.parameters()`` order/layout (OIHW) -> flat arena vector (KRSC) on the engine's device."""
        flat = torch.zeros(self.plan.P, device=self.device, dtype=torch.float32)
        for name, t in zip(self.plan.param_names, tensors):
            v = t.detach().to(self.device, torch.float32)
            v = v.permute(0, 2, 3, 1).reshape(-1) if v.dim() == 4 else v.reshape(-1)
            flat[self.plan.offsets[name]:self.plan.offsets[name] + v.numel()] = v
        return flat

    def unflatten_list(self, flat):
        """Flat arena vector -> list of tensors in the reference's layouts (same device as ``flat``)."""
        return [self._unflatten(flat, name) for name in self.plan.param_names]

    def store_buffers_to_model(self, model):
        with torch.no_grad():
            bufs = dict(model.named_buffers())
            for L in self.plan.layers:
                bufs[f"{L.bn_name}.running_mean"].copy_(self.running_mean[L.ch_off:L.ch_off + L.cout])
                bufs[f"{L.bn_name}.running_var"].copy_(self.running_var[L.ch_off:L.ch_off + L.cout])
                bufs[f"{L.bn_name}.num_batches_tracked"].fill_(self.num_batches_tracked)

    def momentum_state(self):
        flat = self.mom.detach().cpu()
        return [self._unflatten(flat, name) for name in self.plan.param_names]

    def sam_state(self):
        flat = self.e_w.detach().cpu()
        return [self._unflatten(flat, name) for name in self.plan.param_names]

    def load_momentum(self, tensors):
        flat = torch.zeros(self.plan.P)
        for name, t in zip(self.plan.param_names, tensors):
            v = t.detach().float().cpu()
            v = v.permute(0, 2, 3, 1).reshape(-1) if v.dim() == 4 else v.reshape(-1)
            flat[self.plan.offsets[name]:self.plan.offsets[name] + v.numel()] = v
        self.mom.copy_(flat)
        self.first_step = False

    # --------------------------------------------------------------------------------------------------- primitives --
    def prep_weights(self, theta, nsets, per_chunk=False):
        """fp32 master (one shared set or one per chunk) -> compute-dtype forward and transposed dgrad copies."""
        slot = 1 if per_chunk else 0
        wf, wd = self.w_fwd[slot], self.w_dgrad[slot]
        es = wf.element_size()

        def body():
            for li, L in enumerate(self.plan.layers):
                dst_d = wd.data_ptr() + es * L.wc_off if L is not self.plan.stem else None
                am = None
                if self.f32_split == "f16x2":        # one scale per layer and weight set, shared by its forward and transposed copies;
                    am = self.w_amax[slot][li].data_ptr()                       # the copies are then written as fp16x2 planes
                    call("fb_absmax", theta.data_ptr() + 4 * L.w_off, L.cout * L.taps * L.cin_real, nsets, self.plan.P, 1, am)
                call("fb_weight_prep", theta.data_ptr() + 4 * L.w_off, self.plan.P, self.plan.wc_total, nsets, L.cout, L.taps, L.cin_real,
                     L.cin_pad, wf.data_ptr() + es * L.wc_off, dst_d, self.dtc, am)

        self._replayable(("prep", theta.data_ptr(), nsets, slot), body)

    def _amax(self, t, numel, G):
        """Device pointer of the per-chunk largest magnitudes (G floats) of the first ``numel`` values of ``t`` (fp16x2 split scales),
        computed once per content: the cache is dropped when a forward pass starts and when the pool hands the buffer out again."""
        key = t.data_ptr()
        hit = self.amax_map.get(key)
        if hit is not None and hit[1] == numel:
            return hit[0]
        slot = self._amax_slot(t, numel)
        call("fb_absmax", t.data_ptr(), numel // G, G, numel // G, 1, slot)
        return slot

    def _amax_slot(self, t, numel):
        """Slot (G floats) for a tensor whose PRODUCER tracks the largest magnitudes itself (fb_bn_apply / fb_bn_bwd_apply ``amax_out``)."""
        key = t.data_ptr()
        self.amax_handouts += 1
        slot = self.amax_slots.get(key)
        if slot is None:
            idx = len(self.amax_slots)
            if idx >= self.AMAX_SLOT_LIMIT:
                raise lib.EngineError(f"fp16x2 scale slots: more than {self.AMAX_SLOT_LIMIT} distinct operand buffers -- buffers are expected to be "
                                      "persistent (activations, pool, dataset slices); something allocates a new tensor per launch")
            blk, row = divmod(idx, self.AMAX_BLOCK)
            if blk == len(self.amax_blocks):         # (earlier blocks stay where they are: their slot pointers are in flight)
                self.amax_blocks.append(torch.zeros(self.AMAX_BLOCK, self.G, device=self.device, dtype=torch.float32))
            slot = self.amax_slots[key] = self.amax_blocks[blk][row].data_ptr()
        self.amax_map[key] = (slot, numel)
        return slot

    def _amax_pair(self, L, src, numel, wsets, G):
        if self.f32_split != "f16x2":
            return None, None
        return self._amax(src, numel, G), self.w_amax[1 if wsets > 1 else 0][L.li].data_ptr()

    def _stem_ranges(self, G):
        """Chunk ranges [(g0, g_n)] of a group whose stem patches stay below 2^31 bytes each (32-bit buffer offsets of the LDS-DMA kernels)."""
        S = self.plan.stem
        per_chunk = self.chunk * S.hin * S.win * S.cin_pad * torch.empty((), dtype=self.dt).element_size()
        step = max(1, int(os.environ.get("FB_STEM_RANGE_BYTES", (1 << 31) - 1)) // per_chunk)       # (the variable: tests force several ranges)
        return [(g0, min(step, G - g0)) for g0 in range(0, G, step)]

    def _conv_bn_fwd(self, L, src, G, wsets, theta, pidx):
        if L is self.plan.stem and len(self._stem_ranges(G)) > 1 and not getattr(self, "_in_stem_range", False):
            # the stem of a large group, range by range: every per-chunk quantity (statistics tables, scale / shift, outputs) is indexed by
            # chunk, so a range is the same call on offset views
            self._in_stem_range = True
            try:
                amax = self._amax_pair(L, src, G * self.chunk * L.hin * L.win * L.cin_pad, wsets, G)      # per-chunk scales of the WHOLE group, once
                for g0, g_n in self._stem_ranges(G):
                    self._conv_bn_fwd_range(L, src, g0, g_n, wsets, theta, pidx, amax)
            finally:
                self._in_stem_range = False
            return
        self._conv_bn_fwd_range(L, src, 0, G, wsets, theta, pidx)

    def _conv_bn_fwd_range(self, L, src, g0, G, wsets, theta, pidx, amax=None):
        """conv + batch statistics of chunks [g0, g0 + G) of the launch group (g0 > 0 only for the stem of a large group)."""
        es = src.element_size()
        i0 = g0 * self.chunk                              # first image
        src_ptr = src.data_ptr() + i0 * L.hin * L.win * L.cin_pad * es
        x_ptr = L.x.data_ptr() + i0 * L.hout * L.wout * L.cout * es
        n = G * self.chunk
        wf = self.w_fwd[1 if wsets > 1 else 0]
        wptr = wf.data_ptr() + wf.element_size() * L.wc_off
        evalm = getattr(self, "_eval", False)
        if amax is None:
            am_s, am_w = self._amax_pair(L, src, n * L.hin * L.win * L.cin_pad, wsets, G)
        else:                                             # a range of a larger group: per-chunk entries (and per-chunk weight sets) start at chunk g0
            am_s = amax[0] + 4 * g0 if amax[0] is not None else None
            am_w = (amax[1] + (4 * g0 if wsets > 1 else 0)) if amax[1] is not None else None
        if wsets > 1:
            wptr += g0 * self.plan.wc_total * wf.element_size()
        a = lib.ConvArgs(src_ptr, wptr, x_ptr, None, None if evalm else self.stat_ws.data_ptr(), n, L.hin, L.win, L.cin_pad, L.hout, L.wout,
                         L.cout, L.R, L.S, L.stride, L.pad, 0, self.chunk if wsets > 1 else n, self.plan.wc_total if wsets > 1 else 0,
                         0, self.dtc, None, None, None, am_s, am_w, self.chunk)
        call("fb_conv2d", lib.C.byref(a))
        if evalm:
            return
        px = n * L.hout * L.wout
        n_mblocks = (px + 127) // 128
        pstride = self.plan.P if wsets > 1 else 0
        tab_off = 4 * g0 * self.plan.ch_total            # per-chunk rows of the statistics tables / coefficient arrays
        th_off = 4 * g0 * pstride                         # per-chunk parameter sets
        call("fb_bn_fwd_finalize", self.stat_ws.data_ptr(), n_mblocks, G, L.cout, float(self.valid * L.hout * L.wout),
             theta.data_ptr() + 4 * L.g_off + th_off, theta.data_ptr() + 4 * L.b_off + th_off, pstride, BN_EPS,
             self.mean_tab[pidx].data_ptr() + tab_off, self.var_tab[pidx].data_ptr() + tab_off, self.plan.ch_total, L.ch_off,
             L.scale.data_ptr() + 4 * g0 * L.cout, L.shift.data_ptr() + 4 * g0 * L.cout, L.invstd.data_ptr() + 4 * g0 * L.cout)

    def _bn_apply(self, L, out, G, relu=True, res=None, resL=None, pool=None):
        """``pool``: the 2x2-average-pooled copy of ``out`` the NEXT block's shortcut reads, written by the same pass where the library can
        (``fb_bn_apply_can_pool``); returns True if it was."""
        px = G * self.chunk * L.hout * L.wout
        ppg = self.chunk * L.hout * L.wout
        fused = pool is not None and bool(lib.load().fb_bn_apply_can_pool(L.cout, L.wout, ppg, self.dtc)) and os.environ.get("FB_FUSED_POOL", "1") != "0"
        call("fb_bn_apply", L.x.data_ptr(), out.data_ptr(), L.scale.data_ptr(), L.shift.data_ptr(), _ptr(res),
             resL.scale.data_ptr() if resL is not None else None, resL.shift.data_ptr() if resL is not None else None,
             px, L.cout, ppg, self.valid * L.hout * L.wout if self.valid < self.chunk else 0, 1 if relu else 0,
             _ptr(self._mask_of(out)) if relu else None, _ptr(pool) if fused else None, L.wout, self.dtc,
             *((self._amax_slot(out, px * L.cout), self.amax_ws.data_ptr()) if self.f32_split == "f16x2" else (None, None)))
        return fused

    def _mask_of(self, act):
        """ReLU bitmask buffer (1 byte per 16-byte vector) paired with a post-ReLU activation tensor, created on first use."""
        key = act.data_ptr()
        m = self.masks.get(key)
        if m is None:
            m = torch.empty(act.numel() * act.element_size() // 16, device=self.device, dtype=torch.uint8)
            self.masks[key] = m
        return m

    def _sl(self, t, G):
        return t[: G * self.chunk]

    # ------------------------------------------------------------------------------------------------------ forward --
    def forward(self, patches, labels, G, wsets, theta, pidx):
        """patches: [G*chunk, H, W, cin_pad] stem patches; labels int64 [G*chunk].  Fills loss/correct/dlogits."""
        plan = self.plan
        self.amax_map.clear()                        # every activation is about to be rewritten
        self._conv_bn_fwd(plan.stem, patches, G, wsets, theta, pidx)
        self._bn_apply(plan.stem, self.stem_out, G)
        a = self.stem_out
        if plan.stem_pool:
            s = plan.stem
            if self.stem_pool_idx is not None and not getattr(self, "_eval", False):
                call("fb_maxpool3s2_fwd_idx", a.data_ptr(), self.stem_pooled.data_ptr(), self.stem_pool_idx.data_ptr(), G * self.chunk, s.hout, s.wout, 64, self.dtc)
            else:
                call("fb_maxpool3s2_fwd", a.data_ptr(), self.stem_pooled.data_ptr(), G * self.chunk, s.hout, s.wout, 64, self.dtc)
            a = self.stem_pooled
        pooled_ready = False                         # the previous block's output pass already wrote this block's pooled input
        for bi, b in enumerate(plan.blocks):
            nxt = plan.blocks[bi + 1] if bi + 1 < len(plan.blocks) else None
            next_pool = nxt.pooled if nxt is not None else None
            a0 = a
            cur = a0
            for i, L in enumerate(b.convs):
                self._conv_bn_fwd(L, cur, G, wsets, theta, pidx)
                if i < len(b.convs) - 1:
                    self._bn_apply(L, b.mids[i], G)
                    cur = b.mids[i]
            last = b.convs[-1]
            if b.shortcut is not None:
                src = a0
                if b.pooled is not None:
                    if not pooled_ready:
                        call("fb_avgpool2_fwd", a0.data_ptr(), b.pooled.data_ptr(), G * self.chunk, b.hin, b.win, b.cin, self.dtc)
                    src = b.pooled
                self._conv_bn_fwd(b.shortcut, src, G, wsets, theta, pidx)
                pooled_ready = self._bn_apply(last, b.out, G, res=b.shortcut.x, resL=b.shortcut, pool=next_pool)
            else:
                pooled_ready = self._bn_apply(last, b.out, G, res=a0, pool=next_pool)
            a = b.out
        n = G * self.chunk
        hw = plan.h_final * plan.h_final
        call("fb_head_pool", a.data_ptr(), self.feat.data_ptr(), n, hw, plan.feat, self.dtc)
        pstride = plan.P if wsets > 1 else 0
        call("fb_head_loss", self.feat.data_ptr(), theta.data_ptr() + 4 * plan.fcw_off, theta.data_ptr() + 4 * plan.fcb_off, pstride,
             labels.data_ptr(), self.logits.data_ptr(), self.dlogits.data_ptr(), self.loss.data_ptr(), self.correct.data_ptr(), G,
             self.chunk, plan.feat, plan.classes,
             # the training loss of get_loss_fn; evaluation always uses plain cross entropy (reference training.py:345)
             0.0 if getattr(self, "_eval", False) else float(self.label_smoothing), 0 if getattr(self, "_eval", False) else int(self.only_incorrect))
        return a

    # ----------------------------------------------------------------------------------------------------- backward --
    def _bn_bwd(self, L, dout, mask, G, gout, pidx, want_dy, reduced=False, apply=True):
        """BN (+ReLU mask) backward.  Returns (dx, dy_or_None); dgamma/dbeta go to gout[g] at the layer's arena offsets.  ``reduced``: the
        input-gradient convolution that produced ``dout`` has already left the 128-pixel-block sums of dy and dy*x in ``self.stat_ws`` (its
        fused epilogue, ``_dgrad(bst=...)``), the reduction pass is skipped.  ``apply=False``: reduction and coefficients only (``L.coef``) --
        the consumer computes dx itself (``_wgrad(bn=...)``: the stem, whose input gradient nobody needs); returns (None, None)."""
        n = G * self.chunk
        px = n * L.hout * L.wout
        ppg = self.chunk * L.hout * L.wout
        bits = self.masks.get(mask.data_ptr()) if mask is not None else None      # bitmask written by the forward bn_apply
        y = None if bits is not None else mask
        dx = self.pool.get((n, L.hout, L.wout, L.cout)) if apply else None
        dy = self.pool.get((n, L.hout, L.wout, L.cout)) if want_dy else None
        if (apply and not reduced and y is None and self.bn_fused and self.f32_split != "f16x2"
                and lib.load().fb_bn_bwd_fused_supported(px, L.cout, ppg, self.dtc)):
            # one pass over (dout, x): a resident cluster of workgroups holds a chunk's operands in registers between the reduction and the
            # apply step (csrc/bn_bwd_fused.hip) -- 3 tensor passes instead of the 5 of reduce -> finalize -> apply below
            if int(lib.load().fb_ws_bn_bwd_fused_floats(px, L.cout, ppg, self.dtc)) > self.bnf_ws.numel():
                raise lib.EngineError("fb_bn_bwd_fused: partial-row scratch too small for this launch")
            call("fb_bn_bwd_fused", dout.data_ptr(), _ptr(bits), L.x.data_ptr(), self.mean_tab[pidx].data_ptr(), L.invstd.data_ptr(), L.scale.data_ptr(),
                 self.plan.ch_total, L.ch_off, gout.data_ptr() + 4 * L.g_off, gout.data_ptr() + 4 * L.b_off, self.plan.P, L.coef.data_ptr(),
                 dx.data_ptr(), _ptr(dy), px, L.cout, ppg, float(self.valid * L.hout * L.wout), self.dtc, self.bnf_ws.data_ptr(), self.bnf_sync.data_ptr())
            return dx, dy
        if reduced:
            n_mblocks = px // 128
        else:
            n_mblocks = lib.load().fb_bn_bwd_reduce_rows(px, ppg)
            call("fb_bn_bwd_reduce", dout.data_ptr(), _ptr(y), _ptr(bits), L.x.data_ptr(), self.mean_tab[pidx].data_ptr(), L.invstd.data_ptr(),
                 self.plan.ch_total, L.ch_off, self.stat_ws.data_ptr(), px, L.cout, ppg, self.dtc)
        call("fb_bn_bwd_finalize", self.stat_ws.data_ptr(), n_mblocks, G, L.cout, float(self.valid * L.hout * L.wout), L.scale.data_ptr(),
             self.mean_tab[pidx].data_ptr(), L.invstd.data_ptr(), self.plan.ch_total, L.ch_off,
             gout.data_ptr() + 4 * L.g_off, gout.data_ptr() + 4 * L.b_off, self.plan.P, L.coef.data_ptr(), 1 if reduced else 0)
        if not apply:
            return None, None
        call("fb_bn_bwd_apply", dout.data_ptr(), _ptr(y), _ptr(bits), L.x.data_ptr(), L.coef.data_ptr(), dx.data_ptr(), _ptr(dy), px, L.cout, ppg,
             self.dtc, *((self._amax_slot(dx, px * L.cout), self.amax_ws.data_ptr()) if self.f32_split == "f16x2" else (None, None)))
        return dx, dy

    def _bn_bwd2(self, La, Lb, dout, mask, G, gout, pidx):
        """BatchNorm backward of TWO layers that take the same incoming gradient through the same ReLU mask (conv2 and the shortcut convolution of a
        downsampling block): one read of ``dout`` per pass for both (``fb_bn_bwd_reduce2`` / ``fb_bn_bwd_apply2``; bit-identical to two
        ``_bn_bwd`` calls).  Returns (dx_a, dx_b)."""
        n = G * self.chunk
        px, ppg = n * La.hout * La.wout, self.chunk * La.hout * La.wout
        bits = self.masks[mask.data_ptr()]
        rows = lib.load().fb_bn_bwd_reduce_rows(px, ppg)
        half = 2 * rows * La.cout                            # floats of one layer's partial rows
        pa, pb = self.stat_ws.data_ptr(), self.stat_ws.data_ptr() + 4 * half
        call("fb_bn_bwd_reduce2", dout.data_ptr(), bits.data_ptr(), La.x.data_ptr(), La.invstd.data_ptr(), La.ch_off, pa, Lb.x.data_ptr(), Lb.invstd.data_ptr(),
             Lb.ch_off, pb, self.mean_tab[pidx].data_ptr(), self.plan.ch_total, px, La.cout, ppg, self.dtc)
        for L, part in ((La, pa), (Lb, pb)):
            call("fb_bn_bwd_finalize", part, rows, G, L.cout, float(self.valid * L.hout * L.wout), L.scale.data_ptr(), self.mean_tab[pidx].data_ptr(),
                 L.invstd.data_ptr(), self.plan.ch_total, L.ch_off, gout.data_ptr() + 4 * L.g_off, gout.data_ptr() + 4 * L.b_off, self.plan.P, L.coef.data_ptr(), 0)
        dxa, dxb = self.pool.get((n, La.hout, La.wout, La.cout)), self.pool.get((n, La.hout, La.wout, La.cout))
        call("fb_bn_bwd_apply2", dout.data_ptr(), bits.data_ptr(), La.x.data_ptr(), La.coef.data_ptr(), dxa.data_ptr(), Lb.x.data_ptr(), Lb.coef.data_ptr(),
             dxb.data_ptr(), px, La.cout, ppg, self.dtc)
        return dxa, dxb

    def _bn_bwd2_ok(self, La, Lb, G):
        if os.environ.get("FB_BN_BWD_DUAL", "1") == "0" or self.f32_split == "f16x2" or La.cout != Lb.cout or (La.hout, La.wout) != (Lb.hout, Lb.wout):
            return False
        vec = 16 // torch.empty((), dtype=self.dt).element_size()
        n = G * self.chunk
        rows = lib.load().fb_bn_bwd_reduce_rows(n * La.hout * La.wout, self.chunk * La.hout * La.wout)
        return La.cout % vec == 0 and 256 % (La.cout // vec) == 0 and 4 * rows * La.cout <= self.stat_ws.numel()

    def _wgrad_bn_ok(self, L, mask):
        """Can the weight gradient of L apply the BatchNorm backward of its own output itself (``fb_wgrad_args.bn_x``: no dx tensor)?"""
        if os.environ.get("FB_WGRAD_BNF", "1") == "0" or self.f32_split == "f16x2" or self.masks.get(mask.data_ptr()) is None:
            return False                             # (fp16x2 planes: the scale of dx needs the materialised tensor)
        a = lib.WgradArgs(None, None, None, self.chunk, L.hin, L.win, L.cin_pad, L.hout, L.wout, L.cout, L.R, L.S, L.stride, L.pad, self.chunk, 1, self.dtc, 0)
        return bool(lib.load().fb_wgrad_bn_fused_supported(lib.C.byref(a)))

    def _wgrad(self, L, src, dx, G, gout, bn=None):
        """Weight gradient of layer L into gout[g] (per chunk).  Runs on the weight-gradient stream; the event of its completion
        is kept in ``self._wgrad_event`` (callers hand it to the pool with ``dx`` and wait for it at the end of backward).
        ``bn = (dout, mask_act)``: ``dx`` is not materialised -- the kernel computes it in its loader from the gradient w.r.t. the BatchNorm +
        ReLU output, the layer's conv output ``L.x``, the ReLU bitmask and ``L.coef`` (``_bn_bwd(apply=False)`` has just written them)."""
        n = G * self.chunk
        if bn is not None:
            dx = bn[0]
        if self.chain_on and bn is None and any(L is c for c in self.chain_layers):
            # the group's SUM into gsum[w_off:] + the chunks' sums of squares into L.chain_sq (no arena rows for this layer)
            S = min(L.n_chains, G)
            a = lib.WgradArgs(src.data_ptr(), dx.data_ptr(), None, n, L.hin, L.win, L.cin_pad, L.hout, L.wout, L.cout, L.R, L.S, L.stride, L.pad,
                              self.chunk, 1, self.dtc, 0)

            def launch_chain():
                call("fb_conv2d_wgrad_chain", lib.C.byref(a), S, self.chain_ws.data_ptr(), L.chain_sq.data_ptr())
                call("fb_wgrad_reduce", self.chain_ws.data_ptr(), self.gsum.data_ptr() + 4 * L.w_off, 0, 1, S, L.cout, L.taps, L.cin_pad, L.cin_real)

            if self.wstream is None:
                launch_chain()
                return
            ready = self.events.record()
            with torch.cuda.stream(self.wstream):
                self.events.wait(ready)
                launch_chain()
                self._wgrad_event = self.events.record()
            return
        # one K slice and no channel padding: the kernel writes the per-chunk gradients straight into the arena rows
        direct = L.split_k == 1 and L.cin_pad == L.cin_real
        am_x = am_dy = None
        if self.f32_split == "f16x2":                # (computed on the main stream, before the event the weight-gradient stream waits for)
            am_x, am_dy = self._amax(src, n * L.hin * L.win * L.cin_pad, G), self._amax(dx, n * L.hout * L.wout * L.cout, G)
        ranges = self._stem_ranges(G) if L is self.plan.stem else [(0, G)]
        es = src.element_size()
        args = []
        for g0, g_n in ranges:                       # (more than one range: only the stem of a group whose patches exceed 2^31 bytes)
            i0 = g0 * self.chunk
            args.append((g0, g_n, lib.WgradArgs(src.data_ptr() + i0 * L.hin * L.win * L.cin_pad * es, dx.data_ptr() + i0 * L.hout * L.wout * L.cout * es,
                                                gout.data_ptr() + 4 * (L.w_off + g0 * self.plan.P) if direct else self.slab_ws.data_ptr(), g_n * self.chunk,
                                                L.hin, L.win, L.cin_pad, L.hout, L.wout, L.cout, L.R, L.S, L.stride, L.pad, self.chunk, L.split_k, self.dtc,
                                                self.plan.P if direct else 0, am_x + 4 * g0 if am_x is not None else None,
                                                am_dy + 4 * g0 if am_dy is not None else None,
                                                *((L.x.data_ptr() + i0 * L.hout * L.wout * L.cout * es, self.masks[bn[1].data_ptr()].data_ptr() + i0 * L.hout * L.wout * L.cout * es // 16,
                                                   L.coef.data_ptr() + 4 * g0 * L.cout * 3) if bn is not None else (None, None, None)))))

        def launch():
            for g0, g_n, a in args:
                call("fb_conv2d_wgrad", lib.C.byref(a))
                if not direct:
                    call("fb_wgrad_reduce", self.slab_ws.data_ptr(), gout.data_ptr() + 4 * (L.w_off + g0 * self.plan.P), self.plan.P, g_n, L.split_k, L.cout,
                         L.taps, L.cin_pad, L.cin_real)

        if self.wstream is None:
            launch()
            return
        ready = self.events.record()
        with torch.cuda.stream(self.wstream):
            self.events.wait(ready)
            launch()
            self._wgrad_event = self.events.record()

    def _dgrad_args(self, L, src, wptr, dst, addend, G, wsets, addend_mode, addend_mask=None, stat=None, bst_x=None, bst_mask=None, amax=(None, None)):
        n = G * self.chunk
        return lib.ConvArgs(src, wptr, dst, addend, stat, n, L.hout, L.wout, L.cout, L.hin, L.win, L.cin_pad,
                            L.R, L.S, L.stride, L.pad, 1, self.chunk if wsets > 1 else n, self.plan.wc_total if wsets > 1 else 0,
                            addend_mode, self.dtc, addend_mask, bst_x, bst_mask, amax[0], amax[1], self.chunk)

    def _dgrad(self, L, dx, G, wsets, addend=None, addend_mode=0, addend_mask=None, bst=None):
        """Input gradient of layer L.  ``bst = (Lbn, out)``: the BatchNorm (layer ``Lbn``, output tensor ``out``) whose backward consumes the
        result; where the kernel can (fb_conv_bwd_stat_supported) it also leaves that BatchNorm's reduction sums in ``self.stat_ws`` and
        ``self.bst_done`` is set for the following ``_bn_bwd(reduced=...)``: one pass over (dout, x) less."""
        n = G * self.chunk
        out = self.pool.get((n, L.hin, L.win, L.cin_pad))
        wd = self.w_dgrad[1 if wsets > 1 else 0]
        wptr = wd.data_ptr() + wd.element_size() * L.wc_off
        self.bst_done = False
        a = None
        if bst is not None and self.fuse_bwd_stat:
            bits = self.masks.get(bst[1].data_ptr())
            if bits is not None and bst[0].cout == L.cin_pad:
                a = self._dgrad_args(L, dx.data_ptr(), wptr, out.data_ptr(), _ptr(addend), G, wsets, addend_mode, _ptr(addend_mask),
                                     self.stat_ws.data_ptr(), bst[0].x.data_ptr(), bits.data_ptr())
                self.bst_done = bool(lib.load().fb_conv_bwd_stat_supported(lib.C.byref(a)))
        if not self.bst_done:
            a = self._dgrad_args(L, dx.data_ptr(), wptr, out.data_ptr(), _ptr(addend), G, wsets, addend_mode, _ptr(addend_mask),
                                 amax=self._amax_pair(L, dx, n * L.hout * L.wout * L.cout, wsets, G))
        call("fb_conv2d", lib.C.byref(a))
        return out

    def _masked_addend_ok(self, L, G, wsets):
        """Can the input-gradient convolution of L add the residual-branch gradient through its ReLU bitmask (``addend_mask``)?  Then
        the BN backward above it does not have to write the masked copy ``dy`` (2 bytes per element of an HBM-bound pass)."""
        if os.environ.get("FB_MASKED_ADDEND", "1") == "0":
            return False
        one = self.theta.data_ptr()                  # any non-null pointers: host-side check only
        a = self._dgrad_args(L, one, one, one, one, G, wsets, 1, one)
        return bool(lib.load().fb_conv_masked_addend_supported(lib.C.byref(a)))

    def backward(self, patches, G, wsets, theta, gout, pidx, on_block_done=None):
        """Explicit backward through the DAG; per-chunk gradients are written to ``gout[g]`` (shape [G, P]).  ``on_block_done(bi)`` is
        called when every launch that writes the gradients of block ``bi`` (and of everything after it) has been queued."""
        plan, pool = self.plan, self.pool
        n = G * self.chunk
        hw = plan.h_final * plan.h_final
        pstride = plan.P if wsets > 1 else 0
        d = pool.get((n, plan.h_final, plan.h_final, plan.feat))
        call("fb_head_bwd", self.feat.data_ptr(), self.dlogits.data_ptr(), theta.data_ptr() + 4 * plan.fcw_off, pstride,
             gout.data_ptr() + 4 * plan.fcw_off, gout.data_ptr() + 4 * plan.fcb_off, plan.P, d.data_ptr(), G, self.chunk, hw, plan.feat,
             plan.classes, self.dtc)
        d_reduced = False                                # has the producer of ``d`` already reduced it for the BatchNorm that consumes it?
        for bi in range(len(plan.blocks) - 1, -1, -1):
            b = plan.blocks[bi]
            a0 = plan.blocks[bi - 1].out if bi > 0 else (self.stem_pooled if plan.stem_pool else self.stem_out)
            last = b.convs[-1]
            first = b.convs[0]
            # dy = d * (out > 0) is the gradient entering the residual branch.  It is only materialised where its consumer cannot apply
            # the ReLU bitmask of ``b.out`` itself: the shortcut's BN backward always can, the identity branch's input-gradient
            # convolution where fb_conv_masked_addend_supported says so.
            out_bits = self.masks.get(b.out.data_ptr())
            lazy = out_bits is not None and (b.shortcut is not None or self._masked_addend_ok(first, G, wsets))
            dxs_early = None
            if lazy and b.shortcut is not None and not d_reduced and self._bn_bwd2_ok(last, b.shortcut, G):
                dx, dxs_early = self._bn_bwd2(last, b.shortcut, d, b.out, G, gout, pidx)          # conv2's and the shortcut's BatchNorm: one read of d each pass
                dy = None
            else:
                dx, dy = self._bn_bwd(last, d, b.out, G, gout, pidx, want_dy=not lazy, reduced=d_reduced)
            # the BatchNorm that consumes this block's input gradient: the last one of the previous block, or the stem's
            if bi > 0:
                consumer = (plan.blocks[bi - 1].convs[-1], plan.blocks[bi - 1].out)
            else:
                consumer = None if plan.stem_pool else (plan.stem, self.stem_out)
            if not lazy:
                pool.put(d)
            srcs = [a0] + b.mids
            cur_dx = dx
            for i in range(len(b.convs) - 1, -1, -1):
                L = b.convs[i]
                self._wgrad(L, srcs[i], cur_dx, G, gout)
                ev_cur = self._wgrad_event
                if i > 0:
                    d_mid = self._dgrad(L, cur_dx, G, wsets, bst=(b.convs[i - 1], b.mids[i - 1]))
                    pool.put(cur_dx, event=ev_cur)
                    cur_dx, _ = self._bn_bwd(b.convs[i - 1], d_mid, b.mids[i - 1], G, gout, pidx, want_dy=False, reduced=self.bst_done)
                    pool.put(d_mid)
            if b.shortcut is not None:
                S = b.shortcut
                if dxs_early is not None:
                    dxs = dxs_early
                else:
                    dxs, _ = self._bn_bwd(S, d, b.out, G, gout, pidx, want_dy=False) if lazy else self._bn_bwd(S, dy, None, G, gout, pidx, want_dy=False)
                src = b.pooled if b.pooled is not None else a0
                self._wgrad(S, src, dxs, G, gout)
                d_p = self._dgrad(S, dxs, G, wsets)
                pool.put(dxs, event=self._wgrad_event)
                d_in = self._dgrad(first, cur_dx, G, wsets, addend=d_p, addend_mode=2 if b.pooled is not None else 1, bst=consumer)
                pool.put(d_p)
            elif lazy:
                d_in = self._dgrad(first, cur_dx, G, wsets, addend=d, addend_mode=1, addend_mask=out_bits, bst=consumer)
            else:
                d_in = self._dgrad(first, cur_dx, G, wsets, addend=dy, addend_mode=1, bst=consumer)
            d_reduced = self.bst_done
            pool.put(cur_dx, event=ev_cur)
            pool.put(d if lazy else dy)
            d = d_in
            if on_block_done is not None:
                on_block_done(bi)
        S = plan.stem
        if plan.stem_pool:
            d_r = pool.get((n, S.hout, S.wout, 64))
            if self.stem_pool_idx is not None:
                call("fb_maxpool3s2_bwd_idx", self.stem_pool_idx.data_ptr(), d.data_ptr(), d_r.data_ptr(), n, S.hout, S.wout, 64, self.dtc)
            else:
                call("fb_maxpool3s2_bwd", self.stem_out.data_ptr(), d.data_ptr(), d_r.data_ptr(), n, S.hout, S.wout, 64, self.dtc)
            pool.put(d)
            d = d_r
        if self._wgrad_bn_ok(S, self.stem_out) and not (d_reduced and not plan.stem_pool):
            # the stem's input gradient is needed by nobody: its weight gradient applies the BatchNorm backward in its own loader (no dx tensor:
            # one write and one read of the largest activation less per group)
            self._bn_bwd(S, d, self.stem_out, G, gout, pidx, want_dy=False, apply=False)
            self._wgrad(S, patches, None, G, gout, bn=(d, self.stem_out))
            pool.put(d, event=self._wgrad_event)
        else:
            dx, _ = self._bn_bwd(S, d, self.stem_out, G, gout, pidx, want_dy=False, reduced=d_reduced and not plan.stem_pool)
            pool.put(d)
            self._wgrad(S, patches, dx, G, gout)
            pool.put(dx, event=self._wgrad_event)
        if self.wstream is not None:                     # gout is complete (and the activations are free) after this point
            self.events.wait(self.events.record(self.wstream))

    # ------------------------------------------------------------------------------------------- chunk-group gradient --
    def _replayable(self, key, body):
        """Run ``body`` (a static sequence of library launches and event operations that depends on nothing but ``key`` and the
        engine's construction) -- through the interpreter the first time, recording it; as one native replay of the recorded list afterwards."""
        if not self.use_replay:
            return body()
        streams = [torch.cuda.current_stream(), self.wstream]
        # (the list holds stream INDICES: a list recorded with a weight-gradient stream must not be replayed without one, or on another pair)
        key = key + (streams[0].cuda_stream, None if self.wstream is None else self.wstream.cuda_stream)
        cl = self.cmdlists.pop(key, None)
        if cl is not None:
            self.cmdlists[key] = cl                      # most recently used last
            cl.replay(streams)
            self.replays += 1
            # hits pay the eviction count back: a transient phase of keys that never repeat (evaluation over varying shapes, another engine's
            # buffers) must not switch recording off for the steady-state groups that follow it
            if self.cmd_evictions:
                self.cmd_evictions -= 1
                if not self.record_new and self.cmd_evictions <= self.MAX_CMDLISTS // 2:
                    self.record_new = True
            return
        if not self.record_new:
            # ... and a cache full of lists nobody replays any more never hits: after a cache worth of interpreted runs recording is tried again
            self.unrecorded_runs += 1
            self._off_misses += 1
            if self._off_misses < self.MAX_CMDLISTS:
                return body()
            self.record_new, self._off_misses, self.cmd_evictions = True, 0, self.MAX_CMDLISTS // 2
        if len(self.cmdlists) >= self.MAX_CMDLISTS:
            # Keys that never repeat (buffers re-allocated every step) or a working set beyond the cache: the least recently used list goes, its
            # event ids are handed to the next recordings -- and once a whole cache worth of lists has been dropped recording stops paying
            # (cyclic access over more keys than slots misses every time): misses run through the interpreter from then on, hits still replay
            old = self.cmdlists.pop(next(iter(self.cmdlists)))
            self.events.release(old.events)
            self.cmd_evictions += 1
            if self.cmd_evictions >= self.MAX_CMDLISTS:
                self.record_new = False
                if not self._warned_record_off:
                    self._warned_record_off = True
                    warnings.warn(f"fullbatchtraining_amd.Engine: {self.cmd_evictions} command lists evicted without a replay in between -- new launch "
                                  "sequences run through the interpreter until recorded ones are replayed again (FB_MAX_CMDLISTS raises the cache size)")
        with lib.Recorder(streams) as rec:
            body()
        self.cmdlists[key] = rec.finish()

    def group_gradient(self, patches, labels, G, gout, wsets=1, theta=None, pidx=0, on_block_done=None):
        """fwd + bwd for ``G`` chunks: per-chunk raw gradients in gout[:G], losses/corrects in self.loss/self.correct."""
        theta = self.theta if theta is None else theta

        def body():
            self.forward(patches, labels, G, wsets, theta, pidx)
            self.backward(patches, G, wsets, theta, gout, pidx, on_block_done)

        if on_block_done is not None or getattr(self, "_eval", False):
            return body()                            # (a host callback inside the sequence: the multi-GPU late bucket)
        self._replayable(("group", patches.data_ptr(), labels.data_ptr(), G, gout.data_ptr(), wsets, theta.data_ptr(), pidx, self.chunk, self.valid,
                          float(self.label_smoothing), bool(self.only_incorrect), self.fuse_bwd_stat, self.chain_on), body)

    def choose_schedule(self, patches, labels, n_chunks):
        """The stream choice of a wide Bottleneck net NOW, if it is still pending (otherwise the first full_gradient call makes it): a caller that times its first
        step (bench.py with --warmup 0) keeps the four extra group passes out of it.  ``patches`` / ``labels``: the rank's chunks as full_gradient takes them."""
        if self.stream_autotune and (n_chunks > 0 or (torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1)):
            self.stream_autotune = False
            g_n = min(self.G, n_chunks)
            self._autotune_streams(patches[:g_n * self.chunk], labels[:g_n * self.chunk], g_n)

    def _autotune_streams(self, xb, yb, g_n):
        """Times one chunk group (forward + backward into ``self.g``, which the caller's first group overwrites) with the weight gradients on their own
        stream and with everything on one stream -- the MEDIAN of three passes each, alternating (the measured effects are 1-1.5 %, one sample's noise is of that
        size) -- and keeps the faster schedule (two streams unless one is at least 0.5 % faster).  In a job of several ranks rank 0's times decide for all (the two
        schedules give the same bits, but per-rank choices would make step times and the reported ``streams`` depend on noise).  The group pass has no side
        effects beyond its own buffers: running statistics and the mean gradient are updated by full_gradient, not here."""
        two, samples = self.wstream, {"two": [0.0] * 3, "one": [0.0] * 3}
        if g_n > 0:
            samples = {"two": [], "one": []}
        for rep in range(4 if g_n > 0 else 0):           # (pass 0 records the launch sequences / warms the caches; a rank without chunks only takes rank 0's answer)
            for label, ws in (("two", two), ("one", None)):
                self.wstream = ws
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
                self.group_gradient(xb, yb, g_n, self.g, 1, self.theta, 0)
                torch.cuda.synchronize(self.device)
                if rep > 0:
                    samples[label].append(time.perf_counter() - t0)
        times = {k: sorted(v)[1] for k, v in samples.items()}
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            t = torch.tensor([times["two"], times["one"]], dtype=torch.float64,
                             device=self.device if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.broadcast(t, 0)
            times = {"two": float(t[0]), "one": float(t[1])}
        self.stream_times = times
        self.wstream = None if times["one"] < 0.995 * times["two"] else two

    def _fold(self, gbuf, g_n, lo, hi, counter, sq_out, ws, seg_row=0):
        """The running mean over [lo, hi) of the arena advanced by the ``g_n`` chunks in ``gbuf`` (+ their squared norms over that range into
        ``sq_out[:g_n]``): fb_mt_accumulate; with chained weight gradients the chained layers from their group sum (fb_mt_accumulate_sum), the
        rest per chunk with those ranges left alone (fb_mt_accumulate_skip), the norms as the sum of the two kinds of parts.  ``seg_row``: the
        caller's own row of ``self.sq_seg`` (two folds of one step run on different streams with no order between them: the late bucket's on the
        side stream, the early range's on the main stream -- each needs scratch of its own)."""
        P = self.plan.P
        if not self.chain_on:
            call("fb_mt_accumulate", self.avg.data_ptr() + 4 * lo, gbuf.data_ptr() + 4 * lo, P, g_n, hi - lo, counter, _ptr(sq_out), ws.data_ptr())
            return
        parts, skips = [], []
        for L in self.chain_layers:
            size = L.cout * L.taps * L.cin_real
            if lo <= L.w_off and L.w_off + size <= hi:
                call("fb_mt_accumulate_sum", self.avg.data_ptr() + 4 * L.w_off, self.gsum.data_ptr() + 4 * L.w_off, size, counter, g_n)
                parts.append(L.chain_sq[:g_n].sum(1))
                skips += [L.w_off - lo, L.w_off - lo + size]
            elif L.w_off < hi and L.w_off + size > lo:
                raise lib.EngineError("a chained layer straddles the bucket boundary")
        if len(skips) > 8:
            raise lib.EngineError("more than four chained layers in one range")
        skips += [0] * (8 - len(skips))
        call("fb_mt_accumulate_skip", self.avg.data_ptr() + 4 * lo, gbuf.data_ptr() + 4 * lo, P, g_n, hi - lo, counter,
             self.sq_seg[seg_row].data_ptr() if sq_out is not None else None, ws.data_ptr(), *skips)
        parts.append(self.sq_seg[seg_row, :g_n])
        if sq_out is not None:
            sq_out[:g_n].copy_(torch.stack(parts).sum(0))

    # --------------------------------------------------------------------------------------- full-batch gradient + step --
    def full_gradient(self, patches, labels, lr, block_strength=0.0, eps=1e-2, implementation="forward-differences",
                      chunk_ids=None, counter0=0, acc_strength=0.0, after_pre_pass=None, pre_block=None, batch_clip=None, late_bucket=None):
        """Accumulate the regularised gradient over chunks (reference training.py:144-174) into ``self.avg``.

        ``patches``/``labels`` hold the whole resident dataset; chunk k = rows [k*chunk, (k+1)*chunk).  ``chunk_ids``
        must be a contiguous range (this rank's shard).  Returns device tensors (loss_k, correct_k, n_k) for the chunks.
        ``after_pre_pass``: called once ``self.pre`` (the local mean of the ``acc_strength`` pre-pass) is complete -- the multi-GPU
        path turns it into the global mean there.  ``pre_block``: images per BN batch of the pre-pass when it differs from the
        chunk size (the reference's pre-pass runs whole loader blocks, its main loop ``sub_batch`` chunks; a multiple of ``chunk``).
        ``batch_clip``: ``hyp.batch_clip`` -- every (regularised) chunk gradient is clipped to this L2 norm before it enters the running
        mean (reference training.py:166-167, _clip_gradient_list training/utils.py:4-19; also the pre-pass blocks, :138-139);
        ``self.clipped_all`` then holds the per-chunk 0/1 flags of the main loop.
        ``late_bucket = (b, started)``: multi-GPU overlap of the exchange with the backward pass.  As soon as the LAST backward pass of the
        LAST chunk group has left the last stage, the slice [b, P) of the running mean (last stage + classifier, 3/4 of ResNet-18's
        parameters) is completed on a side stream and ``started()`` is called there (the caller starts that bucket's reduce-scatter);
        the slice [0, b) follows at the end as usual.
        """
        chunk, P, G = self.chunk, self.plan.P, self.G
        n_chunks = patches.shape[0] // chunk if chunk_ids is None else len(chunk_ids)
        k_first = 0 if chunk_ids is None else chunk_ids[0]
        f32 = dict(device=self.device, dtype=torch.float32)
        loss_all, correct_all, sq_all = torch.empty(n_chunks, **f32), torch.empty(n_chunks, **f32), torch.empty(n_chunks, **f32)
        self.clipped_all = torch.zeros(n_chunks, **f32) if batch_clip is not None else None
        if batch_clip is not None and getattr(self, "clipped", None) is None:
            self.clipped = torch.zeros(self.G, **f32)
        fd = block_strength != 0 or acc_strength != 0          # GradRegularizer.__init__, modules.py:150-152
        if fd:
            if implementation not in ("forward-differences", "forward-differences-legacy", "central-differences"):
                raise NotImplementedError(f"grad_reg.implementation={implementation!r} needs double backward; the engine implements the "
                                          "finite-difference variants only")
            need = 2 if implementation == "central-differences" else 1
            if self.fd_sets < need:
                raise lib.EngineError(f"Engine was built with fd_sets={self.fd_sets}; {implementation} needs {need}")
        if counter0 == 0:
            self.avg.zero_()
        self.prep_weights(self.theta, 1)
        pre = None
        if acc_strength != 0:
            # pre-pass (reference training.py:128-142): the plain full-batch gradient at theta, needed as a whole before the first
            # chunk is regularised.  It is a train-mode pass of its own: the BN running statistics take one more update per chunk
            if getattr(self, "pre", None) is None:
                self.pre = torch.zeros(P, **f32)
            pre = self.pre
            pre.zero_()
            bsz = chunk if pre_block is None else int(pre_block)          # images per BN batch of the pre-pass
            if bsz != chunk and self.valid != chunk:
                raise NotImplementedError("acc_strength pre-pass over blocks of several sub_batch chunks with a padded (ragged) chunk size")
            per = bsz // chunk
            if bsz % chunk != 0 or n_chunks % per != 0 or per > G:
                raise lib.EngineError(f"pre-pass blocks of {bsz} images do not tile {n_chunks} chunks of {chunk} (group {G})")
            n_batches, g_cap = n_chunks // per, G // per
            saved_valid = self.valid
            self.chunk = bsz                                             # (as evaluate_batch does: all per-batch sizes follow self.chunk)
            self.valid = bsz if bsz != chunk else saved_valid
            try:
                done = 0
                while done < n_batches:
                    g_n = min(g_cap, n_batches - done)
                    lo = k_first * chunk + done * bsz
                    self.group_gradient(patches[lo:lo + g_n * bsz], labels[lo:lo + g_n * bsz], g_n, self.g, 1, self.theta, 0)
                    if batch_clip is not None:
                        call("fb_mt_sqnorm", self.g.data_ptr(), P, g_n, P, 1.0, None, 0.0, self.sq.data_ptr(), self.mt_ws.data_ptr())
                        call("fb_mt_chunk_clip", self.g.data_ptr(), P, g_n, P, self.sq.data_ptr(), float(batch_clip), None)
                    call("fb_mt_accumulate", pre.data_ptr(), self.g.data_ptr(), P, g_n, P, done, None, self.mt_ws.data_ptr())
                    call("fb_bn_running_update", self.running_mean.data_ptr(), self.running_var.data_ptr(), self.mean_tab.data_ptr(),
                         self.var_tab.data_ptr(), 1, self.G * self.plan.ch_total, self._unbias_for(self.valid).data_ptr(), g_n,
                         self.plan.ch_total, BN_MOMENTUM)
                    self.num_batches_tracked += g_n
                    done += g_n
            finally:
                self.chunk, self.valid = chunk, saved_valid
            if after_pre_pass is not None:
                after_pre_pass()
        # With several groups per step the running-mean pass of group k (HBM-bound, 2 x G x 45 MB) runs on the weight-gradient stream
        # beside the forward convolutions of group k+1; the per-chunk gradients then alternate between two arenas (the main stream
        # writes dgamma / dbeta / fc gradients of group k+1 while group k is still being folded in).
        if self.stream_autotune and (n_chunks > 0 or (torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1)):
            self.stream_autotune = False
            g_n = min(G, n_chunks)
            self._autotune_streams(patches[k_first * chunk:(k_first + g_n) * chunk], labels[k_first * chunk:(k_first + g_n) * chunk], g_n)
        overlap = (not fd) and batch_clip is None and self.wstream is not None and n_chunks > G and os.environ.get("FB_ACC_OVERLAP", "1") != "0"
        if overlap and getattr(self, "g_alt", None) is None:
            self.g_alt, self.acc_ws = torch.zeros_like(self.g), torch.zeros_like(self.mt_ws)
        done, group_idx = 0, 0
        # chained weight gradients: only where nothing needs a chunk's gradient itself (no regulariser, no per-chunk clip)
        self.chain_on = bool(self.chain_layers) and not fd and batch_clip is None
        if late_bucket is not None and getattr(self, "side", None) is None:
            self.side = torch.cuda.Stream(device=self.device)
            self.sq_late, self.ws_late = torch.zeros_like(self.sq), torch.zeros_like(self.mt_ws)
        try:
            while done < n_chunks:
                g_n = min(G, n_chunks - done)
                lo = (k_first + done) * chunk
                xb, yb = patches[lo:lo + g_n * chunk], labels[lo:lo + g_n * chunk]
                gbuf = self.g_alt if (overlap and group_idx & 1) else self.g
                group_idx += 1
                # early completion of the late bucket: only for the last group, and only where the running mean is folded in one piece
                early = late_bucket is not None and done + g_n == n_chunks and not overlap and batch_clip is None
                lb = late_bucket[0] if early else 0
                central = fd and implementation == "central-differences"
                legacy = fd and implementation == "forward-differences-legacy"
                cf = (lr / 4 * (block_strength if legacy else 1.0)) if fd else 0.0

                def finish_late(bi, _g_n=g_n, _done=done):
                    """Called by the last backward pass of the group: gradients of the arena slice [lb, P) are complete."""
                    if bi != self.plan.late_block:
                        return
                    ready = [torch.cuda.current_stream().record_event()]
                    if self.wstream is not None:
                        ready.append(self.wstream.record_event())
                    with torch.cuda.stream(self.side):
                        for ev in ready:
                            self.side.wait_event(ev)
                        n_late = P - lb
                        if not fd:
                            self._fold(self.g, _g_n, lb, P, counter0 + _done, self.sq_late, self.ws_late, seg_row=1)
                        else:
                            gb = self.g_fd[1] if central else self.g
                            call("fb_mt_fd_combine_accumulate", self.avg.data_ptr() + 4 * lb, self.g.data_ptr() + 4 * lb, self.g_fd[0].data_ptr() + 4 * lb,
                                 gb.data_ptr() + 4 * lb, P, _g_n, n_late, self.eps_n.data_ptr(), cf, counter0 + _done)
                        late_bucket[1]()

                hook = finish_late if early else None
                self.group_gradient(xb, yb, g_n, gbuf, 1, self.theta, 0, on_block_done=hook if not fd else None)
                loss_all[done:done + g_n].copy_(self.loss[:g_n])
                correct_all[done:done + g_n].copy_(self.correct[:g_n])
                n_passes = 1
                if overlap:
                    ready = torch.cuda.current_stream().record_event()
                    with torch.cuda.stream(self.wstream):
                        self.wstream.wait_event(ready)
                        self._fold(gbuf, g_n, 0, P, counter0 + done, sq_all[done:done + g_n], self.acc_ws)
                elif not fd and batch_clip is not None:
                    call("fb_mt_sqnorm", self.g.data_ptr(), P, g_n, P, 1.0, None, 0.0, self.sq.data_ptr(), self.mt_ws.data_ptr())
                    call("fb_mt_chunk_clip", self.g.data_ptr(), P, g_n, P, self.sq.data_ptr(), float(batch_clip), self.clipped.data_ptr())
                    call("fb_mt_accumulate", self.avg.data_ptr(), self.g.data_ptr(), P, g_n, P, counter0 + done, None, self.mt_ws.data_ptr())
                elif not fd:
                    # (with an early late bucket the slice [lb, P) has been folded on the side stream; |g_k|^2 is the sum of the two parts)
                    self._fold(self.g, g_n, 0, lb if early else P, counter0 + done, self.sq, self.mt_ws)
                    if early:
                        torch.cuda.current_stream().wait_stream(self.side)
                        self.sq[:g_n].add_(self.sq_late[:g_n])
                else:
                    s = 1.0 if legacy else float(block_strength)
                    # finite-difference direction v = s*g_k + acc*pre (modules.py:217-221; the legacy variant ignores pre, :243-245)
                    vpre, vacc = (None, 0.0) if (legacy or pre is None) else (pre.data_ptr(), float(acc_strength))
                    call("fb_mt_sqnorm", self.g.data_ptr(), P, g_n, P, 1.0, None, 0.0, self.sq.data_ptr(), self.mt_ws.data_ptr())
                    call("fb_mt_sqnorm", self.g.data_ptr(), P, g_n, P, s, vpre, vacc, self.vnorm2.data_ptr(), self.mt_ws.data_ptr())
                    call("fb_mt_fd_perturb", self.theta.data_ptr(), self.g.data_ptr(), P, g_n, P, s, float(eps), 0.5 if central else 1.0,
                         self.vnorm2.data_ptr(), self.eps_n.data_ptr(), vpre, vacc, self.theta_k.data_ptr())
                    self.prep_weights(self.theta_k, g_n, per_chunk=True)
                    self.group_gradient(xb, yb, g_n, self.g_fd[0], 2, self.theta_k, 1, on_block_done=None if central else hook)
                    n_passes = 2
                    if central:
                        call("fb_mt_fd_perturb", self.theta.data_ptr(), self.g.data_ptr(), P, g_n, P, s, float(eps), -0.5,
                             self.vnorm2.data_ptr(), self.eps_n.data_ptr(), vpre, vacc, self.theta_k.data_ptr())
                        self.prep_weights(self.theta_k, g_n, per_chunk=True)
                        self.group_gradient(xb, yb, g_n, self.g_fd[1], 2, self.theta_k, 2, on_block_done=hook)
                        n_passes = 3
                    gb = self.g_fd[1] if central else self.g                 # vhp = (g(theta+) - g(theta-)) / eps_n  or  (g(theta+) - g) / eps_n
                    if batch_clip is None:
                        call("fb_mt_fd_combine_accumulate", self.avg.data_ptr(), self.g.data_ptr(), self.g_fd[0].data_ptr(), gb.data_ptr(), P, g_n,
                             lb if early else P, self.eps_n.data_ptr(), cf, counter0 + done)
                        if early:
                            torch.cuda.current_stream().wait_stream(self.side)
                    else:      # the regularised chunk gradients are materialised, clipped one by one, then averaged
                        call("fb_mt_fd_combine", self.g.data_ptr(), self.g_fd[0].data_ptr(), gb.data_ptr(), P, g_n, P, self.eps_n.data_ptr(), cf)
                        call("fb_mt_sqnorm", self.g.data_ptr(), P, g_n, P, 1.0, None, 0.0, self.vnorm2.data_ptr(), self.mt_ws.data_ptr())
                        call("fb_mt_chunk_clip", self.g.data_ptr(), P, g_n, P, self.vnorm2.data_ptr(), float(batch_clip), self.clipped.data_ptr())
                        call("fb_mt_accumulate", self.avg.data_ptr(), self.g.data_ptr(), P, g_n, P, counter0 + done, None, self.mt_ws.data_ptr())
                if not overlap:
                    sq_all[done:done + g_n].copy_(self.sq[:g_n])
                if batch_clip is not None:
                    self.clipped_all[done:done + g_n].copy_(self.clipped[:g_n])
                call("fb_bn_running_update", self.running_mean.data_ptr(), self.running_var.data_ptr(), self.mean_tab.data_ptr(),
                     self.var_tab.data_ptr(), n_passes, self.G * self.plan.ch_total, self.unbias.data_ptr(), g_n, self.plan.ch_total,
                     BN_MOMENTUM)
                self.num_batches_tracked += g_n * n_passes
                done += g_n
            if overlap:
                torch.cuda.current_stream().wait_stream(self.wstream)
        finally:                                     # (an exception inside a group must not leave later group_gradient / evaluate calls chained)
            self.chain_on = False
        return loss_all, correct_all, sq_all

    def check_device_errors(self):
        """Raises if a kernel left its sticky error word (fb_bn_bwd_fused: a cluster waited ~2 s for workgroups that never became resident)."""
        if int(self.bnf_sync[-1]) != 0:
            self.bnf_sync[-1] = 0
            raise lib.EngineError("fb_bn_bwd_fused: a workgroup cluster timed out waiting for its reduction (results of that launch are invalid); "
                                  "FB_BN_BWD_FUSED=0 selects the two-pass BatchNorm backward")

    def grad_and_param_sqnorm(self):
        """Device tensor [|avg|^2, |theta|^2] (clip norm, reference training.py:202-204; param_norm, :92)."""
        call("fb_mt_norms2", self.avg.data_ptr(), self.theta.data_ptr(), self.plan.P, self.norms2.data_ptr(), self.mt_ws.data_ptr())
        return self.norms2

    def sgd_step(self, lr, weight_decay, momentum, dampening, nesterov, grad_clip=None, lo=0, n=None):
        """Clip by the global norm in ``self.norms2[0]`` and apply the Nesterov-SGD update on arena range [lo, lo+n)."""
        n = self.plan.P - lo if n is None else n
        call("fb_mt_clip_sgd", self.theta.data_ptr() + 4 * lo, self.avg.data_ptr() + 4 * lo, self.mom.data_ptr() + 4 * lo, n,
             self.norms2.data_ptr(), -1.0 if grad_clip is None else float(grad_clip), float(lr), float(weight_decay), float(momentum),
             float(dampening), 1 if nesterov else 0, 1 if self.first_step else 0)
        self.first_step = False

    def sgd_step_per_tensor(self, lr, weight_decays, momentum, dampening, nesterov, grad_clip=None):
        """``sgd_step`` with one weight decay per parameter tensor (``model.parameters()`` order): one launch per tensor range."""
        first = self.first_step
        for name, wd in zip(self.plan.param_names, weight_decays):
            self.first_step = first
            self.sgd_step(lr, wd, momentum, dampening, nesterov, grad_clip, lo=self.plan.offsets[name], n=math.prod(self.plan.param_shapes[name]))

    def apply_clip(self, grad_clip):
        """The clip of ``sgd_step`` applied to ``self.avg`` in place (needed when something acts on the clipped gradient before the
        update: gradient noise, reference training.py:205-215); uses the norm in ``self.norms2[0]``."""
        call("fb_mt_clip_scale", self.avg.data_ptr(), self.plan.P, self.norms2.data_ptr(), float(grad_clip))

    def grad_noise(self, noise_tensors, strength, multiplicative):
        """``p.grad.add_(a * noise)`` / ``p.grad.mul_(1 + m * noise)`` (reference training.py:212-215) on the averaged gradient;
        ``noise_tensors``: one tensor per parameter in ``model.parameters()`` layout (the caller draws them), or the flat arena vector."""
        flat = noise_tensors if torch.is_tensor(noise_tensors) else self.flatten([t.detach() for t in noise_tensors])
        call("fb_mt_grad_noise", self.avg.data_ptr(), flat.data_ptr(), self.plan.P, float(strength), 1 if multiplicative else 0)

    def clip_norm_inf(self):
        """``hyp.grad_clip_norm=inf`` (reference training.py:199-200): put (max|avg|)^2 into the clip-norm slot ``self.norms2[0]``."""
        call("fb_mt_absmax2", self.avg.data_ptr(), self.plan.P, self.norms2.data_ptr(), self.mt_ws.data_ptr())

    def clip_norm_p(self, p):
        """``hyp.grad_clip_norm=p`` (reference training.py:201-204): (sum |avg_i|^p)^(2/p) into the clip-norm slot."""
        call("fb_mt_pnorm2", self.avg.data_ptr(), self.plan.P, float(p), self.norms2.data_ptr(), self.mt_ws.data_ptr())

    def norm_bias(self, strength, norm_type, bias):
        """External norm bias on the averaged gradient (reference training.py:188-196); ``self.norms2[1]`` must hold |theta|^2.
        One launch per parameter tensor: the constant of norm_type 1 must not land in the arena's alignment padding."""
        pn2 = self.norms2.data_ptr() + 4
        for name in self.plan.param_names:
            off, n = self.plan.offsets[name], math.prod(self.plan.param_shapes[name])
            call("fb_mt_norm_bias", self.avg.data_ptr() + 4 * off, self.theta.data_ptr() + 4 * off, n, pn2, float(strength), float(bias),
                 int(norm_type))

    def ema_update(self, momentum):
        """``_update_ema`` (reference training/utils.py:22-29) for parameters and BN running statistics; the first call makes the
        copy the reference takes at the start of training (training.py:72-73) -- call ``ema_init`` before the first step."""
        m, om = float(momentum), float(1 - momentum)
        for ema, src in ((self.theta_ema, self.theta), (self.running_mean_ema, self.running_mean), (self.running_var_ema, self.running_var)):
            call("fb_mt_ema", ema.data_ptr(), src.data_ptr(), src.numel(), m, om)

    def ema_init(self):
        self.theta_ema, self.running_mean_ema, self.running_var_ema = self.theta.clone(), self.running_mean.clone(), self.running_var.clone()

    def swap_ema(self):
        """Exchange the live parameters / running statistics with their EMA (evaluation of the EMA model; call twice)."""
        self.theta, self.theta_ema = self.theta_ema, self.theta
        self.running_mean, self.running_mean_ema = self.running_mean_ema, self.running_mean
        self.running_var, self.running_var_ema = self.running_var_ema, self.running_var

    def sam_ascent(self, rho, grad_clip=None):
        """SAM first step (reference additional_optimizers/sam.py:56-69) on the arena: theta += e_w with e_w = rho * g / |g| of the
        (clipped) averaged gradient; needs ``self.norms2[0]`` = |avg|^2 (``grad_and_param_sqnorm``).  e_w is kept for the way back."""
        if getattr(self, "e_w", None) is None:
            self.e_w = torch.zeros_like(self.theta)
        call("fb_mt_sam_ascent", self.theta.data_ptr(), self.avg.data_ptr(), self.e_w.data_ptr(), self.plan.P, self.norms2.data_ptr(),
             -1.0 if grad_clip is None else float(grad_clip), float(rho))

    def sam_restore(self):
        """SAM second step, first half (sam.py:72-77): theta -= e_w (the same rounding as the reference's ``p.sub_(e_w)``)."""
        call("fb_mt_sam_restore", self.theta.data_ptr(), self.e_w.data_ptr(), self.plan.P)

    # ------------------------------------------------------------------------------------------------------ evaluation --
    def evaluate_batch(self, patches, labels):
        """BN in eval mode (running statistics), mean CE and #correct of one batch of <= G*chunk images
        (reference training.py:365-380).  Returns python floats (loss_mean, n_correct)."""
        n = patches.shape[0]
        if n > self.G * self.chunk:
            raise lib.EngineError("evaluate_batch: batch larger than the engine's activation buffers")
        self.prep_weights(self.theta, 1)
        for L in self.plan.layers:
            call("fb_bn_eval_coeffs", self.theta.data_ptr() + 4 * L.g_off, self.theta.data_ptr() + 4 * L.b_off,
                 self.running_mean.data_ptr() + 4 * L.ch_off, self.running_var.data_ptr() + 4 * L.ch_off, BN_EPS, L.scale.data_ptr(),
                 L.shift.data_ptr(), L.cout)
        saved = (self.chunk, self.valid)
        self.chunk, self.valid, self._eval = n, n, True
        try:
            self.forward(patches, labels, 1, 1, self.theta, 0)
        finally:
            (self.chunk, self.valid), self._eval = saved, False
        return float(self.loss[0]), float(self.correct[0])
