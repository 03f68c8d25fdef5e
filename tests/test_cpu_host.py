"""CPU-side tests: C-ABI library loads and exports every declared symbol (no compute), host logic (cfg, LR schedules,
shard plan), no-GPU behaviour is a loud failure, and the multi-GPU collective wiring on world_size-2 gloo."""
import os
import re
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as entry
    from fullbatchtraining_amd import lib

    entry.build()
    handle = lib.load()
    header = open(os.path.join(REPO, "include", "fb_engine.h")).read()
    declared = set(re.findall(r"\b(fb_[a-z0-9_]+)\s*\(", header))
    declared -= {"fb_status", "fb_dtype"}
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(handle, name), name
    assert set(lib.EXPORTS) == declared
    assert handle.fb_abi_version() == lib.EXPECTED_ABI


def test_built_kernels_keep_wide_buffer_store_data_alive():
    """Static check of the built gfx950 code (tools/hazard_scan.py): no 128-bit MUBUF store with an SGPR offset is followed within two wait
    states by a write to one of its data registers.  LLVM adds no wait state behind such stores and gfx950 needs one whenever another stream
    keeps the memory pipeline busy (profiles/r3_notes.md: the rewritten register now and then reached memory instead of the stored dword);
    the kernels hold the data registers with csrc/common.h store_b128_guard.  The scan must also still SEE such stores (the guarded ones)."""
    import importlib.util

    import __graft_entry__ as entry
    from fullbatchtraining_amd import lib

    entry.build()
    spec = importlib.util.spec_from_file_location("hazard_scan", os.path.join(REPO, "tools", "hazard_scan.py"))
    hs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hs)
    ins = hs.disassemble(lib._LIB_PATH)
    assert len({k for k, _ in ins}) > 100 and sum(1 for _, t in ins if t.startswith("buffer_store_dwordx4")) > 100
    bad, _ = hs.scan(ins)
    assert not bad, bad[:3]
    # the scanner itself: the unguarded shape is reported, a guarded one is not
    demo = [("k", "buffer_store_dwordx4 v[44:47], v82, s[44:47], s78 offen"), ("k", "v_cndmask_b32_e64 v45, v115, v90, s[10:11]")]
    assert len(hs.scan(demo)[0]) == 1
    demo.insert(1, ("k", "s_nop 3"))
    assert not hs.scan(demo)[0]


def test_asm_prefetched_loads_of_the_1x1_kernels_are_not_touched_before_a_wait():
    """The streaming / pipelined 1x1 kernels prefetch their addend (and its mask bytes) with inline-asm buffer loads that hipcc neither counts nor waits for; the
    consumer waits with a hand-counted ``s_waitcnt vmcnt(N)`` (csrc/conv1x1_pipe.hip p1_load16 / p1_wait_loads4, csrc/conv1x1_stream.hip).  To the compiler the
    destination registers are defined at the asm, so under register pressure it could copy or spill them between the load and the wait -- stale data, silently.
    Static check of the built code: these kernels use no scratch memory (no spills at all: the hand counts also assume no compiler-inserted VMEM traffic), and no
    instruction reads or writes the destination registers of such a load before the next ``s_waitcnt`` that names vmcnt."""
    import importlib.util
    import re
    import subprocess
    import tempfile

    import __graft_entry__ as entry
    from fullbatchtraining_amd import lib

    entry.build()
    spec = importlib.util.spec_from_file_location("hazard_scan", os.path.join(REPO, "tools", "hazard_scan.py"))
    hs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hs)
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(REPO, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    bdir = os.path.join(REPO, "fullbatchtraining_amd", "csrc", "build")
    with tempfile.TemporaryDirectory() as tmp:
        rows = [k for f in ("conv1x1_pipe.o", "conv1x1_stream.o") for k in kr.kernels_of(os.path.join(bdir, f), tmp)]
    assert len(rows) >= 20 and all(k["scratch"] == 0 for k in rows), [(k["name"], k["scratch"]) for k in rows if k["scratch"]]
    ins = [(k, t) for k, t in hs.disassemble(lib._LIB_PATH) if "conv1x1_pipe_kernel" in k or "conv1x1_stream_kernel" in k]
    assert not any(t.startswith("scratch_") for _, t in ins)
    loads = checked = 0
    pending, kern = {}, None                       # destination register -> index of its load
    for idx, (k, t) in enumerate(ins):
        if k != kern:
            kern, pending = k, {}
        mn = t.split()[0]
        if mn == "s_waitcnt" and "vmcnt" in t:
            checked += len(pending)
            pending = {}
            continue
        regs = set()
        for tok in re.findall(r"v\[\d+:\d+\]|v\d+", t):
            regs |= hs._regs(tok)
        hit = regs & set(pending)
        assert not hit, (k, ins[min(pending[r] for r in hit)][1], t)
        if mn in ("buffer_load_dwordx4", "buffer_load_ubyte") and " lds" not in t and "offen" in t:
            loads += 1
            for r in hs._regs(t[len(mn):].split(",")[0]):
                pending[r] = idx
    assert loads > 50 and checked > 50, (loads, checked)


def test_workspace_size_queries():
    """Host-side arithmetic of the C ABI (no launch): scratch sizes the caller has to provide."""
    from fullbatchtraining_amd import lib
    h = lib.load()
    conv = lib.ConvArgs(n_img=256, Hd=16, Wd=16, Cd=128)
    assert h.fb_ws_conv_stat_floats(lib.C.byref(conv)) == 2 * (256 * 256 // 128) * 128
    wg = lib.WgradArgs(n_img=256, imgs_per_group=128, split_k=5, Cd=128, R=3, S=3, Cs=64)
    assert h.fb_ws_wgrad_slab_floats(lib.C.byref(wg)) == 2 * 5 * 128 * 9 * 64
    assert h.fb_ws_bn_partial_floats(1000, 64) == 2 * 8 * 64
    # fb_bn_bwd_reduce: 128..1024 pixels per partial row, whole rows inside one statistics group, >= 2048 rows when possible
    assert h.fb_bn_bwd_reduce_rows(4992 * 1024, 128 * 1024) == 4992 and h.fb_bn_bwd_reduce_rows(4992 * 256, 128 * 256) == 2496
    assert h.fb_bn_bwd_reduce_rows(4992 * 16, 128 * 16) == 624 and h.fb_bn_bwd_reduce_rows(1000, 128) == 8
    assert h.fb_ws_mt_floats(1) == 2 * lib.MT_BLOCKS and h.fb_ws_mt_floats(39) == 39 * lib.MT_BLOCKS


def test_command_list_host_side():
    """Native launch executor, host side only (no launch): every recordable entry point resolves, argument words are packed as the C side
    unpacks them (one 64-bit word per argument, floats / doubles as bit patterns, the argument struct copied), malformed commands are refused."""
    import ctypes as C
    import struct

    from fullbatchtraining_amd import lib
    h = lib.load()
    for name in ("fb_conv2d", "fb_conv2d_wgrad", "fb_bn_apply", "fb_bn_bwd_apply", "fb_head_loss", "fb_mt_accumulate", "fb_weight_prep", "fb_absmax"):
        fn = h.fb_cmd_fn_id(name.encode())
        assert fn >= 0, name
        assert h.fb_cmd_fn_nargs(fn) == len(lib._SIGS[name]), name          # (including the stream)
    assert h.fb_cmd_fn_id(b"fb_stem_patches") == -1                         # takes a HOST array: cannot be recorded
    a = lib.ConvArgs(n_img=7, Hs=3)
    words, blob = lib._pack_words("fb_conv2d", (C.byref(a),))
    assert blob is a and len(words) == 1
    words, blob = lib._pack_words("fb_bn_fwd_finalize", (0x1234, 5, -1, 64, 2.5, None, 8, 9, 1e-5, 1, 2, 3, 4, 5, 6, 7))
    assert blob is None and words[0] == 0x1234 and words[2] == 0xFFFFFFFFFFFFFFFF and words[5] == 0
    assert struct.unpack("<d", struct.pack("<Q", words[4]))[0] == 2.5
    assert struct.unpack("<f", struct.pack("<I", words[8] & 0xFFFFFFFF))[0] == np.float32(1e-5)
    cl = h.fb_cmdlist_create()
    try:
        fn = h.fb_cmd_fn_id(b"fb_bn_fwd_finalize")
        assert h.fb_cmdlist_add_call(cl, fn, words, len(words), 0, None, 0) == 0
        assert h.fb_cmdlist_add_call(cl, fn, words, len(words) - 1, 0, None, 0) != 0      # wrong word count
        assert h.fb_cmdlist_add_call(cl, 9999, words, len(words), 0, None, 0) != 0
        assert h.fb_cmdlist_add_event(cl, 1, 12345, 0) != 0                                # unknown event
        assert h.fb_cmdlist_size(cl) == 1
        streams = (C.c_void_p * 1)(None)
        assert h.fb_cmdlist_replay(cl, streams, 0) != 0                                    # stream index 0 of 0 streams
        assert b"stream" in h.fb_last_error_string()
    finally:
        h.fb_cmdlist_destroy(cl)


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.engine import Engine
    from fullbatchtraining_amd.lib import EngineError
    from fullbatchtraining_amd.models import construct_model

    model = construct_model(compose([]).model, 3, 10)
    with pytest.raises(EngineError):
        Engine(model, 32, 128, 2)


def test_cfg_surface_and_overrides():
    from fullbatchtraining_amd.cfg import compose

    cfg = compose(["hyp=gradreg", "data.batch_size=32", "impl.checkpoint.name=x.pth", "hyp.grad_reg.eps=1e-3"])
    assert cfg.hyp.optim.lr == 0.8 and cfg.hyp.grad_clip == 0.25 and cfg.hyp.grad_reg.block_strength == 0.5
    assert cfg.hyp.grad_reg.eps == 1e-3 and cfg.hyp.warmup == 400 and cfg.hyp.scheduler == "cosine-4000"
    assert cfg.hyp.optim.weight_decay == 5e-4 and cfg.hyp.optim.nesterov is True and cfg.hyp.train_stochastic is False
    assert cfg.data.batch_size == 32 and cfg.impl.checkpoint.name == "x.pth" and cfg.hyp.sub_batch == 128
    assert dict(**cfg.hyp.grad_reg)["implementation"] == "forward-differences"
    base = compose([])
    assert base.hyp.train_stochastic is True and base.model.name == "ResNet18" and base.impl.accumulation_dtype == "float"
    assert compose(["model=resnet152"]).model.depth == 152 and compose(["impl/setup=distributed"]).impl.setup.dist is True


@pytest.mark.parametrize("hyp", ["fb1", "fb2", "gradreg"])
def test_product_lr_schedule_matches_reference(golden, hyp):
    """optim_interface state containers reproduce the reference LR sequence (warm-up from 0, cosine-4000)."""
    import warnings

    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.training import optim_interface

    _, meta = golden
    cfg = compose([f"hyp={hyp}"])
    optimizer, scheduler = optim_interface(torch.nn.Linear(2, 2), cfg.hyp)
    seq = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(3000 if hyp != "fb1" else 300):
            seq.append(optimizer.param_groups[0]["lr"])
            scheduler.step()
    idx = [i for i in meta["lr_index"] if i < len(seq)]
    assert np.allclose([seq[i] for i in idx], meta["lr"][hyp], rtol=1e-12, atol=0)
    if cfg.hyp.warmup > 0:
        state = scheduler.state_dict()
        assert set(state.keys()) == set(meta["checkpoint"]["scheduler_state"].keys())


def test_out_of_scope_options_raise():
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.training import _check_scope, optim_interface

    with pytest.raises(NotImplementedError):
        _check_scope(compose(["hyp=base_sgd"]))                    # stochastic branch
    _check_scope(compose(["hyp=fb1", "hyp.grad_reg.acc_strength=0.1"]))                       # supported: pre-pass over whole blocks
    _check_scope(compose(["hyp=fb1", "hyp.grad_reg.acc_strength=0.1", "data.batch_size=128", "hyp.sub_batch=32"]))   # ... also sub-chunked
    _check_scope(compose(["hyp=fb1", "hyp.batch_clip=1.0"]))                                  # per-chunk L2 clip (fb_mt_chunk_clip)
    with pytest.raises(NotImplementedError):
        _check_scope(compose(["hyp=fb1", "hyp.batch_clip=1.0", "hyp.grad_clip_norm=inf"]))    # ... in the L2 norm only
    with pytest.raises(NotImplementedError):
        _check_scope(compose(["hyp=fbclip", "hyp.grad_clip_norm=0.5"]))                       # p-norms with p >= 1 (or inf)
    for ok in (["hyp=fbclip", "hyp.grad_clip_norm=inf"], ["hyp=fbclip", "hyp.grad_clip_norm=1"], ["hyp=fb1", "hyp.norm_bias.strength=0.1"], ["hyp=fb1", "hyp.evaluate_ema=True"],
               ["hyp=fb1", "hyp/optim_modification=SAM"], ["hyp=fb1", "hyp/optim_modification=LARC"], ["hyp=fb1", "hyp.grad_noise.additive=0.1"], ["hyp=fb1", "hyp.shuffle=True"]):
        _check_scope(compose(ok))
    cfg = compose(["hyp=fb1"])
    cfg.hyp.optim.name = "L-BFGS"
    with pytest.raises(NotImplementedError):
        optim_interface(torch.nn.Linear(2, 2), cfg.hyp)


def test_padded_chunk_sizes():
    """Chunk sizes as stored: whole 128-pixel statistics blocks on every feature map (data.batch_size=125 -> 128 images at 32 px)."""
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.engine import Plan, padded_chunk
    from fullbatchtraining_amd.models import construct_model

    model = construct_model(compose([]).model, 3, 10)
    assert padded_chunk(Plan(model, 32), 128) == 128 and padded_chunk(Plan(model, 32), 125) == 128 and padded_chunk(Plan(model, 32), 500) == 504
    assert padded_chunk(Plan(model, 16), 25) == 32 and padded_chunk(Plan(model, 16), 32) == 32
    assert Plan(model, 32, arena_align=64 * 7).P % 7 == 0 and Plan(model, 32).P % 64 == 0


def test_group_cap_keeps_activations_in_32bit_range(monkeypatch):
    """A chunk group may be as large as 90 % of the device (less what the caller keeps there: ``reserve_bytes``; 288 GB assumed without a GPU) holds its activations
    and per-chunk arenas of -- in both storage types: every kernel bases its descriptors at its own tile / K slice or uses 64-bit pointers (the fp32 rule "largest
    tensor below 2^31 bytes" was lifted in round 6).  FB_BIG_GROUPS=0 keeps the 2^31 rule; it stays the floor."""
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.engine import Plan, max_group
    from fullbatchtraining_amd.models import construct_model

    plan = Plan(construct_model(compose([]).model, 3, 10), 32)
    big = max_group(plan, 128, torch.bfloat16)
    assert 127 < big <= 1024 and big * 128 * 4_000_000 < 288 << 30         # (~3.2 MB of activations per image + gradient buffers)
    assert max_group(plan, 128, torch.float32) == 63            # BasicBlock nets with fp32 storage keep the 2^31 size (measured: larger groups cost 1 % there)
    monkeypatch.setenv("FB_BIG_GROUPS", "2")
    big32 = max_group(plan, 128, torch.float32)
    assert 63 < big32 < big and max_group(plan, 128, torch.float32, fd_sets=1) < big32      # fp32: twice the bytes; the regulariser's per-chunk arenas count
    monkeypatch.setenv("FB_BIG_GROUPS", "0")
    assert max_group(plan, 128, torch.bfloat16) == 127
    assert max_group(plan, 128, torch.float32) == 63 and max_group(plan, 32, torch.float32) == 255
    deep = Plan(construct_model(compose(["model=resnet152", "model.stem=standard"]).model, 3, 10), 224)
    assert max_group(deep, 128, torch.float32) == 5
    monkeypatch.delenv("FB_BIG_GROUPS")
    # ResNet-152 @224: ~13.9 GB per chunk of 128 images by the estimate (round 5 measured 12.4 GiB per chunk + 9.5 GiB: 207.9 GiB at 16 chunks) -- the 16
    # chunks of a 2048-image step run as ONE group beside 10 GB of patches and images, 20 would not; with fp32 storage and the regulariser (config 5 as BASELINE
    # states it; measured 211.4 GiB at 8 chunks) the 8 chunks of a 1024-image step are one group
    assert max_group(deep, 128, torch.bfloat16, reserve_bytes=10 << 30) == 18
    assert max_group(deep, 128, torch.float32, reserve_bytes=9 << 30, fd_sets=1) == 9
    assert max_group(deep, 128, torch.bfloat16, reserve_bytes=200 << 30) == 10                    # (little room left: the 2^31 rule is the floor)
    monkeypatch.setenv("FB_GROUP_MEM_FRAC", "0.3333")
    assert max_group(deep, 128, torch.bfloat16) == 10
    # the deterministic form (nominal group: total memory only) never depends on what is free
    assert max_group(deep, 128, torch.bfloat16, use_free=False) == max_group(deep, 128, torch.bfloat16)


def test_optimizer_wrappers_have_the_reference_surface():
    """optim_modification = SAM / LARS / LARC (reference optimizers.py:57-67): wrapper objects with ``.optim``, shared param_groups,
    attribute pass-through to the wrapped SGD, scheduler bound to the wrapped optimizer, state_dict of the wrapped optimizer."""
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.training import LARS, SAM, optim_interface

    net = torch.nn.Linear(2, 2)
    for mod, cls in (("SAM", SAM), ("LARS", LARS), ("LARC", LARS)):
        cfg = compose(["hyp=fbclip", f"hyp/optim_modification={mod}"])
        assert cfg.hyp.optim_modification.name == mod
        optimizer, scheduler = optim_interface(net, cfg.hyp)
        assert isinstance(optimizer, cls) and isinstance(optimizer.optim, torch.optim.SGD)
        assert optimizer.param_groups is optimizer.optim.param_groups and optimizer.state is optimizer.optim.state
        assert optimizer.state_dict().keys() == optimizer.optim.state_dict().keys()
        assert scheduler.optimizer is optimizer.optim
        lr0 = optimizer.param_groups[0]["lr"]
        scheduler.step()
        assert optimizer.param_groups[0]["lr"] != lr0            # warm-up drives the shared groups
    sam, _ = optim_interface(net, compose(["hyp=fb1", "hyp/optim_modification=SAM", "hyp.optim_modification.rho=0.1"]).hyp)
    assert sam.rho == 0.1
    larc, _ = optim_interface(net, compose(["hyp=fb1", "hyp/optim_modification=LARC"]).hyp)
    assert larc.clip is True and larc.trust_coefficient == 0.02 and larc.eps == 1e-8
    # only_linear_layers_weight_decay (reference optimizers.py:14-21): one group per tensor, biases without decay
    from fullbatchtraining_amd.models import construct_model
    r18 = construct_model(compose([]).model, 3, 10)
    opt, _ = optim_interface(r18, compose(["hyp=fb1", "hyp.only_linear_layers_weight_decay=True"]).hyp)
    names = [k for k, _ in r18.named_parameters()]
    assert len(opt.param_groups) == len(names) == 62
    assert all((g["weight_decay"] == 0.0) == ("bias" in k) for g, k in zip(opt.param_groups, names))
    assert sum(g["weight_decay"] == 0.0 for g in opt.param_groups) == 21             # 20 BN biases + fc.bias
    bad = compose(["hyp=fb1"])
    bad.hyp.optim_modification.name = "Lookahead"
    with pytest.raises(ValueError):
        optim_interface(net, bad.hyp)


def test_shard_plan_partitions_chunks():
    from fullbatchtraining_amd.parallel import ShardPlan

    for k, w in ((390, 8), (390, 1), (7, 2), (3, 4)):
        plans = [ShardPlan(k, w, r) for r in range(w)]
        assert sum(p.count for p in plans) == k
        assert [p.first for p in plans] == [sum(pl.count for pl in plans[:r]) for r in range(w)]
        assert max(p.count for p in plans) - min(p.count for p in plans) <= 1
    assert [ShardPlan(390, 8, r).count for r in range(8)] == [49] * 6 + [48] * 2
    from fullbatchtraining_amd.parallel import group_size
    assert [group_size(c, 39) for c in (390, 195, 98, 97, 49, 48, 7, 1, 58, 59)] == [39, 39, 33, 49, 49, 48, 7, 1, 58, 30]
    assert [group_size(c, 98) for c in (390, 195, 98, 97, 49)] == [98, 98, 98, 97, 49]
    assert [group_size(c, 98, cap=63) for c in (390, 195, 98, 49, 0)] == [56, 49, 49, 49, 63]      # f32: 63 chunks of 128 x 64 x 32 x 32
    for c in range(1, 400):                      # equal groups, never more than 1.5 x the configured size
        g = group_size(c, 39)
        assert 1 <= g < 59 and -(-c // g) * g - c < -(-c // g)


# --------------------------------------------------------------------------------------------- gloo, world_size = 2 ----
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, K, P, out_dir, bucketed=False):
    import sys
    sys.path.insert(0, REPO)
    from fullbatchtraining_amd.parallel import (ShardOps, ShardPlan, all_gather_chunk_stats, combine_running_stats,
                                                reduce_scatter_update_all_gather)

    from tests.helpers import pg_timeout
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=pg_timeout())
    torch.manual_seed(0)                      # identical "dataset" on every rank
    g = torch.randn(K, P)                     # per-chunk (regularised) gradients
    theta = torch.randn(P)
    mom = torch.randn(P) * 0.1
    stats = torch.rand(K)
    bn_mean = torch.randn(K, 2, 5)            # two EMA updates per chunk (finite differences), 5 channels
    r0 = torch.stack([torch.randn(5), torch.rand(5) + 0.5])
    plan = ShardPlan(K, world, rank)
    # rank-local running mean over the owned chunks (what Engine.full_gradient produces)
    avg = torch.zeros(P)
    for j, k in enumerate(range(plan.first, plan.first + plan.count)):
        avg += (g[k] - avg) / (j + 1)
    lr, wd, mu, clip = 0.1, 5e-4, 0.9, 0.25

    def update(lo, n, gnorm2):
        norm = gnorm2.sqrt()
        coef = clip / (norm + 1e-6) if norm > clip else torch.tensor(1.0)
        gr = avg[lo:lo + n] * coef
        avg[lo:lo + n] = gr
        d = gr + wd * theta[lo:lo + n]
        mom[lo:lo + n] = mu * mom[lo:lo + n] + d
        theta[lo:lo + n] -= lr * (d + mu * mom[lo:lo + n])

    ops = ShardOps(scale=lambda t, a: t.mul_(a), sqnorm=lambda t: t.pow(2).sum(), update=update)
    if bucketed:       # two buckets; even ranks start the late one early (as the engine does from its side stream), odd ranks only in finish()
        from fullbatchtraining_amd.parallel import BucketExchange
        ex = BucketExchange(avg, theta, plan, ops, [0, 64 * 2, P])
        if rank % 2 == 0:
            ex.start(1)
        gnorm2 = ex.finish()
    else:
        gnorm2 = reduce_scatter_update_all_gather(avg, theta, plan, ops)
    full_stats = all_gather_chunk_stats(stats[plan.first:plan.first + plan.count].clone(), plan)
    mine = stats[plan.first:plan.first + plan.count]
    full_rows = all_gather_chunk_stats(torch.stack([mine, 2 * mine, mine + 1]), plan)          # several statistics, one collective
    assert torch.equal(full_rows, torch.stack([stats, 2 * stats, stats + 1]))
    # rank-local EMA of BN statistics
    r_local = r0.clone()
    for k in range(plan.first, plan.first + plan.count):
        for u in range(2):
            r_local = 0.9 * r_local + 0.1 * bn_mean[k, u].expand(2, 5)
    combined = combine_running_stats(r0, r_local, plan, updates_per_chunk=2)
    owned = ex.ranges() if bucketed else [(rank * (P // world), P // world)]
    torch.save(dict(theta=theta, gnorm2=gnorm2, stats=full_stats, running=combined, mom_shard=mom.clone(), owned=owned),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("bucketed,K,world", [(False, 7, 2), (True, 7, 2), (True, 390, 8), (False, 390, 8), (True, 5, 8)])
def test_sharded_step_matches_single_process(tmp_path, bucketed, K, world):
    """gloo ranks (2, and the benchmark's 8 with its 390 chunks: 49 / 48 per rank; 5 chunks on 8 ranks: ranks WITHOUT chunks):
    reduce-scatter(sum of K_r/K-scaled local means) + sharded clip/SGD + all-gather == 1-process step on the exact mean (checked with the
    oracle's SGD), stats gathered in chunk order, BN running stats recombined exactly; bucketed: half of the ranks start their late bucket
    early, the others in finish() -- every rank still issues its collectives in the same order."""
    from oracle import fb_oracle as orc

    P = 64 * 6
    port = _free_port()
    from tests.helpers import spawn_bounded
    spawn_bounded(_worker, (world, port, K, P, str(tmp_path), bucketed), world, timeout=180)
    torch.manual_seed(0)
    g, theta, mom, stats = torch.randn(K, P), torch.randn(P), torch.randn(P) * 0.1, torch.rand(K)
    bn_mean = torch.randn(K, 2, 5)
    r0 = torch.stack([torch.randn(5), torch.rand(5) + 0.5])
    mean = g.mean(0)
    norm = mean.norm()
    grad = mean * (0.25 / (norm + 1e-6)) if norm > 0.25 else mean
    params = {"w": theta.clone()}
    momentum = [mom.clone()]
    orc.sgd_step(params, [grad.clone()], momentum, 0.1, dict(momentum=0.9, weight_decay=5e-4, dampening=0.0, nesterov=True))
    running = r0.clone()
    for k in range(K):
        for u in range(2):
            running = 0.9 * running + 0.1 * bn_mean[k, u].expand(2, 5)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    for o in outs:
        assert torch.allclose(o["theta"], params["w"], rtol=1e-5, atol=1e-6)
        assert torch.allclose(o["gnorm2"], norm ** 2, rtol=1e-5)
        assert torch.equal(o["stats"], stats)
        assert torch.allclose(o["running"], running, rtol=1e-5, atol=1e-6)
        for lo, n in o["owned"]:           # sharded momentum: this rank's range of every bucket
            assert torch.allclose(o["mom_shard"][lo:lo + n], momentum[0][lo:lo + n], rtol=1e-5, atol=1e-6)
    for o in outs[1:]:
        assert torch.equal(outs[0]["theta"], o["theta"])


def _chunk_vector(k, P):
    return torch.randn(P, generator=torch.Generator().manual_seed(1000 + k))


def _worker_arena(rank, world, port, K, depth, out_dir):
    """A rank of the bucketed exchange on the arena of a real Bottleneck plan (ResNet-50, 'standard' stem): bounds from parallel.exchange_bounds."""
    import sys
    import types
    sys.path.insert(0, REPO)
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.engine import Plan
    from fullbatchtraining_amd.models import construct_model
    from fullbatchtraining_amd.parallel import BucketExchange, ShardOps, ShardPlan, exchange_bounds

    from tests.helpers import pg_timeout
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=pg_timeout())
    cfg = compose([f"model=resnet{depth}", "model.stem=standard"])
    torch.manual_seed(0)
    arena = Plan(construct_model(cfg.model, 3, 1000), 64, arena_align=64 * world)
    P = arena.P
    plan = ShardPlan(K, world, rank)
    bounds = exchange_bounds(types.SimpleNamespace(engine=types.SimpleNamespace(plan=arena), shard=plan))
    gen = torch.Generator().manual_seed(7)
    theta, mom = torch.randn(P, generator=gen), torch.randn(P, generator=gen) * 0.1
    avg = torch.zeros(P)
    for j, k in enumerate(range(plan.first, plan.first + plan.count)):
        avg += (_chunk_vector(k, P) - avg) / (j + 1)
    lr, wd, mu, clip = 0.1, 5e-4, 0.9, 0.25

    def update(lo, n, gnorm2):
        norm = gnorm2.sqrt()
        coef = clip / (norm + 1e-6) if norm > clip else torch.tensor(1.0)
        gr = avg[lo:lo + n] * coef
        avg[lo:lo + n] = gr
        d = gr + wd * theta[lo:lo + n]
        mom[lo:lo + n] = mu * mom[lo:lo + n] + d
        theta[lo:lo + n] -= lr * (d + mu * mom[lo:lo + n])

    ex = BucketExchange(avg, theta, plan, ShardOps(scale=lambda t, a: t.mul_(a), sqnorm=lambda t: t.pow(2).sum(), update=update), bounds)
    if rank % 2 == 0:                      # half of the ranks start the late bucket early (the engine: from inside the last backward pass)
        ex.start(1)
    gnorm2 = ex.finish()
    if rank in (0, world - 1):
        torch.save(dict(theta=theta, gnorm2=gnorm2, bounds=bounds, owned=ex.ranges(), mom=mom.clone(), P=P,
                        late_offset=arena.late_offset, conv1=arena.offsets[f"layers.{3}.0.conv1.weight"]), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_bucketed_exchange_on_a_bottleneck_arena_world8(tmp_path):
    """Config 5's 8-GPU exchange on the gloo stand-in: the arena of a real Bottleneck plan (ResNet-50, 'standard' stem, 1000 classes: 25.6 M parameters), K = 16 chunks on
    8 ranks (two each), two buckets cut by ``parallel.exchange_bounds`` -- the late one starts at the first parameter of the last stage (``layers.3.0.conv1.weight``, rounded
    up to the shard granule lcm(4, world): what the backward pass has completed when ``on_block_done(plan.late_block)`` fires) and both split evenly over the ranks -- against
    the 1-process step on the exact mean (the oracle's SGD)."""
    from oracle import fb_oracle as orc
    from tests.helpers import spawn_bounded

    K, world = 16, 8
    spawn_bounded(_worker_arena, (world, _free_port(), K, 50, str(tmp_path)), world, timeout=240)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in (0, world - 1)]
    P, bounds = outs[0]["P"], outs[0]["bounds"]
    granule = 8
    assert bounds[0] == 0 and bounds[2] == P and P % (64 * world) == 0
    assert outs[0]["late_offset"] == outs[0]["conv1"] and 0 <= bounds[1] - outs[0]["conv1"] < granule        # a parameter boundary of the last stage's first layer
    assert bounds[1] % granule == 0 and (P - bounds[1]) % world == 0 and 0.5 < (P - bounds[1]) / P < 0.7     # 15 M of ResNet-50's 25.6 M parameters leave early
    mean = torch.zeros(P)
    for k in range(K):
        mean += _chunk_vector(k, P) / K
    gen = torch.Generator().manual_seed(7)
    theta, mom = torch.randn(P, generator=gen), torch.randn(P, generator=gen) * 0.1
    norm = mean.double().norm().float()          # (torch's fp32 norm of 25.6 M elements on the host is 0.15 % off)
    grad = mean * (0.25 / (norm + 1e-6)) if norm > 0.25 else mean
    params, momentum = {"w": theta.clone()}, [mom.clone()]
    orc.sgd_step(params, [grad], momentum, 0.1, dict(momentum=0.9, weight_decay=5e-4, dampening=0.0, nesterov=True))
    for o in outs:
        assert torch.allclose(o["theta"], params["w"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(o["gnorm2"], norm ** 2, rtol=1e-5)
        for lo, n in o["owned"]:
            assert torch.allclose(o["mom"][lo:lo + n], momentum[0][lo:lo + n], rtol=1e-5, atol=1e-6)
    assert torch.equal(outs[0]["theta"], outs[1]["theta"])
    # the two ranks own different ranges of BOTH buckets
    assert outs[0]["owned"][0][0] == 0 and outs[1]["owned"][1][0] + outs[1]["owned"][1][1] == P


def test_bench_line_stays_below_4k_on_a_canned_run():
    """The driver parses the LAST stdout line of bench.py out of a bounded tail (round 3's 23 KB line came back ``parsed: null``): the line is
    assembled by bench.assemble_line, which keeps numbers only and moves tables / notes / sources to gpurun_out/bench_detail.json.  Canned input:
    round 3's full line (profiles/r3_final_bench_line.json) split back into the objects bench.py hands to the assembler."""
    import json

    import bench

    with open(os.path.join(REPO, "profiles", "r3_final_bench_line.json")) as handle:
        full = json.load(handle)
    base = {k: full[k] for k in ("metric", "value", "unit", "steps_per_sec", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                 "dtype", "data", "train_loss_last", "host_enqueue_ms_per_step", "ms_per_step_with_kernel_events")}
    base["config"] = {k: full["config"][k] for k in ("workload", "chunk_group", "parallelism")}
    line_keys = {"roofline": ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "flop_per_launch", "avg_launch_us",
                              "launches_per_step", "frac_isolated", "step_mfma_frac"),
                 "roofline_hbm": ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us")}
    line_part = {name: {k: full[name][k] for k in keys} for name, keys in line_keys.items()}
    line_part["roofline"]["mfma_per_product"] = 1
    line_part["hbm"] = {"bytes_per_step": full["hbm"]["bytes_per_step"], "tb_per_s": full["hbm"]["tb_per_s"], "frac": full["hbm"]["frac"], "source": "profiles/hbm_traffic.json"}
    line_part["mfma_util_pmc"] = {"step": 0.4012, "dominant_kernel": 0.5123, "executed_mfma_tflop_per_step": 166.25, "source": "profiles/mfma_util.json"}
    detail_part = {"roofline": {k: v for k, v in full["roofline"].items() if k not in line_keys["roofline"]}}
    side = {"configs": full["configs"], "parity": full["parity"]}
    cpu = dict(full["cpu_baseline"], sample_short="17 chunks x 128 images fwd+bwd fp32 oracle (torch autograd), 12.3 s")
    line, detail = bench.assemble_line(base, {"outside_the_step": full["outside_the_step"]}, (line_part, detail_part), side, cpu)
    assert len(line) < 4096 and "\n" not in line, len(line)
    out = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in out, key
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(out["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(out["cpu_baseline"])
    assert out["detail"] == bench.DETAIL_PATH and "model" not in out["config"]
    assert set(out["configs"]) == set(full["configs"]) and out["parity"]["two_streams_vs_one"] is True
    # nothing is lost: the tables live in the detail document
    assert len(detail["roofline"]["per_shape"]) == len(full["roofline"]["per_shape"]) and "note" in detail["parity"]["bf16_vs_f32"]
    # and a line that would not fit is refused by the assembler itself
    with pytest.raises(RuntimeError, match="bench line is"):
        bench.assemble_line(dict(base, padding="x" * 4096), {}, (line_part, detail_part), side, cpu)


def test_bench_line_reports_a_failed_resnet152_child_inside_the_line():
    """`bench.py` times BASELINE config 5's model in a child process; if that child fails (out of memory on a smaller device, a timeout) the headline line must still be
    printed, with the failure reported in `configs.r152_error` -- the compact form keeps the message and stays parseable."""
    import json

    import bench

    side = {"configs": {"k400": {"value": 1.0, "unit": "images/s", "ms_per_step": 2.0, "dtype": "bf16", "steps": 5, "workload": "x" * 300},
                        "r152_error": {"value": None, "unit": "images/s", "ms_per_step": None, "dtype": "bf16", "steps": 0, "error": "RuntimeError: exit code 1: HIP out of memory"}},
            "parity": {"two_streams_vs_one": {"bit_identical": True}}}
    out = bench.compact_side(side)
    assert out["configs"]["r152_error"]["error"].startswith("RuntimeError") and out["configs"]["r152_error"]["value"] is None
    assert "workload" not in out["configs"]["k400"] and out["parity"]["two_streams_vs_one"] is True
    line, _ = bench.assemble_line({"metric": "m", "value": 1.0, "config": {}}, {}, None, side, None)
    assert json.loads(line)["configs"]["r152_error"]["steps"] == 0


def test_power_sampler_without_a_gpu_reports_nothing(tmp_path):
    """bench.PowerSampler reads the amdgpu hwmon files of the process's GPU from a host thread; where there is no such device (this container) or the
    files are unreadable it must stay out of the way: start() is a no-op and result() is None (the line then carries ``"power": null``).  With a
    directory that looks like hwmon it samples, averages and converts the units (microwatts -> W, Hz -> MHz)."""
    import time

    import bench

    ps = bench.PowerSampler(0)
    assert ps.dir is None
    ps.start()
    assert ps.result() is None
    for name, value in (("power1_input", "1364000000"), ("freq1_input", "2130000000"), ("power1_cap", "1400000000")):
        (tmp_path / name).write_text(value + "\n")
    ps = bench.PowerSampler(0)
    ps.dir = str(tmp_path)
    ps.start()
    time.sleep(0.7)
    got = ps.result()
    assert got["avg_w"] == 1364 and got["max_w"] == 1364 and got["cap_w"] == 1400 and got["sclk_mhz"] == 2130 and got["samples"] >= 2
