"""Structural parity of the bf16 path -- the arithmetic behind bench.py's headline number -- at the benchmark's REAL shapes (ResNet-18, 32x32,
one chunk of 128 images), kernel by kernel.

End-to-end a bf16 chunk gradient is 0.2 away from the fp32 one (ReLU masks of 2^-9-rounded pre-activations flip; tests/test_gpu_bf16_parity.py
shows that this is noise that averages out over chunks), which would also hide a REAL error of a few per cent in one bf16 kernel.  This test
removes the chaos instead of averaging over it: it walks the network layer by layer in the order of the reference's graph
(fullbatch/models/resnets.py:179-230 forward, autograd's backward of it) and feeds EVERY library launch the float64 oracle's own tensors of
that point -- bf16-rounded at the engine's storage points (oracle ``q = bf16_round``), ReLU masks included -- so that each kernel is compared
with exact arithmetic on identical inputs.  What remains is one bf16 rounding of the output (2^-9 relative) plus fp32 accumulation: every
tensor must agree to 1e-3 relative L2 (measured 4e-5) and element-wise to 2 ulp (plus a floor for cancelling sums).  A wrong tap, parity class, mask bit,
channel slice or coefficient in any of the production bf16 kernels (resident-filter 64-channel, persistent halo, implicit GEMM, stride-2 quad
input gradient, streaming stem, all-taps / per-tap weight gradients, BN apply with residual / pooled output, BN backward with bitmask) moves a
tensor by >= 1/9 and fails here; the suite's statistical bf16 assertions would not see it.
"""
import numpy as np
import pytest
import torch

from tests.helpers import make_data, oracle_state, to_oracle

pytestmark = pytest.mark.gpu

REL_L2 = 1e-3               # per tensor (measured: <= 4.1e-5 on all 209 tensors; a wrong tap moves a tensor by >= 0.1)
ULP2 = 2.0 ** -7            # element-wise: 2 bf16 ulp of the reference value ...
FLOOR = 2.0 ** -9           # ... plus this fraction of the tensor's rms (outputs that are small differences of large terms)
BAD_FRACTION = 1e-3         # elements allowed outside the element-wise bound (1-ulp accumulator differences next to a rounding boundary)


def _nhwc(t):
    """oracle NCHW float64 (bf16-exact values) -> device NHWC bf16"""
    return t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()


def _nchw(t, device):
    """device NHWC -> NCHW float64 on ``device`` (the oracle's: on a GPU box the comparison never leaves the device -- a ResNet-50 walk at 224 px compares ~6 G elements)"""
    return t.to(device).double().permute(0, 3, 1, 2)


def _close(got_nhwc, ref_nchw, what, report):
    ref = ref_nchw.double()
    got = _nchw(got_nhwc, ref.device)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    diff = (got - ref).abs()
    rms = float(ref.pow(2).mean().sqrt())
    rel = float(diff.norm() / max(float(ref.norm()), 1e-30))
    bad = float((diff > ULP2 * ref.abs() + FLOOR * rms).double().mean())
    report.append((what, rel, bad))
    assert np.isfinite(rel) and rel < REL_L2, f"{what}: relative L2 {rel:.3e} (limit {REL_L2:.3e})"
    assert bad < BAD_FRACTION, f"{what}: {bad:.2e} of the elements beyond 2 ulp (+ {FLOOR:.1e} rms)"


def _close_f32(got, ref, what, report, tol=1e-3):
    ref = ref.double().reshape(-1)
    got = got.double().to(ref.device).reshape(-1)
    rel = float((got - ref).norm() / max(float(ref.norm()), 1e-30))
    report.append((what, rel, 0.0))
    assert np.isfinite(rel) and rel < tol, f"{what}: relative L2 {rel:.3e} (limit {tol:.1e})"


def _mask_bytes(positive_nchw):
    """ReLU bitmask in the layout fb_bn_apply writes: one byte per 16-byte vector (8 bf16 channels) of the NHWC tensor, bit k = element k > 0"""
    bits = positive_nchw.permute(0, 2, 3, 1).contiguous().reshape(-1, 8).to(torch.int32)
    weights = (2 ** torch.arange(8, dtype=torch.int32, device=bits.device))
    return (bits * weights).sum(1).to(torch.uint8).cuda()


def _walk(depth, stem, pixels, chunk, classes=10):
    """Forward and backward walk through ``resnet<depth>`` with the given stem: every production bf16 launch on the oracle's own tensors.  Returns the report
    [(what, relative L2, fraction beyond 2 ulp)]."""
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.engine import Engine, stem_patches
    from fullbatchtraining_amd.lib import call
    from fullbatchtraining_amd.models import construct_model
    from oracle import fb_oracle as orc

    G = 1
    cfg = compose([f"model=resnet{depth}", f"model.stem={stem}"])
    torch.manual_seed(0)
    model = construct_model(cfg.model, 3, classes)
    eng = Engine(model, pixels, chunk, G, compute_dtype=torch.bfloat16)
    eng.use_replay = False                       # primitives are called one by one with injected tensors
    plan = eng.plan
    x, y = make_data(chunk, pixels, classes)
    q = lambda t: t.to(torch.bfloat16).to(t.dtype)          # noqa: E731  (orc.bf16_round returns float32; the walk runs in float64)
    spec = orc.Spec(depth, stem=stem, classes=classes)
    params, buffers = oracle_state(model)                  # the walk runs on the oracle's device (tests/helpers.oracle_device)
    xo, yo = to_oracle(x, y)
    logits_o, tape = orc.forward(spec, params, buffers, xo, q, update_bn=False, train=True)
    loss_o, correct_o, dlogits_o = orc.cross_entropy_fwd_bwd(logits_o, yo)
    report = []
    gout = eng.g
    gout.zero_()
    eng.prep_weights(eng.theta, 1)

    def grad_of(name):
        return eng._unflatten(gout[0], name)

    def put_mask(act, positive):
        eng._mask_of(act).copy_(_mask_bytes(positive))

    def fwd_conv(L, src, rec, tag):
        """conv + batch statistics of layer L on the oracle's input; raw output and statistics against the oracle; then the oracle's own
        rounded raw output replaces the engine's (so that everything downstream of this layer starts from identical values)"""
        eng._conv_bn_fwd(L, src, G, 1, eng.theta, 0)
        raw = orc.conv_fwd(rec["x"], rec["w"], rec["stride"], rec["pad"])
        _close(L.x, q(raw), f"{tag} conv output", report)
        mean, var = raw.mean(dim=(0, 2, 3)), raw.var(dim=(0, 2, 3), unbiased=False)
        _close_f32(eng.mean_tab[0, 0, L.ch_off:L.ch_off + L.cout], mean, f"{tag} batch mean", report, 1e-4)
        _close_f32(eng.var_tab[0, 0, L.ch_off:L.ch_off + L.cout], var, f"{tag} batch var", report, 1e-4)
        _close_f32(L.invstd[0], rec["bn"][1], f"{tag} invstd", report, 1e-4)
        L.x.copy_(_nhwc(q(raw)))

    # ------------------------------------------------------------------------------------------------------------ forward walk --
    stem_e = tape[0]
    patches = stem_patches(x.cuda(), plan.stem, torch.bfloat16)
    fwd_conv(plan.stem, patches, stem_e["rec"], "stem")
    eng._bn_apply(plan.stem, eng.stem_out, G)
    stem_act = stem_e["relu_out"] if plan.stem_pool else stem_e["out"]
    _close(eng.stem_out, stem_act, "stem BN+ReLU", report)
    eng.stem_out.copy_(_nhwc(stem_act))
    put_mask(eng.stem_out, stem_act > 0)
    a_prev = eng.stem_out
    if plan.stem_pool:                           # MaxPool2d(3, 2, 1) of the ImageNet stem, remembering its argmax (reference resnets.py:74-79)
        s_ = plan.stem
        assert eng.stem_pool_idx is not None
        call("fb_maxpool3s2_fwd_idx", eng.stem_out.data_ptr(), eng.stem_pooled.data_ptr(), eng.stem_pool_idx.data_ptr(), G * chunk, s_.hout, s_.wout, 64, eng.dtc)
        _close(eng.stem_pooled, stem_e["out"], "stem MaxPool2d(3,2,1)", report)
        assert torch.equal(eng.stem_pooled, _nhwc(stem_e["out"]))          # (a selection: no rounding at all)
        a_prev = eng.stem_pooled
    for bi, b in enumerate(plan.blocks):
        e = tape[1 + bi]
        tag = f"block {bi}"
        recs = e["recs"]
        nxt = plan.blocks[bi + 1] if bi + 1 < len(plan.blocks) else None
        cur = a_prev
        for i, L in enumerate(b.convs[:-1]):
            fwd_conv(L, cur, recs[i], f"{tag} conv{i + 1}")
            eng._bn_apply(L, b.mids[i], G)
            _close(b.mids[i], e["mids"][i], f"{tag} BN{i + 1}+ReLU", report)
            b.mids[i].copy_(_nhwc(e["mids"][i]))
            put_mask(b.mids[i], e["mids"][i] > 0)
            cur = b.mids[i]
        last = b.convs[-1]
        fwd_conv(last, cur, recs[-1], f"{tag} conv{len(b.convs)}")
        next_pool = nxt.pooled if nxt is not None else None
        if b.shortcut is not None:
            src = a_prev
            if b.pooled is not None:
                # (the previous block's output pass wrote this pooled input in production; here the stand-alone kernel on the oracle's tensor)
                call("fb_avgpool2_fwd", a_prev.data_ptr(), b.pooled.data_ptr(), G * chunk, b.hin, b.win, b.cin, eng.dtc)
                _close(b.pooled, e["rd"]["x"], f"{tag} AvgPool2d(2,2)", report)
                b.pooled.copy_(_nhwc(e["rd"]["x"]))
                src = b.pooled
            fwd_conv(b.shortcut, src, e["rd"], f"{tag} shortcut conv")
            fused = eng._bn_apply(last, b.out, G, res=b.shortcut.x, resL=b.shortcut, pool=next_pool)
        else:
            fused = eng._bn_apply(last, b.out, G, res=a_prev, pool=next_pool)
        _close(b.out, e["out"], f"{tag} last BN + residual + ReLU", report)
        if fused:                                  # the pooled copy of this output for the next block's shortcut, written by the same pass
            _close(nxt.pooled, q(orc.avgpool2_fwd(e["out"])), f"{tag} fused pooled output", report)
        b.out.copy_(_nhwc(e["out"]))
        put_mask(b.out, e["out"] > 0)
        a_prev = b.out
    n, hw = G * chunk, plan.h_final * plan.h_final
    call("fb_head_pool", a_prev.data_ptr(), eng.feat.data_ptr(), n, hw, plan.feat, eng.dtc)
    call("fb_head_loss", eng.feat.data_ptr(), eng.theta.data_ptr() + 4 * plan.fcw_off, eng.theta.data_ptr() + 4 * plan.fcb_off, 0, y.cuda().data_ptr(),
         eng.logits.data_ptr(), eng.dlogits.data_ptr(), eng.loss.data_ptr(), eng.correct.data_ptr(), G, chunk, plan.feat, plan.classes, 0.0, 0)
    _close_f32(eng.logits, logits_o, "logits", report, 1e-5)
    _close_f32(eng.dlogits, dlogits_o, "dlogits", report, 1e-5)
    assert abs(float(eng.loss[0]) - float(loss_o)) < 1e-5 * float(loss_o) and float(eng.correct[0]) == float(correct_o)

    # ----------------------------------------------------------------------------------------------------------- backward walk --
    head = tape[-1]
    d = eng.pool.get((n, plan.h_final, plan.h_final, plan.feat))
    call("fb_head_bwd", eng.feat.data_ptr(), eng.dlogits.data_ptr(), eng.theta.data_ptr() + 4 * plan.fcw_off, 0, gout.data_ptr() + 4 * plan.fcw_off,
         gout.data_ptr() + 4 * plan.fcb_off, plan.P, d.data_ptr(), G, chunk, hw, plan.feat, plan.classes, eng.dtc)
    da = q(((dlogits_o @ params["fc.weight"]) / head["spatial"])[:, :, None, None].expand(head["shape"]).contiguous())
    _close(d, da, "head input gradient", report)
    _close_f32(grad_of("fc.weight"), dlogits_o.t() @ head["feat"], "fc.weight gradient", report, 1e-4)

    def sync_wgrad():
        if eng.wstream is not None:
            torch.cuda.current_stream().wait_stream(eng.wstream)

    def bwd_layer(L, rec, d_dev, mask_act, dy_o, src_dev, tag, want_dy=False):
        """BN backward (reduce, finalize, apply -- with the ReLU bitmask of ``mask_act``) on the oracle's incoming gradient, then the weight
        gradient on the oracle's (activation, dx) pair.  Returns the oracle's rounded dx on the device and, with ``want_dy``, checks dy."""
        dxc, dgam, dbet = orc.bn_train_bwd(dy_o, rec["gamma"], rec["bn"])
        dxc_q = q(dxc)
        dx_e, dy_e = eng._bn_bwd(L, d_dev, mask_act, G, gout, 0, want_dy=want_dy)
        _close(dx_e, dxc_q, f"{tag} BN backward dx", report)
        if want_dy:
            _close(dy_e, dy_o, f"{tag} masked gradient dy", report)
            eng.pool.put(dy_e)
        _close_f32(gout[0, L.g_off:L.g_off + L.cout], dgam, f"{tag} dgamma", report, 2e-4)
        _close_f32(gout[0, L.b_off:L.b_off + L.cout], dbet, f"{tag} dbeta", report, 2e-4)
        eng.pool.put(dx_e)
        dx_dev = _nhwc(dxc_q)
        _, dw = orc.conv_bwd(rec["x"], rec["w"], dxc_q, rec["stride"], rec["pad"], need_dx=False)
        eng._wgrad(L, src_dev, dx_dev, G, gout)
        sync_wgrad()
        _close_f32(grad_of(f"{L.conv_name}.weight"), dw, f"{tag} weight gradient", report, 1e-3)
        return dxc_q, dx_dev

    def oracle_dx(rec, dxc_q):
        return torch.nn.grad.conv2d_input(rec["x"].shape, rec["w"], dxc_q, rec["stride"], rec["pad"])

    stem_res = eng.stem_pooled if plan.stem_pool else eng.stem_out
    for bi in range(len(plan.blocks) - 1, -1, -1):
        b, e = plan.blocks[bi], tape[1 + bi]
        tag = f"block {bi}"
        recs = e["recs"]
        first = b.convs[0]
        a0_dev = plan.blocks[bi - 1].out if bi > 0 else stem_res
        srcs = [a0_dev] + b.mids
        d_dev = _nhwc(da)
        dy_o = q(da * (e["out"] > 0))
        out_bits = eng.masks.get(b.out.data_ptr())
        lazy = b.shortcut is not None or eng._masked_addend_ok(first, G, 1)
        # last conv / BatchNorm of the block (the gradient enters through the block output's ReLU mask), then down the chain
        nc = len(b.convs)
        dxc_q, dx_dev = bwd_layer(b.convs[-1], recs[-1], d_dev, b.out, dy_o, srcs[-1], f"{tag} conv{nc}", want_dy=not lazy)
        for i in range(nc - 1, 0, -1):
            d_mid = eng._dgrad(b.convs[i], dx_dev, G, 1)
            d_mid_o = q(oracle_dx(recs[i], dxc_q))
            _close(d_mid, d_mid_o, f"{tag} conv{i + 1} input gradient", report)
            eng.pool.put(d_mid)
            dy_mid_o = q(d_mid_o * (e["mids"][i - 1] > 0))
            dxc_q, dx_dev = bwd_layer(b.convs[i - 1], recs[i - 1], _nhwc(d_mid_o), b.mids[i - 1], dy_mid_o, srcs[i - 1], f"{tag} conv{i}")
        dx0_o = oracle_dx(recs[0], dxc_q)
        if b.shortcut is not None:
            S, rd = b.shortcut, e["rd"]
            src = b.pooled if b.pooled is not None else a0_dev
            dxcs_q, dxs_dev = bwd_layer(S, rd, d_dev, b.out, dy_o, src, f"{tag} shortcut")
            d_p = eng._dgrad(S, dxs_dev, G, 1)
            dp_o = q(oracle_dx(rd, dxcs_q))
            _close(d_p, dp_o, f"{tag} shortcut input gradient", report)
            eng.pool.put(d_p)
            d_in = eng._dgrad(first, dx_dev, G, 1, addend=_nhwc(dp_o), addend_mode=2 if b.pooled is not None else 1)
            d_o = dx0_o + (orc.avgpool2_bwd(dp_o) if b.stride == 2 else dp_o)
        elif lazy:
            d_in = eng._dgrad(first, dx_dev, G, 1, addend=d_dev, addend_mode=1, addend_mask=out_bits)
            d_o = dx0_o + dy_o
        else:
            d_in = eng._dgrad(first, dx_dev, G, 1, addend=_nhwc(dy_o), addend_mode=1)
            d_o = dx0_o + dy_o
        da = q(d_o)
        _close(d_in, da, f"{tag} input gradient (conv1 dgrad + residual branch{', masked addend' if (lazy and b.shortcut is None) else ''})", report)
        eng.pool.put(d_in)
    S = plan.stem
    if plan.stem_pool:
        # MaxPool backward from the remembered argmax (the oracle scatters through torch's own indices of the same rounded tensor)
        r = stem_e["relu_out"]
        dr = torch.zeros_like(r).flatten(2)
        dr.scatter_add_(2, stem_e["pool_idx"].flatten(2), da.flatten(2))
        dr = q(dr.view_as(r))
        d_r = eng.pool.get((n, S.hout, S.wout, 64))
        call("fb_maxpool3s2_bwd_idx", eng.stem_pool_idx.data_ptr(), _nhwc(da).data_ptr(), d_r.data_ptr(), n, S.hout, S.wout, 64, eng.dtc)
        _close(d_r, dr, "stem MaxPool backward", report)
        eng.pool.put(d_r)
        da = dr
    dy_o = q(da * (stem_act > 0))
    bwd_layer(S, stem_e["rec"], _nhwc(da), eng.stem_out, dy_o, patches, "stem")
    if eng._wgrad_bn_ok(S, eng.stem_out):
        # the production form of the stem's backward: no dx tensor -- the weight gradient applies the BatchNorm backward in its operand loader
        gout[0, S.w_off:S.w_off + S.cout * S.taps * S.cin_real].zero_()
        d_dev = _nhwc(da)
        eng._bn_bwd(S, d_dev, eng.stem_out, G, gout, 0, want_dy=False, apply=False)
        eng._wgrad(S, patches, None, G, gout, bn=(d_dev, eng.stem_out))
        sync_wgrad()
        rec = stem_e["rec"]
        dxc, _, _ = orc.bn_train_bwd(dy_o, rec["gamma"], rec["bn"])
        _, dw = orc.conv_bwd(rec["x"], rec["w"], q(dxc), rec["stride"], rec["pad"], need_dx=False)   # (dx becomes a bf16 MFMA operand in the loader: the same rounding point)
        _close_f32(grad_of(f"{S.conv_name}.weight"), dw, "stem weight gradient (BatchNorm backward in the loader)", report, 1e-3)
    torch.cuda.synchronize()

    worst = sorted(report, key=lambda r: -r[1])[:8]
    print(f"resnet{depth} / {stem} stem / {pixels} px / chunk {chunk}: {len(report)} tensors compared; largest relative L2 distances:")
    for what, rel, bad in worst:
        print(f"  {what}: {rel:.3e} ({bad:.1e} of the elements beyond 2 ulp)")
    return report


def test_resnet18_bf16_kernels_layer_by_layer_against_the_oracle():
    report = _walk(18, "CIFAR", 32, 128)
    assert len(report) > 180


@pytest.mark.parametrize("pixels,chunk", [(64, 32), (224, 128)])
def test_resnet50_bottleneck_bf16_kernels_layer_by_layer_against_the_oracle(pixels, chunk):
    """The Bottleneck path of bench.py's ResNet-152 lines (reference resnets.py:271-316, 'standard' stem :74-79), kernel by kernel: ResNet-50 has every layer shape
    of ResNet-152 (the deeper net repeats the identity blocks of stages 2 and 3).  At 224 px with one chunk of 128 images these are the production launches of
    BASELINE config 5's bf16 form: the 7x7 stem on 160-value patches and its 64 x 160 weight-gradient tile, MaxPool with a remembered argmax, streaming / pipelined
    1x1 kernels on 56 / 28 / 14 / 7 maps incl. the masked residual addend of the identity blocks, the all-taps 3x3 weight gradients on ImageNet-shaped maps, the
    1x1 weight-gradient GEMM, stride-2 3x3 layers, shortcuts with and without AvgPool; 64 px / chunks of 32: the shapes of the engine-level oracle tests."""
    report = _walk(50, "standard", pixels, chunk)
    names = [r[0] for r in report]
    assert sum("masked addend" in w for w in names) >= 10 and any("MaxPool backward" in w for w in names)
    assert len(report) > 500
