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


def _nchw(t):
    return t.float().cpu().double().permute(0, 3, 1, 2)


def _close(got_nhwc, ref_nchw, what, report):
    ref = ref_nchw.double()
    got = _nchw(got_nhwc).to(ref.device)
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


def test_resnet18_bf16_kernels_layer_by_layer_against_the_oracle():
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.engine import Engine, stem_patches
    from fullbatchtraining_amd.lib import call
    from fullbatchtraining_amd.models import construct_model
    from oracle import fb_oracle as orc

    pixels, chunk, G = 32, 128, 1
    cfg = compose([])
    torch.manual_seed(0)
    model = construct_model(cfg.model, 3, 10)
    eng = Engine(model, pixels, chunk, G, compute_dtype=torch.bfloat16)
    eng.use_replay = False                       # primitives are called one by one with injected tensors
    plan = eng.plan
    x, y = make_data(chunk, pixels)
    q = lambda t: t.to(torch.bfloat16).to(t.dtype)          # noqa: E731  (orc.bf16_round returns float32; the walk runs in float64)
    spec = orc.Spec(18)
    params, buffers = oracle_state(model)                  # the walk runs on the oracle's device (tests/helpers.oracle_device)
    xo, yo = to_oracle(x, y)
    logits_o, tape = orc.forward(spec, params, buffers, xo, q, update_bn=False, train=True)
    loss_o, correct_o, dlogits_o = orc.cross_entropy_fwd_bwd(logits_o, yo)
    report = []
    gout = eng.g
    gout.zero_()
    eng.prep_weights(eng.theta, 1)

    def grad_of(name):
        return eng._unflatten(gout[0].cpu(), name)

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
    _close(eng.stem_out, stem_e["out"], "stem BN+ReLU", report)
    eng.stem_out.copy_(_nhwc(stem_e["out"]))
    put_mask(eng.stem_out, stem_e["out"] > 0)
    a_prev = eng.stem_out
    for bi, b in enumerate(plan.blocks):
        e = tape[1 + bi]
        tag = f"block {bi}"
        c1, c2 = b.convs
        r1, r2 = e["recs"]
        nxt = plan.blocks[bi + 1] if bi + 1 < len(plan.blocks) else None
        fwd_conv(c1, a_prev, r1, f"{tag} conv1")
        eng._bn_apply(c1, b.mids[0], G)
        _close(b.mids[0], e["mids"][0], f"{tag} BN1+ReLU", report)
        b.mids[0].copy_(_nhwc(e["mids"][0]))
        put_mask(b.mids[0], e["mids"][0] > 0)
        fwd_conv(c2, b.mids[0], r2, f"{tag} conv2")
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
            fused = eng._bn_apply(c2, b.out, G, res=b.shortcut.x, resL=b.shortcut, pool=next_pool)
        else:
            fused = eng._bn_apply(c2, b.out, G, res=a_prev, pool=next_pool)
        _close(b.out, e["out"], f"{tag} BN2 + residual + ReLU", report)
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

    for bi in range(len(plan.blocks) - 1, -1, -1):
        b, e = plan.blocks[bi], tape[1 + bi]
        tag = f"block {bi}"
        c1, c2 = b.convs
        r1, r2 = e["recs"]
        a0_dev = plan.blocks[bi - 1].out if bi > 0 else eng.stem_out
        d_dev = _nhwc(da)
        dy_o = q(da * (e["out"] > 0))
        out_bits = eng.masks.get(b.out.data_ptr())
        lazy = b.shortcut is not None or eng._masked_addend_ok(c1, G, 1)
        # conv2 / bn2 (the gradient enters through the block output's ReLU mask)
        dxc2_q, dx2_dev = bwd_layer(c2, r2, d_dev, b.out, dy_o, b.mids[0], f"{tag} conv2", want_dy=not lazy)
        d_mid = eng._dgrad(c2, dx2_dev, G, 1)
        d_mid_o = q(oracle_dx(r2, dxc2_q))
        _close(d_mid, d_mid_o, f"{tag} conv2 input gradient", report)
        eng.pool.put(d_mid)
        # conv1 / bn1
        dy1_o = q(d_mid_o * (e["mids"][0] > 0))
        dxc1_q, dx1_dev = bwd_layer(c1, r1, _nhwc(d_mid_o), b.mids[0], dy1_o, a0_dev, f"{tag} conv1")
        dx0_o = oracle_dx(r1, dxc1_q)
        if b.shortcut is not None:
            S, rd = b.shortcut, e["rd"]
            src = b.pooled if b.pooled is not None else a0_dev
            dxcs_q, dxs_dev = bwd_layer(S, rd, d_dev, b.out, dy_o, src, f"{tag} shortcut")
            d_p = eng._dgrad(S, dxs_dev, G, 1)
            dp_o = q(oracle_dx(rd, dxcs_q))
            _close(d_p, dp_o, f"{tag} shortcut input gradient", report)
            eng.pool.put(d_p)
            d_in = eng._dgrad(c1, dx1_dev, G, 1, addend=_nhwc(dp_o), addend_mode=2 if b.pooled is not None else 1)
            d_o = dx0_o + (orc.avgpool2_bwd(dp_o) if b.stride == 2 else dp_o)
        elif lazy:
            d_in = eng._dgrad(c1, dx1_dev, G, 1, addend=d_dev, addend_mode=1, addend_mask=out_bits)
            d_o = dx0_o + dy_o
        else:
            d_in = eng._dgrad(c1, dx1_dev, G, 1, addend=_nhwc(dy_o), addend_mode=1)
            d_o = dx0_o + dy_o
        da = q(d_o)
        _close(d_in, da, f"{tag} input gradient (conv1 dgrad + residual branch)", report)
        eng.pool.put(d_in)
    dy_o = q(da * (stem_e["out"] > 0))
    bwd_layer(plan.stem, stem_e["rec"], _nhwc(da), eng.stem_out, dy_o, patches, "stem")
    torch.cuda.synchronize()

    worst = sorted(report, key=lambda r: -r[1])[:8]
    print(f"{len(report)} tensors compared; largest relative L2 distances:")
    for what, rel, bad in worst:
        print(f"  {what}: {rel:.3e} ({bad:.1e} of the elements beyond 2 ulp)")
    assert len(report) > 180
