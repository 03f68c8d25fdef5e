"""GPU parity of every libfbengine.so entry point against plain torch CPU math (same inputs, seeded).

f32 kernels (exact-f32 MFMA) are held to accumulation-order roundoff; bf16 kernels are compared against the same math
evaluated on bf16-rounded inputs with fp32 accumulation (tolerance = bf16 output rounding, 2^-8 relative).
"""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _lib():
    from fullbatchtraining_amd import lib
    return lib


def q(t, dtype):
    return t.to(dtype).float()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def tol(dtype, k=1.0):
    return (2e-5 if dtype == torch.float32 else 1.2e-2) * k


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def krsc(w):  # [co, ci, kh, kw] -> [co, kh*kw, ci]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1, w.shape[1]).contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,k,stride,hw,n", [(64, 64, 3, 1, 8, 8), (64, 128, 3, 2, 8, 8), (128, 256, 1, 1, 4, 16),
                                                   (32, 64, 1, 1, 8, 4), (256, 256, 3, 1, 4, 32), (96, 64, 3, 1, 6, 3),
                                                   (64, 64, 3, 1, 32, 2), (128, 128, 3, 1, 16, 3), (64, 128, 3, 1, 16, 2),   # LDS-halo kernel
                                                   (256, 128, 3, 1, 8, 4), (128, 64, 3, 1, 8, 12), (192, 64, 3, 1, 32, 1),
                                                   (64, 64, 3, 1, 8, 2400), (64, 128, 3, 1, 16, 300), (64, 64, 3, 1, 32, 160),   # > 2 tiles per persistent workgroup
                                                   (32, 64, 1, 1, 32, 40), (32, 64, 1, 1, 32, 300),   # stem on patches: streaming K = 32 kernel (bf16), grid-stride
                                                   (64, 128, 3, 2, 32, 3), (128, 256, 3, 2, 16, 6), (96, 64, 3, 2, 16, 4),   # more stride-2 shapes
                                                   (128, 128, 3, 1, 4, 48), (512, 512, 3, 1, 4, 272), (64, 256, 3, 1, 4, 16), (64, 64, 3, 1, 4, 32),   # 4x4 maps: compact halo layout (bf16)
                                                   # short-K 1x1 convolutions: streaming kernel (bf16), every (K, channels-per-wave) variant, ragged pixel counts,
                                                   # several units per persistent workgroup
                                                   (64, 128, 1, 1, 16, 3), (128, 256, 1, 1, 8, 5), (256, 512, 1, 1, 4, 16), (256, 1024, 1, 1, 14, 2), (64, 256, 1, 1, 8, 3),
                                                   (128, 128, 1, 1, 8, 40), (256, 128, 1, 1, 6, 3), (64, 128, 1, 1, 6, 3), (64, 128, 1, 1, 16, 700), (256, 256, 1, 1, 14, 300),
                                                   # 1x1 with K >= 512: persistent GEMM tiles through a three-stage LDS ring (bf16); ragged pixel counts, one and several
                                                   # tiles per workgroup, one to sixteen channel tiles
                                                   (512, 128, 1, 1, 8, 5), (1024, 256, 1, 1, 14, 3), (2048, 512, 1, 1, 7, 4), (512, 2048, 1, 1, 7, 2), (576, 128, 1, 1, 14, 400)])
def test_conv_fwd_and_stats(dtype, cin, cout, k, stride, hw, n, monkeypatch):
    lib = _lib()
    torch.manual_seed(0)
    pad = k // 2
    x = q(torch.randn(n, cin, hw, hw), dtype)
    w = q(torch.randn(cout, cin, k, k) * 0.1, dtype)
    ref = F.conv2d(x, w, None, stride, pad)
    ho = ref.shape[2]
    xd = nhwc(x).to(dtype).cuda()
    wd = krsc(w).to(dtype).cuda()
    out = torch.empty(n, ho, ho, cout, dtype=dtype, device="cuda")
    nblk = (n * ho * ho + 127) // 128
    stat = torch.zeros(2, nblk, cout, device="cuda")
    lib.conv2d(xd, wd, out, k, k, stride, pad, 0, stat_partial=stat)
    got = nchw(out.float().cpu())
    assert rel(got, ref) < tol(dtype)
    refn = nhwc(ref).reshape(-1, cout)
    assert rel(stat[0].sum(0).cpu(), refn.sum(0)) < 1e-4
    assert rel(stat[1].sum(0).cpu(), (refn * refn).sum(0)) < 1e-4
    if k == 1 and dtype == torch.bfloat16 and cin in (64, 128, 256) and cout % 128 == 0:
        # the pipelined streaming kernel (round 5) against the round-3 form: same products in the same order, same per-lane order of the statistics' sums
        monkeypatch.setenv("FB_C1S_PIPE", "0")
        out2, stat2 = torch.empty_like(out), torch.zeros_like(stat)
        lib.conv2d(xd, wd, out2, k, k, stride, pad, 0, stat_partial=stat2)
        monkeypatch.delenv("FB_C1S_PIPE")
        assert torch.equal(out, out2) and torch.equal(stat, stat2)
    if k == 1 and dtype == torch.bfloat16 and cin >= 512:
        # the K >= 512 GEMM kernel against the implicit GEMM: the same K-steps in the same order (same output bits); the statistics' fp32 sums associate
        # differently (one wave per 128-pixel block here, two half-block waves there)
        monkeypatch.setenv("FB_C1G", "1")                # (every forward call it can take: the default asks for five tiles per workgroup)
        out2, stat2 = torch.empty_like(out), torch.zeros_like(stat)
        lib.conv2d(xd, wd, out2, k, k, stride, pad, 0, stat_partial=stat2)
        monkeypatch.delenv("FB_C1G")
        assert torch.equal(out, out2) and rel(stat, stat2) < 1e-5
    if stride == 2 and dtype == torch.bfloat16:
        # implicit GEMM: the double-buffered two-workgroup form (FB_IGEMM_STAGES=2) multiplies the same K-steps in the same order
        monkeypatch.setenv("FB_IGEMM_STAGES", "2")
        out2, stat2 = torch.empty_like(out), torch.zeros_like(stat)
        lib.conv2d(xd, wd, out2, k, k, stride, pad, 0, stat_partial=stat2)
        assert torch.equal(out, out2) and torch.equal(stat, stat2)
    if hw == 4 and k == 3 and dtype == torch.bfloat16 and cout % 128 == 0:
        # the padded 4x4 layout with 128-channel tiles (FB_H4_COMPACT=0) adds the same products in the same order plus zeros: same bits
        monkeypatch.setenv("FB_H4_COMPACT", "0")
        out2, stat2 = torch.empty_like(out), torch.zeros_like(stat)
        lib.conv2d(xd, wd, out2, k, k, stride, pad, 0, stat_partial=stat2)
        assert torch.equal(out, out2) and torch.equal(stat, stat2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,k,stride,hw,n,amode", [(64, 64, 3, 1, 8, 8, 0), (64, 128, 3, 2, 8, 8, 2), (128, 64, 1, 1, 4, 16, 1),
                                                          (64, 64, 3, 1, 8, 4, 1), (256, 128, 3, 2, 8, 4, 0),
                                                          (64, 64, 3, 1, 32, 2, 1), (128, 64, 3, 1, 16, 2, 0), (64, 64, 3, 1, 16, 2, 2),  # halo kernel
                                                          (128, 256, 3, 1, 8, 8, 1), (128, 64, 3, 1, 8, 4, 2), (256, 256, 3, 1, 8, 4, 0),
                                                          (64, 64, 3, 1, 32, 136, 1), (128, 64, 3, 1, 16, 600, 0),   # persistent workgroups, several tiles each
                                                          # stride-2 quad kernel (bf16): dY 16x16 / 8x8 / 4x4, ragged image counts, all addend modes
                                                          (64, 128, 3, 2, 32, 3, 2), (128, 256, 3, 2, 16, 5, 1), (64, 64, 3, 2, 8, 11, 2),
                                                          (128, 96, 3, 2, 16, 2, 0), (64, 128, 3, 2, 32, 70, 0),
                                                          (256, 128, 3, 1, 4, 32, 1), (128, 256, 3, 1, 4, 16, 0), (512, 512, 3, 1, 4, 272, 1), (64, 64, 3, 1, 4, 16, 2),   # 4x4 maps: compact halo layout
                                                          # 1x1 input gradients: streaming kernel (K = the forward layer's output channels <= 256), with / without the
                                                          # same-shape addend; pooled addend and K = 512 stay on the implicit GEMM
                                                          (256, 128, 1, 1, 8, 5, 0), (1024, 256, 1, 1, 7, 4, 1), (512, 128, 1, 1, 8, 3, 1), (128, 64, 1, 1, 16, 300, 1),
                                                          (256, 64, 1, 1, 8, 6, 2), (256, 512, 1, 1, 4, 16, 1),
                                                          # 1x1 input gradients with K >= 512 (the GEMM kernel; with an addend they stay on the implicit GEMM)
                                                          (128, 512, 1, 1, 8, 5, 0), (256, 1024, 1, 1, 14, 3, 0), (512, 2048, 1, 1, 7, 6, 0), (256, 1024, 1, 1, 14, 3, 1)])
def test_conv_dgrad(dtype, cin, cout, k, stride, hw, n, amode, monkeypatch):
    lib = _lib()
    torch.manual_seed(1)
    pad = k // 2
    ho = (hw + 2 * pad - k) // stride + 1
    w = q(torch.randn(cout, cin, k, k) * 0.1, dtype)
    dy = q(torch.randn(n, cout, ho, ho), dtype)
    ref = torch.nn.grad.conv2d_input((n, cin, hw, hw), w, dy, stride, pad)
    addend = None
    if amode == 1:
        addend = q(torch.randn(n, cin, hw, hw), dtype)
        ref = ref + addend
        add_d = nhwc(addend).to(dtype).cuda()
    elif amode == 2:
        addend = q(torch.randn(n, cin, hw // 2, hw // 2), dtype)
        ref = ref + 0.25 * addend.repeat_interleave(2, 2).repeat_interleave(2, 3)
        add_d = nhwc(addend).to(dtype).cuda()
    else:
        add_d = None
    # dgrad weights: [ci][tap][co]
    wt = w.permute(1, 2, 3, 0).reshape(cin, k * k, cout).contiguous().to(dtype).cuda()
    dyd = nhwc(dy).to(dtype).cuda()
    out = torch.empty(n, hw, hw, cin, dtype=dtype, device="cuda")
    lib.conv2d(dyd, wt, out, k, k, stride, pad, 1, addend=add_d, addend_mode=amode)
    assert rel(nchw(out.float().cpu()), ref) < tol(dtype)
    if hw == 4 and k == 3 and stride == 1 and dtype == torch.bfloat16 and cin % 128 == 0:
        monkeypatch.setenv("FB_H4_COMPACT", "0")       # padded 4x4 layout, 128-channel tiles: same bits
        out2 = torch.empty_like(out)
        lib.conv2d(dyd, wt, out2, k, k, stride, pad, 1, addend=add_d, addend_mode=amode)
        assert torch.equal(out, out2)
    if k == 1 and amode in (0, 1) and dtype == torch.bfloat16 and cout <= 256 and cin % 128 == 0:
        # 1x1 input gradients, three forms of one arithmetic (same MFMA order, the addend added in fp32 before the one rounding): the pipelined
        # streaming kernel (round 5: epilogue threaded through the next group's MFMAs, addend as prefetched 16-byte loads), the round-3 streaming
        # kernel with prefetched addend, and with the addend as 8-byte loads where it is consumed -- same bits
        for env in ({"FB_C1S_PIPE": "0"}, {"FB_C1S_PIPE": "0", "FB_C1S_ADD_ASM": "0"}):
            for key, val in env.items():
                monkeypatch.setenv(key, val)
            out2 = torch.empty_like(out)
            lib.conv2d(dyd, wt, out2, k, k, stride, pad, 1, addend=add_d, addend_mode=amode)
            assert torch.equal(out, out2), env
        monkeypatch.delenv("FB_C1S_PIPE"), monkeypatch.delenv("FB_C1S_ADD_ASM")
    if k == 1 and amode == 0 and dtype == torch.bfloat16 and cout >= 512:
        monkeypatch.setenv("FB_C1G", "2")                # K >= 512: the GEMM kernel (opt-in for input gradients) and the implicit GEMM multiply the same K-steps in the same order
        out2 = torch.empty_like(out)
        lib.conv2d(dyd, wt, out2, k, k, stride, pad, 1)
        monkeypatch.delenv("FB_C1G")
        assert torch.equal(out, out2)
    if stride == 2 and dtype == torch.bfloat16:
        # the quad kernel's double-buffered one-workgroup form (FB_S2Q_STAGES=2) accumulates in the same order: same bits
        monkeypatch.setenv("FB_S2Q_STAGES", "2")
        out2 = torch.empty_like(out)
        lib.conv2d(dyd, wt, out2, k, k, stride, pad, 1, addend=add_d, addend_mode=amode)
        assert torch.equal(out, out2)


@pytest.mark.parametrize("cin,cout,hw,n", [(256, 1024, 14, 1024), (64, 256, 56, 256), (256, 512, 4, 12544), (128, 512, 28, 512)])
def test_conv1x1_kernels_agree_at_full_size(cin, cout, hw, n, monkeypatch):
    """The 1x1 kernels of round 5 at the sizes the benchmarks run (one ResNet-152 chunk group of 1024 images @14x14, the ResNet-18 shortcut of a 98-chunk
    group; 50 000+ units per launch, every persistent workgroup walks tens of them): the pipelined streaming kernel, the round-3 streaming kernel and
    the round-3 kernel with the addend as in-place 8-byte loads give the SAME BITS -- outputs, BatchNorm partial sums, input gradients with addend --
    and the input gradient is linear in the addend's absence / presence (out(addend) - out(0) == addend wherever no rounding intervenes is too strong
    for bf16; instead: out(addend = 0 tensor) == out(no addend))."""
    lib = _lib()
    torch.manual_seed(5)
    dt = torch.bfloat16
    x = torch.randn(n, hw, hw, cin, device="cuda").to(dt)
    w = (torch.randn(cout, 1, cin, device="cuda") * 0.05).to(dt)
    wt = (torch.randn(cin, 1, cout, device="cuda") * 0.05).to(dt)
    dy = torch.randn(n, hw, hw, cout, device="cuda").to(dt)
    add = torch.randn(n, hw, hw, cin, device="cuda").to(dt)
    nblk = (n * hw * hw + 127) // 128

    def run():
        out, stat = torch.empty(n, hw, hw, cout, device="cuda", dtype=dt), torch.zeros(2, nblk, cout, device="cuda")
        lib.conv2d(x, w, out, 1, 1, 1, 0, 0, stat_partial=stat)
        dx0, dx1, dxz = (torch.zeros(n, hw, hw, cin, device="cuda", dtype=dt) for _ in range(3))
        if cin % 128 == 0:               # (input gradients onto 64 channels stay on the implicit GEMM)
            lib.conv2d(dy, wt, dx0, 1, 1, 1, 0, 1)
            lib.conv2d(dy, wt, dx1, 1, 1, 1, 0, 1, addend=add, addend_mode=1)
            lib.conv2d(dy, wt, dxz, 1, 1, 1, 0, 1, addend=torch.zeros_like(add), addend_mode=1)
        torch.cuda.synchronize()
        return out, stat, dx0, dx1, dxz

    monkeypatch.setenv("FB_C1G", "0")                        # (this test is about the two streaming kernels: a forward call of this size with K = 256 goes to the GEMM kernel by default)
    new = run()
    monkeypatch.setenv("FB_C1S_PIPE", "0")
    old = run()
    monkeypatch.setenv("FB_C1S_ADD_ASM", "0")
    older = run()
    for a, b, c in zip(new, old, older):
        assert torch.equal(a, b) and torch.equal(b, c)
    if cin % 128 == 0:
        assert torch.equal(new[2], new[4])                                  # a zero addend changes nothing
        assert not torch.equal(new[2], new[3])
    # statistics of the whole launch against torch's own matmul of the same bf16 operands (fp32 accumulation)
    ref = (x.reshape(-1, cin).float() @ w.reshape(cout, cin).float().t())
    assert rel(new[1][0].sum(0), ref.sum(0)) < 2e-3 and rel(new[1][1].sum(0), (ref * ref).sum(0)) < 1e-4
    assert rel(new[0].reshape(-1, cout).float(), ref) < tol(dt)


@pytest.mark.parametrize("cin,cout,hw,n", [(1024, 256, 14, 1024), (2048, 512, 7, 1000), (512, 2048, 7, 300), (256, 1024, 14, 700),
                                           (128, 512, 28, 300), (128, 512, 28, 37), (192, 768, 14, 500)])   # K = 128 / 192: two / three K-steps per tile (round 2 of the pixel ring is the NEXT tile's)
def test_conv1x1_gemm_kernel_agrees_with_the_implicit_gemm_at_full_size(cin, cout, hw, n, monkeypatch):
    """The ping-pong GEMM kernel (csrc/conv1x1_gemm.hip; FB_C1G=2: every call it can take) at a ResNet-152 chunk group's size: persistent workgroups walk three or four
    256 x 256 tiles each (784 tiles @14x14; 192 pixel tiles x 2 or 58 x 8 channel tiles @7x7 with a ragged last pixel tile), the two wave groups half a
    step apart, rings running across tile boundaries.  Same K-steps in the same order as the implicit GEMM: the SAME BITS for forward outputs and input
    gradients; the statistics' fp32 sums associate differently (one wave per 128-pixel block here) and agree to 1e-5; both against torch's matmul."""
    lib = _lib()
    torch.manual_seed(11)
    dt = torch.bfloat16
    x = torch.randn(n, hw, hw, cin, device="cuda").to(dt)
    w = (torch.randn(cout, 1, cin, device="cuda") * 0.03).to(dt)
    wt = (torch.randn(cout, 1, cin, device="cuda") * 0.03).to(dt)       # as a transposed set: input gradient of a cout <- cin ... convolution with K = cin
    nblk = (n * hw * hw + 127) // 128

    def run():
        out, stat, dx = torch.empty(n, hw, hw, cout, device="cuda", dtype=dt), torch.zeros(2, nblk, cout, device="cuda"), torch.empty(n, hw, hw, cout, device="cuda", dtype=dt)
        lib.conv2d(x, w, out, 1, 1, 1, 0, 0, stat_partial=stat)
        lib.conv2d(x, wt, dx, 1, 1, 1, 0, 1)
        torch.cuda.synchronize()
        return out, stat, dx

    monkeypatch.setenv("FB_C1G", "0")                        # (unset, forward calls with five tiles or more per workgroup take the GEMM kernel by themselves)
    base = run()
    monkeypatch.setenv("FB_C1G", "2")
    for _ in range(2):                                       # (twice: the rings and barriers leave no state behind)
        got = run()
        assert torch.equal(base[0], got[0]) and torch.equal(base[2], got[2])
        assert rel(base[1], got[1]) < 1e-5
    ref = x.reshape(-1, cin).float() @ w.reshape(cout, cin).float().t()
    assert rel(got[0].reshape(-1, cout).float(), ref) < tol(dt)
    assert rel(got[1][0].sum(0), ref.sum(0)) < 2e-3 and rel(got[1][1].sum(0), (ref * ref).sum(0)) < 1e-4


@pytest.mark.parametrize("cin,cout,hw,n", [(128, 512, 28, 512), (128, 512, 28, 2048), (256, 1024, 14, 2048), (1024, 256, 14, 2048), (512, 2048, 7, 2048), (2048, 512, 7, 2048),
                                           (64, 256, 56, 512), (256, 64, 56, 512), (512, 128, 28, 1024), (256, 512, 4, 12544)])
def test_conv1x1_default_dispatch_at_group_sizes(cin, cout, hw, n, monkeypatch):
    """What fb_conv2d's DEFAULT dispatch (no FB_C1G / FB_C1S_* in the environment) selects for the 1x1 layers of a ResNet-50 / -152 chunk group of 512-2048 images
    (and of the ResNet-18 shortcut at a 98-chunk group) against the kernels it replaces (FB_C1G=0 FB_C1S_PIPE=0: round-3 streaming kernel / implicit GEMM) and against torch's
    own matmul of the same bf16 operands: forward with statistics, input gradient, input gradient with addend.  Round 5's suite pinned FB_C1G in every full-size test,
    so the shipped default (the GEMM kernel for large K = 128 / 256 / >= 512 forward calls, several tiles per workgroup) was never compared with anything: its K = 128
    form multiplied wrong pixel rows in every tile behind a workgroup's first."""
    lib = _lib()
    for key in ("FB_C1G", "FB_C1S_PIPE", "FB_C1S_ADD_ASM"):
        monkeypatch.delenv(key, raising=False)
    torch.manual_seed(13)
    dt = torch.bfloat16
    x = torch.randn(n, hw, hw, cin, device="cuda").to(dt)
    w = (torch.randn(cout, 1, cin, device="cuda") * 0.05).to(dt)
    dy = torch.randn(n, hw, hw, cout, device="cuda").to(dt)
    wt = w.reshape(cout, cin).t().contiguous().reshape(cin, 1, cout)
    add = torch.randn(n, hw, hw, cin, device="cuda").to(dt)
    nblk = (n * hw * hw + 127) // 128

    def run():
        out, stat = torch.empty(n, hw, hw, cout, device="cuda", dtype=dt), torch.zeros(2, nblk, cout, device="cuda")
        dx0, dx1 = (torch.empty(n, hw, hw, cin, device="cuda", dtype=dt) for _ in range(2))
        lib.conv2d(x, w, out, 1, 1, 1, 0, 0, stat_partial=stat)
        lib.conv2d(dy, wt, dx0, 1, 1, 1, 0, 1)
        lib.conv2d(dy, wt, dx1, 1, 1, 1, 0, 1, addend=add, addend_mode=1)
        torch.cuda.synchronize()
        return out, stat, dx0, dx1

    got = run()
    monkeypatch.setenv("FB_C1G", "0"), monkeypatch.setenv("FB_C1S_PIPE", "0")
    base = run()
    assert torch.equal(got[0], base[0]) and torch.equal(got[2], base[2]) and torch.equal(got[3], base[3])
    assert rel(got[1], base[1]) < 1e-5
    ref = x.reshape(-1, cin).float() @ w.reshape(cout, cin).float().t()
    assert rel(got[0].reshape(-1, cout).float(), ref) < tol(dt)
    assert rel(got[1][0].sum(0), ref.sum(0)) < 2e-3 and rel(got[1][1].sum(0), (ref * ref).sum(0)) < 1e-4
    del ref
    refd = dy.reshape(-1, cout).float() @ w.reshape(cout, cin).float()
    assert rel(got[2].reshape(-1, cin).float(), refd) < tol(dt)
    assert rel(got[3].reshape(-1, cin).float(), refd + add.reshape(-1, cin).float()) < tol(dt)


@pytest.mark.parametrize("magnitude", [1.0, 3e-6, 4e4])
@pytest.mark.parametrize("cin,cout,k,stride,hw,n", [(64, 64, 3, 1, 8, 8), (64, 128, 3, 2, 8, 8), (128, 256, 1, 1, 4, 16), (64, 64, 3, 1, 32, 2),
                                                   (128, 128, 3, 1, 16, 3), (256, 128, 3, 1, 8, 4), (512, 512, 3, 1, 4, 32), (96, 64, 3, 1, 6, 3)])
def test_conv_f32_fp16x2_split(cin, cout, k, stride, hw, n, magnitude):
    """fp32 convolutions on the fp16 matrix pipe (fb_conv_args.amax_src / amax_wgt: two scaled fp16 pieces per operand, three MFMAs per
    product): forward and input gradient against float64, for activations / gradients of very different magnitudes (the
    per-tensor power-of-two scales come from fb_absmax; the weights are the fp16x2 planes fb_weight_prep writes).  22 significand bits per
    operand: the error stays within a small multiple of the six-product bf16 path's, and fb_absmax itself is exact."""
    lib = _lib()
    torch.manual_seed(3)
    pad = k // 2
    x = (torch.randn(n, cin, hw, hw) * torch.exp(torch.randn(n, cin, hw, hw))) * magnitude     # heavy-tailed, like gradients
    w = torch.randn(cout, cin, k, k) * 0.1
    am = torch.zeros(4, device="cuda")
    xd, master = nhwc(x).cuda(), krsc(w).cuda()
    if n % 2 == 0 and (n // 2) * hw * hw % 128 == 0:         # two "chunks" with a scale each, the second one 64x smaller
        x[n // 2:] /= 64
        xd = nhwc(x).cuda()
        am2 = torch.zeros(2, device="cuda")
        lib.call("fb_absmax", xd.data_ptr(), xd.numel() // 2, 2, xd.numel() // 2, 1, am2.data_ptr())
        assert torch.equal(am2.cpu(), x.reshape(2, -1).abs().max(1).values)
        src_scale = dict(amax_src=am2, amax_imgs=n // 2)
    else:
        src_scale = dict(amax_src=am[0:])
    lib.call("fb_absmax", xd.data_ptr(), xd.numel(), 1, 0, 0, am.data_ptr())
    lib.call("fb_absmax", master.data_ptr(), master.numel() // 2, 2, master.numel() // 2, 0, am.data_ptr() + 4)   # two slices = the two halves
    assert float(am[0]) == float(x.abs().max()) and float(am[1]) == float(w.abs().max())
    w_plain, wt_plain, w_pl, wt_pl = (torch.zeros(cout * k * k * cin, device="cuda") for _ in range(4))
    lib.weight_prep(master, 0, 0, 1, cout, k * k, cin, cin, w_plain, wt_plain, torch.float32)
    lib.weight_prep(master, 0, 0, 1, cout, k * k, cin, cin, w_pl, wt_pl, torch.float32, amax=am[1:])
    ref = F.conv2d(x.double(), w.double(), None, stride, pad)
    ho = ref.shape[2]
    out_h, out_s = (torch.empty(n, ho, ho, cout, device="cuda") for _ in range(2))
    lib.conv2d(xd, w_pl, out_h, k, k, stride, pad, 0, amax_wgt=am[1:], **src_scale)
    lib.conv2d(xd, w_plain, out_s, k, k, stride, pad, 0)
    eh, es = rel(nchw(out_h.cpu()).double(), ref), rel(nchw(out_s.cpu()).double(), ref)
    assert eh < 1.5e-6 and eh < 8 * es + 2e-7, (eh, es)
    if cin % 64 != 0:                               # (input gradients need Cd = cin in multiples of 64)
        return
    dy = (torch.randn(n, cout, ho, ho) * torch.exp(torch.randn(n, cout, ho, ho))) * magnitude
    refd = torch.nn.grad.conv2d_input((n, cin, hw, hw), w.double(), dy.double(), stride, pad)
    dyd = nhwc(dy).cuda()
    lib.call("fb_absmax", dyd.data_ptr(), dyd.numel(), 1, 0, 0, am.data_ptr() + 8)
    dh, ds = (torch.empty(n, hw, hw, cin, device="cuda") for _ in range(2))
    lib.conv2d(dyd, wt_pl, dh, k, k, stride, pad, 1, amax_src=am[2:], amax_wgt=am[1:])
    lib.conv2d(dyd, wt_plain, ds, k, k, stride, pad, 1)
    eh, es = rel(nchw(dh.cpu()).double(), refd), rel(nchw(ds.cpu()).double(), refd)
    assert eh < 1.5e-6 and eh < 8 * es + 2e-7, (eh, es)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,k,stride,hw,ipg,groups,split", [(64, 64, 3, 1, 8, 8, 2, 1), (64, 64, 3, 1, 8, 8, 2, 3),
                                                                  (64, 128, 3, 2, 8, 8, 1, 2), (256, 256, 3, 1, 4, 16, 2, 1),
                                                                  (128, 256, 1, 1, 4, 16, 1, 2), (32, 64, 1, 1, 8, 4, 2, 1),
                                                                  (128, 128, 3, 1, 8, 4, 1, 1),
                                                                  (64, 64, 3, 1, 32, 4, 2, 2), (128, 64, 3, 1, 16, 4, 2, 4),   # all-taps halo kernel
                                                                  (64, 128, 3, 1, 16, 6, 1, 3),
                                                                  (128, 64, 3, 1, 8, 6, 2, 2), (64, 128, 3, 1, 4, 8, 2, 2),    # 8x8 / 4x4 maps
                                                                  (64, 64, 3, 1, 4, 6, 1, 3), (64, 64, 3, 1, 4, 6, 1, 2),
                                                                  (128, 64, 3, 1, 4, 12, 2, 2), (64, 64, 3, 1, 4, 20, 1, 3), (512, 512, 3, 1, 4, 128, 2, 1),   # 4x4, compact halo (chunks of 4k images)
                                                                  (64, 128, 3, 2, 32, 4, 2, 2), (128, 64, 3, 2, 16, 6, 1, 3), (64, 64, 3, 2, 8, 8, 2, 2),   # stride 2, all taps
                                                                  (128, 128, 3, 2, 16, 5, 1, 1),
                                                                  (64, 64, 3, 1, 16, 10, 2, 3), (64, 64, 3, 1, 32, 7, 1, 4), (64, 64, 3, 1, 4, 10, 2, 3), (64, 128, 3, 2, 16, 7, 2, 5),   # ragged K slices
                                                                  # ImageNet-shaped maps (56 / 28 / 14): all-taps kernel with image rows padded to 32-pixel blocks
                                                                  (64, 64, 3, 1, 56, 2, 2, 1), (128, 128, 3, 1, 28, 4, 2, 3), (256, 256, 3, 1, 14, 8, 2, 3), (64, 128, 3, 1, 14, 7, 1, 2),
                                                                  (128, 64, 3, 1, 28, 3, 2, 1),
                                                                  # 1x1 weight gradients: large-tile LDS-DMA kernel (bf16), every wave-tile variant, ragged pixel ranges, empty last slice
                                                                  (256, 1024, 1, 1, 7, 4, 2, 3), (1024, 256, 1, 1, 7, 4, 1, 2), (64, 256, 1, 1, 8, 3, 2, 2), (256, 64, 1, 1, 8, 3, 1, 1),
                                                                  (128, 128, 1, 1, 8, 5, 2, 3), (64, 128, 1, 1, 8, 4, 1, 2), (128, 64, 1, 1, 6, 3, 2, 2), (512, 128, 1, 1, 14, 2, 2, 5),
                                                                  (256, 512, 1, 1, 4, 4, 1, 8)])
def test_conv_wgrad(dtype, cin, cout, k, stride, hw, ipg, groups, split, monkeypatch):
    lib = _lib()
    torch.manual_seed(2)
    pad = k // 2
    n = ipg * groups
    ho = (hw + 2 * pad - k) // stride + 1
    x = q(torch.randn(n, cin, hw, hw), dtype)
    dy = q(torch.randn(n, cout, ho, ho), dtype)
    xd, dyd = nhwc(x).to(dtype).cuda(), nhwc(dy).to(dtype).cuda()
    slab = torch.full((groups, split, cout, k * k, cin), float("nan"), device="cuda")
    lib.conv2d_wgrad(xd, dyd, slab, k, k, stride, pad, ipg, split)
    out = torch.zeros(groups, cout * k * k * cin + 8, device="cuda")
    lib.wgrad_reduce(slab, out, out.shape[1], groups, split, cout, k * k, cin, cin)
    for g in range(groups):
        sl = slice(g * ipg, (g + 1) * ipg)
        ref = torch.nn.grad.conv2d_weight(x[sl], (cout, cin, k, k), dy[sl], stride, pad)
        got = out[g, : cout * k * k * cin].view(cout, k, k, cin).permute(0, 3, 1, 2).cpu()
        assert rel(got, ref) < (1e-5 if dtype == torch.float32 else 1e-5), (g, rel(got, ref))
    if hw == 4 and k == 3 and stride == 1 and ipg % 4 == 0 and dtype == torch.bfloat16:
        # the padded 4x4 layout (image pairs per K-step) adds the same products in the same order plus zeros: same bits
        monkeypatch.setenv("FB_WGRAD3_COMPACT", "0")
        slab2 = torch.full_like(slab, float("nan"))
        lib.conv2d_wgrad(xd, dyd, slab2, k, k, stride, pad, ipg, split)
        monkeypatch.delenv("FB_WGRAD3_COMPACT")
        if ((ipg + split - 1) // split) % 4 == 0:          # same K slices in both layouts
            assert torch.equal(slab, slab2)
    if k == 1 and dtype == torch.bfloat16 and (cin % 256 == 0 or cout % 256 == 0):
        # 1x1 weight gradients: 128-channel tiles per side by default (two workgroups per CU), 256-channel tiles with FB_W1_CAP_M / _N = 8 -- every output element adds the
        # same pixels up in the same order: same bits
        monkeypatch.setenv("FB_W1_CAP_M", "8"), monkeypatch.setenv("FB_W1_CAP_N", "8")
        slab2 = torch.full_like(slab, float("nan"))
        lib.conv2d_wgrad(xd, dyd, slab2, k, k, stride, pad, ipg, split)
        monkeypatch.delenv("FB_W1_CAP_M"), monkeypatch.delenv("FB_W1_CAP_N")
        assert torch.equal(slab, slab2)
    if split == 1:      # group_stride: per-chunk gradients written straight into arena rows, no reduce pass
        arena = torch.full((groups, cout * k * k * cin + 40), float("nan"), device="cuda")
        lib.conv2d_wgrad(xd, dyd, arena[:, 8:], k, k, stride, pad, ipg, 1, group_stride=arena.shape[1])
        assert torch.equal(arena[:, 8:8 + cout * k * k * cin], out[:, : cout * k * k * cin])
        assert bool(torch.isnan(arena[:, :8]).all()) and bool(torch.isnan(arena[:, 8 + cout * k * k * cin:]).all())


def test_mt_accumulate_skip_and_sum():
    """fb_mt_accumulate_skip = fb_mt_accumulate with up to four ranges left alone (mean untouched there, norms without them);
    fb_mt_accumulate_sum = the same recurrence advanced by a group's chunks at once from their sum."""
    lib = _lib()
    torch.manual_seed(3)
    P, G, c0 = 4096 + 64, 5, 3
    g = torch.randn(G, P, device="cuda")
    avg0 = torch.randn(P, device="cuda")
    ws = torch.zeros(int(lib.load().fb_ws_mt_floats(G)), device="cuda")
    ref, sq_ref = avg0.clone(), torch.zeros(G, device="cuda")
    lib.call("fb_mt_accumulate", ref.data_ptr(), g.data_ptr(), P, G, P, c0, sq_ref.data_ptr(), ws.data_ptr())
    skips = [(128, 640), (1024, 1028), (4000, 4160)]
    got, sq = avg0.clone(), torch.zeros(G, device="cuda")
    flat = [v for r in skips for v in r] + [0, 0]
    lib.call("fb_mt_accumulate_skip", got.data_ptr(), g.data_ptr(), P, G, P, c0, sq.data_ptr(), ws.data_ptr(), *flat)
    keep = torch.ones(P, dtype=torch.bool, device="cuda")
    for a, b in skips:
        keep[a:b] = False
    assert torch.equal(got[keep], ref[keep]) and torch.equal(got[~keep], avg0[~keep])
    want = (g.double() ** 2 * keep).sum(1)
    assert float(((sq.double() - want).abs() / want).max()) < 1e-6
    # the skipped ranges from the group sum: what G steps of the recurrence amount to
    for a, b in skips:
        gs = g[:, a:b].sum(0).contiguous()
        part = avg0[a:b].clone()
        lib.call("fb_mt_accumulate_sum", part.data_ptr(), gs.data_ptr(), b - a, c0, G)
        assert float((part.double() - ref[a:b].double()).abs().max()) < 2e-6 * float(ref[a:b].abs().max() + 1)


@pytest.mark.parametrize("cin,cout,ipg,chunks,chains", [(64, 64, 8, 5, 2), (128, 64, 16, 7, 3), (64, 128, 4, 6, 6), (512, 512, 128, 9, 4), (256, 128, 12, 3, 1)])
def test_conv_wgrad_chunk_chain(cin, cout, ipg, chunks, chains):
    """fb_conv2d_wgrad_chain (ABI v12): the sum over the chunks of the per-chunk weight gradients and every chunk's sum of squares, against the
    per-chunk kernel (whose chunk gradients are held to torch by test_conv_wgrad): same products, the sum taken in another order."""
    lib = _lib()
    torch.manual_seed(5)
    n = ipg * chunks
    x = q(torch.randn(n, cin, 4, 4), torch.bfloat16)
    dy = q(torch.randn(n, cout, 4, 4), torch.bfloat16)
    xd, dyd = nhwc(x).to(torch.bfloat16).cuda(), nhwc(dy).to(torch.bfloat16).cuda()
    per = torch.full((chunks, 1, cout, 9, cin), float("nan"), device="cuda")
    lib.conv2d_wgrad(xd, dyd, per, 3, 3, 1, 1, ipg, 1)
    per = per[:, 0].double()
    a = lib.WgradArgs(xd.data_ptr(), dyd.data_ptr(), None, n, 4, 4, cin, 4, 4, cout, 3, 3, 1, 1, ipg, 1, lib.dtype_code(torch.bfloat16), 0)
    assert lib.load().fb_wgrad_chain_supported(lib.C.byref(a))
    tiles = (cout // 64) * (cin // 64)
    slabs = torch.full((chains, cout, 9, cin), float("nan"), device="cuda")
    sqp = torch.full((chunks, tiles, 8), float("nan"), device="cuda")
    lib.call("fb_conv2d_wgrad_chain", lib.C.byref(a), chains, slabs.data_ptr(), sqp.data_ptr())
    total = torch.full((cout, 9, cin), float("nan"), device="cuda")
    lib.wgrad_reduce(slabs, total, 0, 1, chains, cout, 9, cin, cin)
    want = per.sum(0)
    assert float((total.double() - want).norm() / want.norm()) < 2e-6
    sq = sqp.double().sum((1, 2))
    ref = per.pow(2).sum((1, 2, 3))
    assert float(((sq - ref).abs() / ref).max()) < 1e-5, (sq, ref)
    # a chain's slab = the sum of ITS chunks
    for s_ in range(chains):
        w = per[s_::chains].sum(0)
        assert float((slabs[s_].double() - w).norm() / w.norm()) < 2e-6


@pytest.mark.parametrize("magnitude", [1.0, 2e-6])
@pytest.mark.parametrize("cin,cout,k,stride,hw,ipg,groups,split", [(64, 64, 3, 1, 8, 8, 2, 3), (64, 128, 3, 2, 8, 8, 1, 2), (256, 256, 3, 1, 4, 16, 2, 1),
                                                                  (128, 256, 1, 1, 4, 16, 1, 2), (32, 64, 1, 1, 8, 4, 2, 1), (64, 64, 3, 1, 32, 4, 2, 2),
                                                                  (128, 64, 3, 1, 16, 4, 2, 4), (64, 64, 3, 1, 16, 10, 2, 3), (128, 64, 3, 1, 8, 6, 2, 2), (256, 256, 3, 1, 8, 16, 1, 1), (64, 128, 3, 1, 4, 8, 2, 2), (64, 64, 3, 1, 4, 6, 1, 3), (512, 512, 3, 1, 4, 16, 2, 1),
                                                                  (64, 128, 3, 2, 32, 4, 2, 2), (128, 64, 3, 2, 16, 6, 1, 3), (64, 64, 3, 2, 8, 8, 2, 2), (128, 128, 3, 2, 16, 5, 1, 1)])   # stride 2, all taps
def test_conv_wgrad_f32_fp16x2_split(cin, cout, k, stride, hw, ipg, groups, split, magnitude):
    """Weight gradients of fp32 tensors on the fp16 matrix pipe (fb_wgrad_args.amax_x / amax_dy): two scaled fp16 planes per operand, three
    MFMAs per product; against float64, with gradients of realistic (tiny) magnitude, next to the six-product bf16 path."""
    lib = _lib()
    torch.manual_seed(4)
    pad = k // 2
    n = ipg * groups
    ho = (hw + 2 * pad - k) // stride + 1
    x = torch.randn(n, cin, hw, hw) * torch.exp(0.5 * torch.randn(n, cin, hw, hw))
    dy = torch.randn(n, cout, ho, ho) * torch.exp(torch.randn(n, cout, ho, ho)) * magnitude
    xd, dyd = nhwc(x).cuda(), nhwc(dy).cuda()
    am = torch.zeros(2, groups, device="cuda")                # one scale per chunk (group) and operand
    lib.call("fb_absmax", xd.data_ptr(), xd.numel() // groups, groups, xd.numel() // groups, 1, am[0].data_ptr())
    lib.call("fb_absmax", dyd.data_ptr(), dyd.numel() // groups, groups, dyd.numel() // groups, 1, am[1].data_ptr())
    assert torch.equal(am[0].cpu(), x.reshape(groups, -1).abs().max(1).values) and torch.equal(am[1].cpu(), dy.reshape(groups, -1).abs().max(1).values)
    errs = []
    for amax in ((am[0], am[1]), (None, None)):
        slab = torch.full((groups, split, cout, k * k, cin), float("nan"), device="cuda")
        lib.conv2d_wgrad(xd, dyd, slab, k, k, stride, pad, ipg, split, amax_x=amax[0], amax_dy=amax[1])
        out = torch.zeros(groups, cout * k * k * cin, device="cuda")
        lib.wgrad_reduce(slab, out, out.shape[1], groups, split, cout, k * k, cin, cin)
        e = 0.0
        for g in range(groups):
            sl = slice(g * ipg, (g + 1) * ipg)
            ref = torch.nn.grad.conv2d_weight(x[sl].double(), (cout, cin, k, k), dy[sl].double(), stride, pad)
            e = max(e, rel(out[g].view(cout, k, k, cin).permute(0, 3, 1, 2).cpu().double(), ref))
        errs.append(e)
    assert errs[0] < 1.5e-6 and errs[0] < 8 * errs[1] + 2e-7, errs


@pytest.mark.parametrize("dtype", DTYPES)
def test_weight_prep(dtype):
    lib = _lib()
    torch.manual_seed(3)
    co, taps, ci, cip, sets = 64, 9, 27, 32, 2
    master = torch.randn(sets, 4000 + co * taps * ci).cuda()
    wf = torch.zeros(sets, co * taps * cip + 64, dtype=dtype, device="cuda")
    wd = torch.zeros_like(wf)
    lib.call("fb_weight_prep", master.data_ptr() + 4 * 100, master.shape[1], wf.shape[1], sets, co, taps, ci, cip, wf.data_ptr(),
             wd.data_ptr(), lib.dtype_code(dtype), None)
    for s in range(sets):
        m = master[s, 100:100 + co * taps * ci].view(co, taps, ci).cpu()
        ref = torch.zeros(co, taps, cip)
        ref[..., :ci] = m
        assert torch.equal(wf[s, : co * taps * cip].float().cpu().view(co, taps, cip), q(ref, dtype))
        assert torch.equal(wd[s, : co * taps * cip].float().cpu().view(cip, taps, co), q(ref.permute(2, 1, 0).contiguous(), dtype))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,hw,ipg,groups", [(64, 8, 8, 2), (512, 4, 16, 3)])
def test_batchnorm_fwd_bwd(dtype, C, hw, ipg, groups):
    """conv-epilogue statistics are emulated with a 1x1 identity-free path: partial sums are produced on the host here,
    the finalize/apply/backward kernels are what is under test."""
    lib = _lib()
    torch.manual_seed(4)
    n = ipg * groups
    x = q(torch.randn(n, C, hw, hw) * 1.5 + 0.3, dtype)
    res = q(torch.randn(n, C, hw, hw), dtype)
    gamma, beta = torch.rand(groups, C) + 0.5, torch.randn(groups, C) * 0.1
    dout = q(torch.randn(n, C, hw, hw), dtype)
    xn = nhwc(x).reshape(-1, C)
    px = xn.shape[0]
    nblk = px // 128
    part = torch.stack([xn.view(nblk, 128, C).sum(1), (xn * xn).view(nblk, 128, C).sum(1)]).cuda()
    ch_total, ch_off = C + 96, 32
    mean_tab = torch.zeros(groups, ch_total, device="cuda")
    var_tab = torch.zeros_like(mean_tab)
    scale, shift, invstd = (torch.zeros(groups, C, device="cuda") for _ in range(3))
    pg = torch.zeros(groups, 2 * C + 64)
    pg[:, :C], pg[:, C:2 * C] = gamma, beta
    pgd = pg.cuda()
    ppg = ipg * hw * hw
    lib.call("fb_bn_fwd_finalize", part.data_ptr(), nblk, groups, C, float(ppg), pgd.data_ptr(), pgd.data_ptr() + 4 * C, pg.shape[1], 1e-5,
             mean_tab.data_ptr(), var_tab.data_ptr(), ch_total, ch_off, scale.data_ptr(), shift.data_ptr(), invstd.data_ptr())
    xd, resd, doutd = (nhwc(t).to(dtype).cuda() for t in (x, res, dout))
    y = torch.empty_like(xd)
    bits = torch.zeros(xd.numel() * xd.element_size() // 16, dtype=torch.uint8, device="cuda")
    amax = torch.zeros(2, groups, device="cuda")             # fp32 only: per-group largest magnitudes tracked by the apply passes themselves
    amax_ws = torch.zeros(int(lib.load().fb_ws_bn_amax_floats(px, C, ppg)), device="cuda")
    lib.call("fb_bn_apply", xd.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), resd.data_ptr(), None, None, px, C, ppg, 0, 1,
             bits.data_ptr(), None, 0, lib.dtype_code(dtype), *((amax[0].data_ptr(), amax_ws.data_ptr()) if dtype == torch.float32 else (None, None)))
    # reference, per group
    ys, dxs, dgs, dbs = [], [], [], []
    for g in range(groups):
        sl = slice(g * ipg, (g + 1) * ipg)
        xg = x[sl].double()
        mean, var = xg.mean((0, 2, 3)), xg.var((0, 2, 3), unbiased=False)
        assert rel(mean_tab[g, ch_off:ch_off + C].cpu(), mean) < 1e-5
        assert rel(var_tab[g, ch_off:ch_off + C].cpu(), var) < 1e-5
        xhat = (xg - mean[None, :, None, None]) / (var + 1e-5).sqrt()[None, :, None, None]
        yg = torch.relu(xhat * gamma[g].double()[None, :, None, None] + beta[g].double()[None, :, None, None] + res[sl].double())
        ys.append(yg)
        dy = dout[sl].double() * (yg > 0)
        m = ipg * hw * hw
        db, dg = dy.sum((0, 2, 3)), (dy * xhat).sum((0, 2, 3))
        dx = (gamma[g].double() / (var + 1e-5).sqrt())[None, :, None, None] * (dy - db[None, :, None, None] / m - xhat * dg[None, :, None, None] / m)
        dxs.append(dx), dgs.append(dg), dbs.append(db)
    yref = torch.cat(ys)
    assert rel(nchw(y.float().cpu()), yref) < tol(dtype, 0.5)
    # backward uses the *reference* y as mask to avoid sign flips of near-zero outputs in bf16
    yd = nhwc(yref.float()).to(dtype).cuda()
    part2 = torch.zeros(2, nblk, C, device="cuda")
    # ReLU bitmask (1 byte per 16-byte vector) must encode y > 0; the backward accepts either y or the bitmask
    vec = 16 // xd.element_size()
    ybits = ((y.float().reshape(-1, vec) > 0).to(torch.int32) << torch.arange(vec, device="cuda")).sum(1).to(torch.uint8)
    assert torch.equal(bits, ybits)
    ref_bits = ((yd.float().reshape(-1, vec) > 0).to(torch.int32) << torch.arange(vec, device="cuda")).sum(1).to(torch.uint8)
    lib.call("fb_bn_bwd_reduce", doutd.data_ptr(), None, ref_bits.data_ptr(), xd.data_ptr(), mean_tab.data_ptr(), invstd.data_ptr(), ch_total, ch_off,
             part2.data_ptr(), px, C, ppg, lib.dtype_code(dtype))
    gout = torch.zeros(groups, 2 * C + 64, device="cuda")
    coef = torch.zeros(groups, C, 3, device="cuda")
    lib.call("fb_bn_bwd_finalize", part2.data_ptr(), lib.load().fb_bn_bwd_reduce_rows(px, ppg), groups, C, float(ppg), scale.data_ptr(), mean_tab.data_ptr(), invstd.data_ptr(),
             ch_total, ch_off, gout.data_ptr(), gout.data_ptr() + 4 * C, gout.shape[1], coef.data_ptr(), 0)
    dx = torch.empty_like(xd)
    dy_out = torch.empty_like(xd)
    lib.call("fb_bn_bwd_apply", doutd.data_ptr(), yd.data_ptr(), None, xd.data_ptr(), coef.data_ptr(), dx.data_ptr(), dy_out.data_ptr(), px, C, ppg,
             lib.dtype_code(dtype), *((amax[1].data_ptr(), amax_ws.data_ptr()) if dtype == torch.float32 else (None, None)))
    if dtype == torch.float32:
        assert torch.equal(amax[0], y.reshape(groups, -1).abs().max(1).values) and torch.equal(amax[1], dx.reshape(groups, -1).abs().max(1).values)
    assert rel(gout[:, :C].cpu(), torch.stack(dgs)) < 1e-4
    assert rel(gout[:, C:2 * C].cpu(), torch.stack(dbs)) < 1e-4
    assert rel(nchw(dx.float().cpu()), torch.cat(dxs)) < tol(dtype, 0.5)
    assert rel(nchw(dy_out.float().cpu()), dout.double() * (yref > 0)) < tol(dtype, 0.5)


def test_bn_running_update():
    lib = _lib()
    torch.manual_seed(5)
    G, ch = 3, 200
    rm, rv = torch.randn(ch), torch.rand(ch) + 0.5
    m0, v0, m1, v1 = torch.randn(G, ch), torch.rand(G, ch), torch.randn(G, ch), torch.rand(G, ch)
    ub = torch.full((ch,), 128.0 / 127.0)
    d = [t.cuda() for t in (rm, rv, torch.stack([m0, m1]), torch.stack([v0, v1]), ub)]
    lib.call("fb_bn_running_update", d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 2, G * ch, d[4].data_ptr(), G, ch, 0.1)
    erm, erv = rm.clone(), rv.clone()
    for g in range(G):
        for m, v in ((m0, v0), (m1, v1)):
            erm = 0.9 * erm + 0.1 * m[g]
            erv = 0.9 * erv + 0.1 * (v[g] * ub)
    assert torch.allclose(d[0].cpu(), erm, rtol=1e-6, atol=1e-7) and torch.allclose(d[1].cpu(), erv, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hw,n,C", [(8, 3, 64), (7, 2, 64), (14, 5, 128), (112, 2, 64)])
def test_maxpool_backward_with_ties(dtype, hw, n, C):
    """MaxPool2d(3, 2, 1) backward on POST-ReLU input (exact zeros: whole windows tie, the first maximum in row-major window order takes the
    gradient, torch's CPU semantics), even and odd map sizes, the ImageNet stem's 112x112."""
    lib = _lib()
    torch.manual_seed(7)
    x = q(torch.relu(torch.randn(n, C, hw, hw) - 0.3), dtype)
    xr = x.clone().requires_grad_(True)
    mp = F.max_pool2d(xr, 3, 2, 1)
    dmp = q(torch.randn_like(mp), dtype)
    mp.backward(dmp)
    xd = nhwc(x).to(dtype).cuda()
    dxm = torch.full_like(xd, float("nan"))
    lib.call("fb_maxpool3s2_bwd", xd.data_ptr(), nhwc(dmp).to(dtype).cuda().data_ptr(), dxm.data_ptr(), n, hw, hw, C, lib.dtype_code(dtype))
    got = nchw(dxm.float().cpu())
    assert bool(torch.isfinite(got).all())
    assert rel(got, xr.grad) < tol(dtype, 0.5)
    assert float((got - q(xr.grad, dtype)).abs().max()) <= (0 if dtype == torch.float32 else 2 ** -7 * float(xr.grad.abs().max()))
    # the form the engine runs (ABI v13): the forward pass remembers every window's argmax as one byte, the backward pass reads it instead of the pre-pool
    # tensor -- the same outputs and the same input gradient BIT FOR BIT, and the remembered positions are torch's own max_pool2d_with_indices
    ho = (hw + 1) // 2
    y0, y1 = (torch.empty(n, ho, ho, C, dtype=dtype, device="cuda") for _ in range(2))
    idx = torch.full((n, ho, ho, C), 255, dtype=torch.uint8, device="cuda")
    lib.call("fb_maxpool3s2_fwd", xd.data_ptr(), y0.data_ptr(), n, hw, hw, C, lib.dtype_code(dtype))
    lib.call("fb_maxpool3s2_fwd_idx", xd.data_ptr(), y1.data_ptr(), idx.data_ptr(), n, hw, hw, C, lib.dtype_code(dtype))
    dxi = torch.full_like(xd, float("nan"))
    lib.call("fb_maxpool3s2_bwd_idx", idx.data_ptr(), nhwc(dmp).to(dtype).cuda().data_ptr(), dxi.data_ptr(), n, hw, hw, C, lib.dtype_code(dtype))
    assert torch.equal(y0, y1) and torch.equal(dxi, dxm)
    _, tidx = F.max_pool2d(x, 3, 2, 1, return_indices=True)                 # flat index into the hw x hw plane
    pos = nchw(idx.cpu()).long()
    oy = torch.arange(ho).view(1, 1, ho, 1)
    ox = torch.arange(ho).view(1, 1, 1, ho)
    flat = (2 * oy - 1 + pos // 3) * hw + (2 * ox - 1 + pos % 3)
    assert int(pos.max()) <= 8 and torch.equal(flat, tidx)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pools_and_head(dtype):
    lib = _lib()
    torch.manual_seed(6)
    n, C, hw, classes, ipg = 8, 64, 8, 10, 4
    groups = n // ipg
    x = q(torch.randn(n, C, hw, hw), dtype)
    xd = nhwc(x).to(dtype).cuda()
    y = torch.empty(n, hw // 2, hw // 2, C, dtype=dtype, device="cuda")
    lib.call("fb_avgpool2_fwd", xd.data_ptr(), y.data_ptr(), n, hw, hw, C, lib.dtype_code(dtype))
    assert rel(nchw(y.float().cpu()), F.avg_pool2d(x, 2)) < tol(dtype, 0.5)
    ym = torch.empty(n, hw // 2, hw // 2, C, dtype=dtype, device="cuda")
    lib.call("fb_maxpool3s2_fwd", xd.data_ptr(), ym.data_ptr(), n, hw, hw, C, lib.dtype_code(dtype))
    xr = x.clone().requires_grad_(True)
    mp = F.max_pool2d(xr, 3, 2, 1)
    assert torch.equal(nchw(ym.float().cpu()), mp.detach())
    dmp = q(torch.randn_like(mp), dtype)
    mp.backward(dmp)
    dxm = torch.empty_like(xd)
    lib.call("fb_maxpool3s2_bwd", xd.data_ptr(), nhwc(dmp).to(dtype).cuda().data_ptr(), dxm.data_ptr(), n, hw, hw, C, lib.dtype_code(dtype))
    assert rel(nchw(dxm.float().cpu()), xr.grad) < tol(dtype, 0.5)
    # head
    feat = torch.empty(n, C, device="cuda")
    lib.call("fb_head_pool", xd.data_ptr(), feat.data_ptr(), n, hw * hw, C, lib.dtype_code(dtype))
    fref = x.mean((2, 3))
    assert rel(feat.cpu(), fref) < 1e-5
    P = classes * C + 16 + 64
    theta = torch.randn(groups, P) * 0.2
    labels = torch.randint(0, classes, (n,))
    thd, lab = theta.cuda(), labels.cuda()
    logits, dlogits = torch.empty(n, classes, device="cuda"), torch.empty(n, classes, device="cuda")
    loss, correct = torch.empty(groups, device="cuda"), torch.empty(groups, device="cuda")
    boff = classes * C
    lib.call("fb_head_loss", feat.data_ptr(), thd.data_ptr(), thd.data_ptr() + 4 * boff, P, lab.data_ptr(), logits.data_ptr(),
             dlogits.data_ptr(), loss.data_ptr(), correct.data_ptr(), groups, ipg, C, classes, 0.0, 0)
    gout = torch.zeros(groups, P, device="cuda")
    d_a = torch.empty(n, hw, hw, C, dtype=dtype, device="cuda")
    lib.call("fb_head_bwd", feat.data_ptr(), dlogits.data_ptr(), thd.data_ptr(), P, gout.data_ptr(), gout.data_ptr() + 4 * boff, P,
             d_a.data_ptr(), groups, ipg, hw * hw, C, classes, lib.dtype_code(dtype))
    for g in range(groups):
        sl = slice(g * ipg, (g + 1) * ipg)
        W = theta[g, :boff].view(classes, C).clone().requires_grad_(True)
        b = theta[g, boff:boff + classes].clone().requires_grad_(True)
        f = feat[sl].cpu().clone().requires_grad_(True)
        z = f @ W.t() + b
        l = F.cross_entropy(z, labels[sl])
        l.backward()
        assert abs(float(loss[g]) - float(l)) < 1e-5
        assert float(correct[g]) == float((z.argmax(-1) == labels[sl]).sum())
        assert rel(gout[g, :boff].cpu(), W.grad.reshape(-1)) < 1e-5
        assert rel(gout[g, boff:boff + classes].cpu(), b.grad) < 1e-5
        dref = (f.grad / (hw * hw))[:, :, None, None].expand(-1, -1, hw, hw)
        assert rel(nchw(d_a[sl].float().cpu()), dref) < tol(dtype, 0.5)


def test_multi_tensor_ops():
    lib = _lib()
    torch.manual_seed(7)
    P, G = 100_003 + 1, 11          # deliberately not a multiple of 4 per tensor; group stride padded
    stride = (P + 63) // 64 * 64
    g = torch.randn(G, stride)
    g[:, P:] = 0
    gd = g.cuda()
    ws = torch.zeros(max(G, 2) * lib.MT_BLOCKS, device="cuda")
    out = torch.zeros(G, device="cuda")
    lib.call("fb_mt_sqnorm", gd.data_ptr(), stride, G, P, 0.5, None, 0.0, out.data_ptr(), ws.data_ptr())
    assert torch.allclose(out.cpu(), (0.5 * g[:, :P]).double().pow(2).sum(1).float(), rtol=1e-5)
    # running mean + fused norms
    avg = torch.randn(stride)
    avgd = avg.cuda()
    sq = torch.zeros(G, device="cuda")
    lib.call("fb_mt_accumulate", avgd.data_ptr(), gd.data_ptr(), stride, G, P, 5, sq.data_ptr(), ws.data_ptr())
    ref = avg[:P].clone()
    for j in range(G):
        ref = ref + (g[j, :P] - ref) * torch.tensor(1.0 / (5 + j + 1), dtype=torch.float32)
    assert torch.allclose(avgd[:P].cpu(), ref, rtol=1e-6, atol=1e-7)
    assert torch.allclose(sq.cpu(), g[:, :P].double().pow(2).sum(1).float(), rtol=1e-5)
    # FD perturb / combine
    theta0 = torch.randn(stride)
    th0 = theta0.cuda()
    vn = (0.5 * g[:, :P]).double().pow(2).sum(1).float().cuda()
    eps_n = torch.zeros(G, device="cuda")
    thk = torch.zeros(G, stride, device="cuda")
    lib.call("fb_mt_fd_perturb", th0.data_ptr(), gd.data_ptr(), stride, G, P, 0.5, 1e-2, 1.0, vn.data_ptr(), eps_n.data_ptr(), None, 0.0, thk.data_ptr())
    en = 1e-2 / vn.cpu().sqrt()
    assert torch.allclose(eps_n.cpu(), en, rtol=1e-6)
    assert torch.allclose(thk[:, :P].cpu(), theta0[None, :P] + en[:, None] * (0.5 * g[:, :P]), rtol=1e-6, atol=1e-7)
    # acc_strength variants: direction v = 0.5*g + 0.3*pre (pre shared by the groups)
    pre = torch.randn(stride)
    pre[P:] = 0
    pred = pre.cuda()
    lib.call("fb_mt_sqnorm", gd.data_ptr(), stride, G, P, 0.5, pred.data_ptr(), 0.3, out.data_ptr(), ws.data_ptr())
    v = 0.5 * g[:, :P] + 0.3 * pre[None, :P]
    assert torch.allclose(out.cpu(), v.double().pow(2).sum(1).float(), rtol=1e-5)
    eps_v = torch.zeros(G, device="cuda")
    lib.call("fb_mt_fd_perturb", th0.data_ptr(), gd.data_ptr(), stride, G, P, 0.5, 1e-2, -0.5, out.data_ptr(), eps_v.data_ptr(), pred.data_ptr(), 0.3,
             thk.data_ptr())
    env = 1e-2 / out.cpu().sqrt()
    assert torch.allclose(thk[:, :P].cpu(), theta0[None, :P] - 0.5 * env[:, None] * v, rtol=1e-6, atol=1e-7)
    g2 = g + 0.01 * torch.randn(G, stride)
    g2d = g2.cuda()
    avg2 = torch.zeros(stride, device="cuda")
    lib.call("fb_mt_fd_combine_accumulate", avg2.data_ptr(), gd.data_ptr(), g2d.data_ptr(), gd.data_ptr(), stride, G, P, eps_n.data_ptr(), 0.2, 0)
    ref = torch.zeros(P)
    for j in range(G):
        gt = g[j, :P] + 0.2 * ((g2[j, :P] - g[j, :P]) / en[j])
        ref = ref + (gt - ref) * torch.tensor(1.0 / (j + 1), dtype=torch.float32)
    assert torch.allclose(avg2[:P].cpu(), ref, rtol=1e-4, atol=1e-3)   # (g2-g)/eps_n amplifies fp32 roundoff by 1/eps_n ~ 1e4
    # norms + clip + SGD
    theta, grad, mom = torch.randn(P), torch.randn(P) * 0.01, torch.randn(P) * 0.01
    td, grd, md = theta.cuda(), grad.cuda(), mom.cuda()
    n2 = torch.zeros(2, device="cuda")
    lib.call("fb_mt_norms2", grd.data_ptr(), td.data_ptr(), P, n2.data_ptr(), ws.data_ptr())
    assert torch.allclose(n2.cpu(), torch.stack([grad.double().pow(2).sum(), theta.double().pow(2).sum()]).float(), rtol=1e-5)
    for first in (1, 0):
        lib.call("fb_mt_clip_sgd", td.data_ptr(), grd.data_ptr(), md.data_ptr(), P, n2.data_ptr(), 0.25, 0.1, 5e-4, 0.9, 0.0, 1, first)
        norm = n2[0].sqrt().cpu()
        coef = (0.25 / (norm + 1e-6)) if norm > 0.25 else torch.tensor(1.0)
        grad = grad * coef
        d = grad + 5e-4 * theta
        mom = d.clone() if first else 0.9 * mom + d
        theta = theta - 0.1 * (d + 0.9 * mom)
        assert torch.allclose(td.cpu(), theta, rtol=1e-5, atol=1e-6) and torch.allclose(md.cpu(), mom, rtol=1e-5, atol=1e-7)
        assert torch.allclose(grd.cpu(), grad, rtol=1e-6, atol=1e-9)
        lib.call("fb_mt_norms2", grd.data_ptr(), td.data_ptr(), P, n2.data_ptr(), ws.data_ptr())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,stride,pad,hw,cin_pad,aug", [(3, 1, 1, 32, 32, False), (3, 1, 1, 32, 32, True), (7, 2, 3, 64, 160, False),
                                                         (3, 1, 1, 16, 64, True)])
def test_stem_patches_and_device_augmentation(dtype, k, stride, pad, hw, cin_pad, aug):
    """fb_stem_patches vs torch: unfold of the (augmented) image; augmentation = torchvision's RandomCrop(hw, 4) on the image
    padded with black (-mean/std after normalisation), then RandomHorizontalFlip, with given per-image draws."""
    lib = _lib()
    torch.manual_seed(12)
    n, c, cp = 9, 3, 4
    x = torch.randn(n, c, hw, hw)
    pv = [-0.49 / 0.25, -0.48 / 0.24, -0.45 / 0.26]
    ref_img = x
    oy = ox = fl = None
    if aug:
        oy, ox = torch.randint(0, 2 * cp + 1, (2, n), dtype=torch.int8)
        fl = (torch.rand(n) < 0.5).to(torch.int8)
        padded = torch.empty(n, c, hw + 2 * cp, hw + 2 * cp)
        for ch in range(c):
            padded[:, ch] = pv[ch]
        padded[:, :, cp:cp + hw, cp:cp + hw] = x
        ref_img = torch.stack([padded[i, :, int(oy[i]):int(oy[i]) + hw, int(ox[i]):int(ox[i]) + hw] for i in range(n)])
        ref_img = torch.stack([im.flip(-1) if int(fl[i]) else im for i, im in enumerate(ref_img)])
    ho = (hw + 2 * pad - k) // stride + 1
    cols = F.unfold(ref_img, kernel_size=k, padding=pad, stride=stride).view(n, c, k * k, ho * ho).permute(0, 3, 2, 1).reshape(n, ho, ho, k * k * c)
    ref = torch.zeros(n, ho, ho, cin_pad)
    ref[..., : k * k * c] = cols
    out = torch.full((n, ho, ho, cin_pad), float("nan"), device="cuda").to(dtype)
    pvc = (lib.c_float * c)(*pv) if aug else None
    xd = x.cuda()                                   # (kept alive: the call only sees its address)
    lib.call("fb_stem_patches", xd.data_ptr(), out.data_ptr(), n, c, hw, hw, k, stride, pad, cin_pad,
             _dp(oy), _dp(ox), _dp(fl), cp if aug else 0, pvc, lib.dtype_code(dtype))
    got, want = out.float().cpu(), q(ref, dtype)
    bad = (got != want).nonzero()
    assert bad.shape[0] == 0, (bad.shape[0], bad[:6].tolist(), [float(got[tuple(b)]) for b in bad[:6]], [float(want[tuple(b)]) for b in bad[:6]],
                               None if oy is None else (oy.tolist(), ox.tolist(), fl.tolist()))


_KEEP = []


def _dp(t):
    if t is None:
        return None
    d = t.cuda()
    _KEEP.append(d)
    return d.data_ptr()


@pytest.mark.parametrize("clip", [None, 0.25, 1e3])
def test_sam_ascent_and_restore(clip):
    """fb_mt_sam_ascent / fb_mt_sam_restore vs the reference's first_step / second_step arithmetic (sam.py:56-77) after the
    closure's clip (training.py:198-206)."""
    lib = _lib()
    torch.manual_seed(5)
    n, rho = 100_003, 0.05
    theta, g = torch.randn(n), torch.randn(n) * 0.01
    td, gd, ed = theta.cuda(), g.cuda(), torch.zeros(n, device="cuda")
    norms2, ws = torch.zeros(2, device="cuda"), torch.zeros(lib.load().fb_ws_mt_floats(1), device="cuda")
    lib.call("fb_mt_norms2", gd.data_ptr(), None, n, norms2.data_ptr(), ws.data_ptr())
    lib.call("fb_mt_sam_ascent", td.data_ptr(), gd.data_ptr(), ed.data_ptr(), n, norms2.data_ptr(), -1.0 if clip is None else clip, rho)
    gc = g.clone()
    norm = gc.norm()
    if clip is not None and norm > clip:
        gc.mul_(clip / (norm + 1e-6))
    e_ref = gc * (rho / (gc.norm() + 1e-12))
    assert rel(ed.cpu(), e_ref) < 1e-6 and abs(float(ed.norm()) - rho) < 1e-6
    assert torch.allclose(td.cpu(), theta + e_ref, rtol=0, atol=1e-6)
    climbed = td.clone()
    lib.call("fb_mt_sam_restore", td.data_ptr(), ed.data_ptr(), n)
    assert torch.equal(td, climbed - ed)                       # the reference's p.sub_(e_w), same rounding
    assert torch.allclose(td.cpu(), theta, rtol=0, atol=1e-6)


def test_absmax_norm_bias_and_ema_kernels():
    """fb_mt_absmax2 (L-inf clip norm, training.py:199-200), fb_mt_norm_bias (training.py:188-196), fb_mt_ema (training/utils.py:22-29)."""
    lib = _lib()
    torch.manual_seed(9)
    n = 300_001
    g, theta, ema = torch.randn(n) * 0.01, torch.randn(n), torch.randn(n)
    g[123_457] = -0.75
    gd, td, ed = g.cuda(), theta.cuda(), ema.cuda()
    out, ws = torch.zeros(2, device="cuda"), torch.zeros(lib.load().fb_ws_mt_floats(1), device="cuda")
    lib.call("fb_mt_absmax2", gd.data_ptr(), n, out.data_ptr(), ws.data_ptr())
    assert float(out[0]) == 0.75 * 0.75
    for pnorm in (1.0, 3.0):
        lib.call("fb_mt_pnorm2", gd.data_ptr(), n, pnorm, out.data_ptr(), ws.data_ptr())
        assert abs(float(out[0].sqrt()) - float(torch.norm(g.double(), pnorm))) < 1e-5 * float(torch.norm(g.double(), pnorm))
    pn2 = torch.tensor([float(theta.pow(2).sum())], device="cuda")
    for norm_type, bias in ((1, 10.0), (1, 1e4), (2, 70.0)):
        gd = g.cuda()
        lib.call("fb_mt_norm_bias", gd.data_ptr(), td.data_ptr(), n, pn2.data_ptr(), 0.01, bias, norm_type)
        diff = pn2.cpu()[0] - bias ** 2
        ref = g + (0.01 * diff.sign() if norm_type == 1 else (0.01 * (2 * diff)) * theta)
        assert torch.allclose(gd.cpu(), ref, rtol=1e-6, atol=1e-7)
    lib.call("fb_mt_ema", ed.data_ptr(), td.data_ptr(), n, 0.99, float(1 - 0.99))
    assert torch.equal(ed.cpu(), 0.99 * ema + (1 - 0.99) * theta)            # the reference expression, bit for bit


def test_conv_masked_addend_equals_materialised_mask():
    """fb_conv_args.addend_mask (input gradients of the 3x3 layers of identity blocks): adding `addend` through the ReLU bitmask gives bit
    for bit what adding the materialised d * (out > 0) gives; unsupported shapes say so instead of ignoring the mask."""
    lib = _lib()
    torch.manual_seed(11)
    n = 16
    dy = (torch.randn(n, 32, 32, 64, device="cuda") * 0.1).bfloat16()
    w = (torch.randn(64, 9, 64, device="cuda") * 0.05).bfloat16()
    d = torch.randn(n, 32, 32, 64, device="cuda").bfloat16()
    out_act = torch.randn(n, 32, 32, 64, device="cuda")                       # stands in for the block output (sign = ReLU mask)
    bits = ((out_act.reshape(-1, 8) > 0).to(torch.int32) << torch.arange(8, device="cuda")).sum(1).to(torch.uint8)
    masked = torch.where(out_act > 0, d, torch.zeros_like(d))
    a, b = torch.empty_like(d), torch.empty_like(d)
    lib.conv2d(dy, w, a, 3, 3, 1, 1, 1, addend=masked, addend_mode=1)
    lib.conv2d(dy, w, b, 3, 3, 1, 1, 1, addend=d, addend_mode=1, addend_mask=bits)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert not torch.equal(a, d)                                               # (the convolution did something)
    # the persistent halo kernel (128 / 256 / 512 channels on 16x16 / 8x8 / 4x4 maps, round 5): the same identity
    for C, hw2 in ((128, 16), (256, 8), (512, 4)):
        dy2 = (torch.randn(n, hw2, hw2, C, device="cuda") * 0.1).bfloat16()
        w2 = (torch.randn(C, 9, C, device="cuda") * 0.03).bfloat16()
        d2 = torch.randn(n, hw2, hw2, C, device="cuda").bfloat16()
        act2 = torch.randn(n, hw2, hw2, C, device="cuda")
        bits2 = ((act2.reshape(-1, 8) > 0).to(torch.int32) << torch.arange(8, device="cuda")).sum(1).to(torch.uint8)
        a2, b2 = torch.empty_like(d2), torch.full_like(d2, float("nan"))
        lib.conv2d(dy2, w2, a2, 3, 3, 1, 1, 1, addend=torch.where(act2 > 0, d2, torch.zeros_like(d2)), addend_mode=1)
        lib.conv2d(dy2, w2, b2, 3, 3, 1, 1, 1, addend=d2, addend_mode=1, addend_mask=bits2)
        torch.cuda.synchronize()
        assert torch.equal(a2, b2), (C, hw2)
    # ... and with fp32 storage (the regulariser's passes): there fb_bn_apply's mask byte covers the FOUR channels of a 16-byte vector
    for C, hw2 in ((64, 32), (128, 16), (256, 8)):
        dy3 = torch.randn(n, hw2, hw2, C, device="cuda") * 0.1
        w3 = torch.randn(C, 9, C, device="cuda") * 0.03
        d3 = torch.randn(n, hw2, hw2, C, device="cuda")
        act3 = torch.randn(n, hw2, hw2, C, device="cuda")
        bits3 = ((act3.reshape(-1, 4) > 0).to(torch.int32) << torch.arange(4, device="cuda")).sum(1).to(torch.uint8)
        a3, b3 = torch.empty_like(d3), torch.full_like(d3, float("nan"))
        lib.conv2d(dy3, w3, a3, 3, 3, 1, 1, 1, addend=torch.where(act3 > 0, d3, torch.zeros_like(d3)), addend_mode=1)
        lib.conv2d(dy3, w3, b3, 3, 3, 1, 1, 1, addend=d3, addend_mode=1, addend_mask=bits3)
        torch.cuda.synchronize()
        assert torch.equal(a3, b3), (C, hw2)
    # a call no kernel takes the mask for says so instead of ignoring it: a pooled addend (addend_mode 2), or the implicit GEMM's mask path switched off for a 1x1 layer with K >= 512
    d4 = torch.randn(n, 8, 8, 256, device="cuda").bfloat16()
    bits4 = ((torch.randn(n, 8, 8, 256, device="cuda").reshape(-1, 8) > 0).to(torch.int32) << torch.arange(8, device="cuda")).sum(1).to(torch.uint8)
    dy4, w4 = torch.randn(n, 8, 8, 1024, device="cuda").bfloat16(), torch.randn(256, 1, 1024, device="cuda").bfloat16()
    os.environ["FB_IGEMM_NO_MASK"] = "1"
    try:
        with pytest.raises(lib.EngineError):
            lib.conv2d(dy4, w4, torch.empty_like(d4), 1, 1, 1, 0, 1, addend=d4, addend_mode=1, addend_mask=bits4)
    finally:
        del os.environ["FB_IGEMM_NO_MASK"]
    with pytest.raises(lib.EngineError):
        lib.conv2d(dy, w, torch.empty_like(d), 3, 3, 1, 1, 1, addend=d[:, ::2, ::2].contiguous(), addend_mode=2, addend_mask=bits)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,cd,r,hw,n", [(512, 2048, 1, 7, 9), (1024, 256, 1, 14, 3), (64, 256, 1, 56, 2), (256, 1024, 1, 14, 5), (128, 512, 1, 28, 3),
                                         (256, 256, 3, 14, 4), (64, 64, 3, 56, 1), (128, 128, 3, 28, 2), (512, 512, 3, 7, 6), (96, 64, 3, 6, 3)])
def test_implicit_gemm_masked_addend_equals_materialised_mask(dtype, k, cd, r, hw, n, monkeypatch):
    """fb_conv_args.addend_mask in the implicit GEMM's epilogue (round 6): the input gradients of identity blocks that no specialised kernel serves -- every
    Bottleneck identity block with fp32 storage (the regulariser's passes of BASELINE config 5: 1x1 convolutions with K = 64 ... 512 on 56 / 28 / 14 / 7 maps), the
    512-channel ones in bf16, 3x3 layers on ImageNet-shaped maps.  Taking `d` through the ReLU bitmask of the block output (one byte per 16-byte vector: 4 fp32 or 8
    bf16 channels) gives bit for bit what adding the materialised d * (out > 0) gives."""
    lib = _lib()
    for key in ("FB_C1G", "FB_C1S_PIPE"):
        monkeypatch.setenv(key, "0")                     # (bf16: the streaming kernels take K <= 256 themselves; here the implicit GEMM is under test for every shape)
    torch.manual_seed(k + hw + r)
    pad = r // 2
    dy = (torch.randn(n, hw, hw, k, device="cuda") * 0.1).to(dtype)
    wt = (torch.randn(cd, r * r, k, device="cuda") * 0.05).to(dtype)
    d = torch.randn(n, hw, hw, cd, device="cuda").to(dtype)
    out_act = torch.randn(n, hw, hw, cd, device="cuda")
    vec = 16 // d.element_size()
    bits = ((out_act.reshape(-1, vec) > 0).to(torch.int32) << torch.arange(vec, device="cuda")).sum(1).to(torch.uint8)
    masked = torch.where(out_act > 0, d, torch.zeros_like(d))
    a, b, plain = torch.empty_like(d), torch.full_like(d, float("nan")), torch.empty_like(d)
    args = lib.ConvArgs(dy.data_ptr(), wt.data_ptr(), b.data_ptr(), d.data_ptr(), None, n, hw, hw, k, hw, hw, cd, r, r, 1, pad, 1, 0, 0, 1, lib.dtype_code(dtype),
                        bits.data_ptr(), None, None, None, None, 0)
    assert lib.load().fb_conv_masked_addend_supported(lib.C.byref(args))
    lib.conv2d(dy, wt, a, r, r, 1, pad, 1, addend=masked, addend_mode=1)
    lib.conv2d(dy, wt, b, r, r, 1, pad, 1, addend=d, addend_mode=1, addend_mask=bits)
    lib.conv2d(dy, wt, plain, r, r, 1, pad, 1)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and not torch.equal(a, plain)
    assert rel((a.float() - plain.float()).cpu(), masked.float().cpu()) < (1e-5 if dtype == torch.float32 else 5e-2)     # the difference IS the masked addend
    monkeypatch.setenv("FB_IGEMM_NO_MASK", "1")
    assert not lib.load().fb_conv_masked_addend_supported(lib.C.byref(args))


@pytest.mark.parametrize("k,cd,hw,n", [(256, 1024, 14, 64), (256, 1024, 14, 3), (128, 512, 28, 5), (64, 256, 56, 2), (256, 512, 7, 9), (256, 128, 8, 4)])
def test_conv1x1_masked_addend_equals_materialised_mask(k, cd, hw, n, monkeypatch):
    """fb_conv_args.addend_mask in the streaming 1x1 input gradients (round 5: the identity Bottleneck blocks, reference resnets.py:312-316 -- the gradient
    entering the residual branch is d * (out > 0)): taking `d` through the ReLU bitmask of the block output gives bit for bit what adding the materialised
    d * (out > 0) gives, at every K the pipelined kernel serves, with ragged last groups (pixel counts that are no multiple of 64) and 4-wave
    workgroups (128 output channels); with the kernel switched off the layer says "not supported" and the engine materialises the mask."""
    lib = _lib()
    torch.manual_seed(k + hw)
    dy = (torch.randn(n, hw, hw, k, device="cuda") * 0.1).bfloat16()
    wt = (torch.randn(cd, 1, k, device="cuda") * 0.05).bfloat16()
    d = torch.randn(n, hw, hw, cd, device="cuda").bfloat16()
    out_act = torch.randn(n, hw, hw, cd, device="cuda")
    bits = ((out_act.reshape(-1, 8) > 0).to(torch.int32) << torch.arange(8, device="cuda")).sum(1).to(torch.uint8)
    masked = torch.where(out_act > 0, d, torch.zeros_like(d))
    a, b = torch.empty_like(d), torch.full_like(d, float("nan"))
    args = lib.ConvArgs(dy.data_ptr(), wt.data_ptr(), b.data_ptr(), d.data_ptr(), None, n, hw, hw, k, hw, hw, cd, 1, 1, 1, 0, 1, 0, 0, 1, lib.dtype_code(torch.bfloat16),
                        bits.data_ptr(), None, None, None, None, 0)
    assert lib.load().fb_conv_masked_addend_supported(lib.C.byref(args))
    lib.conv2d(dy, wt, a, 1, 1, 1, 0, 1, addend=masked, addend_mode=1)
    lib.conv2d(dy, wt, b, 1, 1, 1, 0, 1, addend=d, addend_mode=1, addend_mask=bits)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    plain = torch.empty_like(d)
    lib.conv2d(dy, wt, plain, 1, 1, 1, 0, 1)
    assert not torch.equal(a, plain)
    # the 256 x 256 GEMM kernel (large calls of the expanding layers by default; FB_C1G=2: every call it can take) adds the same values in its epilogue: same bits,
    # with the mask, with a plain addend, without
    monkeypatch.setenv("FB_C1G", "2")
    c, e, f = torch.full_like(d, float("nan")), torch.full_like(d, float("nan")), torch.full_like(d, float("nan"))
    lib.conv2d(dy, wt, c, 1, 1, 1, 0, 1, addend=d, addend_mode=1, addend_mask=bits)
    lib.conv2d(dy, wt, e, 1, 1, 1, 0, 1, addend=masked, addend_mode=1)
    lib.conv2d(dy, wt, f, 1, 1, 1, 0, 1)
    torch.cuda.synchronize()
    assert torch.equal(a, c) and torch.equal(a, e) and torch.equal(plain, f)
    # with both switched off the call falls to the implicit GEMM, which applies the mask in its epilogue (round 6): same bits again; with that off too it says so
    monkeypatch.setenv("FB_C1G", "0")
    monkeypatch.setenv("FB_C1S_PIPE", "0")
    g = torch.full_like(d, float("nan"))
    lib.conv2d(dy, wt, g, 1, 1, 1, 0, 1, addend=d, addend_mode=1, addend_mask=bits)
    torch.cuda.synchronize()
    assert torch.equal(a, g)
    monkeypatch.setenv("FB_IGEMM_NO_MASK", "1")
    assert not lib.load().fb_conv_masked_addend_supported(lib.C.byref(args))
    with pytest.raises(lib.EngineError):
        lib.conv2d(dy, wt, b, 1, 1, 1, 0, 1, addend=d, addend_mode=1, addend_mask=bits)


@pytest.mark.parametrize("C,hw,amode", [(64, 32, 0), (64, 32, 1), (64, 32, 3), (128, 16, 0), (128, 16, 1), (256, 8, 0), (256, 8, 1), (512, 4, 0),
                                         (512, 4, 1)])
def test_conv_dgrad_fused_bn_backward_reduction(C, hw, amode):
    """The BatchNorm-backward reduction fused into the input-gradient epilogues (fb_conv_args.bst_x / bst_mask): the gradient
    itself is bit-identical to the plain launch, and dgamma / dbeta / the fb_bn_bwd_apply coefficients obtained from the fused 128-pixel
    sums of (g, g*x) equal those of the separate fb_bn_bwd_reduce pass over the stored gradient (amode 1: + addend, 3: + masked addend)."""
    lib = _lib()
    torch.manual_seed(C + hw + amode)
    ipg, groups = 128, 3
    n = ipg * groups
    dt = torch.bfloat16
    dy = torch.randn(n, hw, hw, C, device="cuda").to(dt)
    wt = (torch.randn(C, 9, C, device="cuda") * 0.05).to(dt)
    x = (torch.randn(n, hw, hw, C, device="cuda") * 1.5 + 0.7).to(dt)                      # input of the consuming BatchNorm
    bits = torch.randint(0, 256, (n * hw * hw * C // 8,), device="cuda", dtype=torch.uint8)  # its ReLU bitmask
    addend = torch.randn(n, hw, hw, C, device="cuda").to(dt) if amode else None
    amask = torch.randint(0, 256, (n * hw * hw * C // 8,), device="cuda", dtype=torch.uint8) if amode == 3 else None
    px, ppg = n * hw * hw, ipg * hw * hw
    plain, fused = torch.empty_like(dy), torch.empty_like(dy)
    part_f = torch.zeros(2, px // 128, C, device="cuda")
    lib.conv2d(dy, wt, plain, 3, 3, 1, 1, 1, addend=addend, addend_mode=1 if amode else 0, addend_mask=amask)
    lib.conv2d(dy, wt, fused, 3, 3, 1, 1, 1, addend=addend, addend_mode=1 if amode else 0, addend_mask=amask, stat_partial=part_f, bst_x=x, bst_mask=bits)
    assert torch.equal(plain, fused)
    mean_tab = x.float().reshape(groups, -1, C).mean(1).contiguous()
    invstd = (1.0 / (x.float().reshape(groups, -1, C).var(1, unbiased=False) + 1e-5).sqrt()).contiguous()
    scale = torch.rand(groups, C, device="cuda") + 0.5
    rows = lib.load().fb_bn_bwd_reduce_rows(px, ppg)
    part_s = torch.zeros(2, rows, C, device="cuda")
    lib.call("fb_bn_bwd_reduce", plain.data_ptr(), None, bits.data_ptr(), x.data_ptr(), mean_tab.data_ptr(), invstd.data_ptr(), C, 0, part_s.data_ptr(), px, C,
             ppg, lib.dtype_code(dt))
    outs = []
    for part, nrows, raw in ((part_s, rows, 0), (part_f, px // 128, 1)):
        gout = torch.zeros(groups, 2 * C, device="cuda")
        coef = torch.zeros(groups, C, 3, device="cuda")
        lib.call("fb_bn_bwd_finalize", part.data_ptr(), nrows, groups, C, float(ppg), scale.data_ptr(), mean_tab.data_ptr(), invstd.data_ptr(), C, 0,
                 gout.data_ptr(), gout.data_ptr() + 4 * C, 2 * C, coef.data_ptr(), raw)
        outs.append((gout.double().cpu(), coef.double().cpu()))
    # float64 reference of the sums from the stored gradient
    g = plain.double() * ((bits.view(-1, 1).to(torch.int32) >> torch.arange(8, device="cuda")) & 1).reshape(plain.shape).double()
    xhat = (x.double().reshape(groups, -1, C) - mean_tab.double()[:, None]) * invstd.double()[:, None]
    dbeta, dgamma = g.reshape(groups, -1, C).sum(1).cpu(), (g.reshape(groups, -1, C) * xhat).sum(1).cpu()
    for gout, coef in outs:
        assert rel(gout[:, :C], dgamma) < 2e-5 and rel(gout[:, C:], dbeta) < 2e-5
    assert rel(outs[1][1], outs[0][1]) < 2e-5


def test_fused_reduction_and_amax_fallbacks_are_loud_or_exact():
    """The optional fused paths say so when a shape is not theirs: fb_conv_bwd_stat_supported is 0 for fp32 / stride 2 / 1x1 calls and fb_conv2d
    then refuses bst_x instead of ignoring it; fb_bn_apply's amax_out takes the separate fb_absmax pass when the channel count does not fit
    the span kernel (C = 96) and still returns the exact per-group maxima; amax_out on bf16 tensors is refused."""
    lib = _lib()
    torch.manual_seed(5)
    n, hw, C = 4, 8, 128
    dy = torch.randn(n, hw, hw, C, device="cuda")
    wt = torch.randn(C, 9, C, device="cuda") * 0.05
    x = torch.randn(n, hw, hw, C, device="cuda")
    bits = torch.randint(0, 256, (n * hw * hw * C // 8,), device="cuda", dtype=torch.uint8)
    part = torch.zeros(2, n * hw * hw // 128, C, device="cuda")
    with pytest.raises(lib.EngineError):                                        # fp32: not implemented, and loudly so
        lib.conv2d(dy, wt, torch.empty_like(dy), 3, 3, 1, 1, 1, stat_partial=part, bst_x=x, bst_mask=bits)
    a = lib.ConvArgs(dy.data_ptr(), wt.data_ptr(), dy.data_ptr(), None, part.data_ptr(), n, hw, hw, C, hw, hw, C, 1, 1, 1, 0, 1, 0, 0, 0,
                     lib.dtype_code(torch.bfloat16), None, x.data_ptr(), bits.data_ptr(), None, None, 0)
    assert lib.load().fb_conv_bwd_stat_supported(lib.C.byref(a)) == 0          # 1x1: no fused reduction
    # amax_out through the grid-stride form of fb_bn_apply (C = 96: 24 vectors per pixel do not divide 256)
    C2, groups, ipg = 96, 2, 4
    px, ppg = groups * ipg * hw * hw, ipg * hw * hw
    x2 = torch.randn(px, C2, device="cuda") * 3
    y2 = torch.empty_like(x2)
    scale, shift = torch.rand(groups, C2, device="cuda") + 0.5, torch.randn(groups, C2, device="cuda")
    amax = torch.zeros(groups, device="cuda")
    ws = torch.zeros(int(lib.load().fb_ws_bn_amax_floats(px, C2, ppg)) + 1, device="cuda")
    lib.call("fb_bn_apply", x2.data_ptr(), y2.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, None, px, C2, ppg, 0, 1, None, None, 0,
             lib.dtype_code(torch.float32), amax.data_ptr(), ws.data_ptr())
    ref = torch.relu(x2.view(groups, -1, C2) * scale[:, None] + shift[:, None])
    assert torch.allclose(y2.view(groups, -1, C2), ref, rtol=1e-6, atol=1e-6) and torch.equal(amax, y2.reshape(groups, -1).abs().max(1).values)
    with pytest.raises(lib.EngineError):
        xb = x2[:, :64].contiguous().bfloat16()
        lib.call("fb_bn_apply", xb.data_ptr(), torch.empty_like(xb).data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, None, px, 64, ppg, 0, 1,
                 None, None, 0, lib.dtype_code(torch.bfloat16), amax.data_ptr(), ws.data_ptr())


def test_engine_per_tensor_weight_decay_matches_torch_sgd():
    """Engine.sgd_step_per_tensor (hyp.only_linear_layers_weight_decay) vs torch.optim.SGD with one param group per tensor, two steps
    (first-step momentum initialisation included)."""
    from fullbatchtraining_amd.cfg import compose
    from fullbatchtraining_amd.engine import Engine
    from fullbatchtraining_amd.models import construct_model
    from fullbatchtraining_amd.training import optim_interface

    torch.manual_seed(3)
    cfg = compose(["hyp=fb1", "hyp.only_linear_layers_weight_decay=True", "hyp.optim.weight_decay=0.05"])
    model = construct_model(cfg.model, 3, 10)
    eng = Engine(model, 16, 32, 1, compute_dtype=torch.float32)
    opt, _ = optim_interface(model, cfg.hyp)
    wds = [g["weight_decay"] for g in opt.param_groups]
    o = cfg.hyp.optim
    for step in range(2):
        grads = [torch.randn_like(p) * 0.01 for p in model.parameters()]
        flat = torch.zeros(eng.plan.P)
        for name, g in zip(eng.plan.param_names, grads):
            v = g.permute(0, 2, 3, 1).reshape(-1) if g.dim() == 4 else g.reshape(-1)
            flat[eng.plan.offsets[name]:eng.plan.offsets[name] + v.numel()] = v
        eng.avg.copy_(flat)
        eng.sgd_step_per_tensor(0.1, wds, o.momentum, o.dampening, o.nesterov, None)
        for p, g in zip(model.parameters(), grads):
            p.grad = g
        for g in opt.param_groups:
            g["lr"] = 0.1
        opt.step()
    got = eng.theta.cpu()
    for name, p in zip(eng.plan.param_names, model.parameters()):
        assert torch.allclose(eng._unflatten(got, name), p.detach(), rtol=1e-6, atol=1e-7), name


@pytest.mark.parametrize("smoothing,only_incorrect", [(0.1, 0), (0.0, 1), (0.05, 1)])
def test_head_loss_variants(smoothing, only_incorrect):
    """fb_head_loss with the loss functions of get_loss_fn (reference training.py:391-413, modules.py:86-119) vs autograd on the
    reference formulas."""
    lib = _lib()
    torch.manual_seed(13)
    groups, ipg, C, classes = 3, 32, 64, 10
    n = groups * ipg
    feat = torch.randn(n, C)
    W, b = torch.randn(classes, C) * 0.3, torch.randn(classes) * 0.1
    labels = torch.randint(0, classes, (n,))
    theta = torch.cat([W.reshape(-1), b])
    thd, lab, fd = theta.cuda(), labels.cuda(), feat.cuda()
    logits, dlogits = torch.empty(n, classes, device="cuda"), torch.empty(n, classes, device="cuda")
    loss, correct = torch.empty(groups, device="cuda"), torch.empty(groups, device="cuda")
    lib.call("fb_head_loss", fd.data_ptr(), thd.data_ptr(), thd.data_ptr() + 4 * classes * C, 0, lab.data_ptr(), logits.data_ptr(),
             dlogits.data_ptr(), loss.data_ptr(), correct.data_ptr(), groups, ipg, C, classes, smoothing, only_incorrect)
    for g in range(groups):
        z = (feat[g * ipg:(g + 1) * ipg].double() @ W.double().t() + b.double()).requires_grad_(True)
        y = labels[g * ipg:(g + 1) * ipg]
        log_prob = torch.nn.functional.log_softmax(z, dim=-1)
        weight = torch.ones_like(z) * smoothing / (classes - 1.0)
        weight.scatter_(-1, y.unsqueeze(-1), 1.0 - smoothing)
        per_sample = (-weight * log_prob).sum(dim=-1)
        if only_incorrect:
            per_sample = per_sample * (1 - (z.argmax(dim=1) == y).double())
        ref = per_sample.mean()
        ref.backward()
        assert abs(float(loss[g]) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
        assert rel(dlogits[g * ipg:(g + 1) * ipg].cpu(), z.grad) < 1e-5
        assert float(correct[g]) == float((z.argmax(dim=1) == y).sum())


def test_clip_scale_and_grad_noise_kernels():
    """fb_mt_clip_scale / fb_mt_grad_noise vs the reference's in-place tensor expressions (training.py:205-215), bit for bit."""
    lib = _lib()
    torch.manual_seed(17)
    n = 200_003
    g, noise = torch.randn(n) * 0.01, torch.randn(n)
    gd, nd = g.cuda(), noise.cuda()
    norms2, ws = torch.zeros(2, device="cuda"), torch.zeros(lib.load().fb_ws_mt_floats(1), device="cuda")
    lib.call("fb_mt_norms2", gd.data_ptr(), None, n, norms2.data_ptr(), ws.data_ptr())
    lib.call("fb_mt_clip_scale", gd.data_ptr(), n, norms2.data_ptr(), 0.25)
    norm = norms2[0].sqrt().cpu()
    ref = g * (0.25 / (norm + 1e-6)) if norm > 0.25 else g.clone()
    assert torch.equal(gd.cpu(), ref)
    lib.call("fb_mt_clip_scale", gd.data_ptr(), n, norms2.data_ptr(), 1e6)       # no clip: untouched
    assert torch.equal(gd.cpu(), ref)
    lib.call("fb_mt_grad_noise", gd.data_ptr(), nd.data_ptr(), n, 0.01, 0)
    ref.add_(0.01 * noise)
    assert torch.equal(gd.cpu(), ref)
    lib.call("fb_mt_grad_noise", gd.data_ptr(), nd.data_ptr(), n, 0.1, 1)
    ref.mul_(1 + 0.1 * noise)
    assert torch.equal(gd.cpu(), ref)


@pytest.mark.parametrize("C,W", [(64, 32), (128, 16), (256, 8)])
def test_bn_apply_fused_avgpool_is_bit_identical(C, W):
    """fb_bn_apply(pool_out=...): the 2x2 average pooling of the block output for the next block's 'C' shortcut (reference resnets.py:149),
    written by the BN pass itself, equals fb_avgpool2_fwd applied to the stored activation bit for bit (bf16, with residual + ReLU)."""
    from fullbatchtraining_amd import lib

    G, ipg = 3, 4
    n = G * ipg
    px, ppg = n * W * W, ipg * W * W
    dt = lib.dtype_code(torch.bfloat16)
    assert lib.load().fb_bn_apply_can_pool(C, W, ppg, dt) == 1 and lib.load().fb_bn_apply_can_pool(C, W, ppg, lib.dtype_code(torch.float32)) == 0
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, W, W, C, generator=gen).bfloat16().cuda()
    res = torch.randn(n, W, W, C, generator=gen).bfloat16().cuda()
    scale, shift = (torch.rand(G, C, generator=gen) + 0.5).cuda(), torch.randn(G, C, generator=gen).cuda()
    y, y2 = torch.empty_like(x), torch.empty_like(x)
    pooled = torch.full((n, W // 2, W // 2, C), 7.0, dtype=torch.bfloat16, device="cuda")
    want = torch.empty_like(pooled)
    bits = torch.zeros(x.numel() // 8, dtype=torch.uint8, device="cuda")
    lib.call("fb_bn_apply", x.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), res.data_ptr(), None, None, px, C, ppg, 0, 1,
             bits.data_ptr(), pooled.data_ptr(), W, dt, None, None)
    lib.call("fb_bn_apply", x.data_ptr(), y2.data_ptr(), scale.data_ptr(), shift.data_ptr(), res.data_ptr(), None, None, px, C, ppg, 0, 1,
             bits.data_ptr(), None, 0, dt, None, None)
    lib.call("fb_avgpool2_fwd", y2.data_ptr(), want.data_ptr(), n, W, W, C, dt)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    assert torch.equal(pooled.view(torch.int16), want.view(torch.int16))
    ref = torch.relu(x.float() * scale.repeat_interleave(ipg, 0)[:, None, None, :] + shift.repeat_interleave(ipg, 0)[:, None, None, :] + res.float())
    ref = ref.bfloat16().float().view(n, W // 2, 2, W // 2, 2, C).mean(dim=(2, 4))
    assert float((pooled.float() - ref).abs().max()) < 2e-2 * float(ref.abs().max())


# ---------------------------------------------------------------------------------------------------------------------------------
# Entry points whose only check used to be an end-to-end training scenario (round 3 verdict): each against a plain torch restatement
# of the reference lines it implements.
def test_chunk_clip_and_fd_combine_vs_torch():
    """fb_mt_fd_combine: g[j] += cf * (ga[j] - gb[j]) / eps_n[j] (reference modules.py:232-240 without the average);
    fb_mt_chunk_clip: _clip_gradient_list with p = 2 (reference training/utils.py:4-19): norm > clip -> g *= clip / (norm + 1e-6)."""
    lib = _lib()
    torch.manual_seed(21)
    P, G = 50_001, 9                 # odd length: the vector tail; padded group stride
    stride = (P + 63) // 64 * 64
    g, ga, gb = (torch.randn(G, stride) for _ in range(3))
    for t in (g, ga, gb):
        t[:, P:] = 0
    eps_n = torch.rand(G) * 1e-2 + 1e-3
    gd, gad, gbd, ed = g.cuda(), ga.cuda(), gb.cuda(), eps_n.cuda()
    lib.call("fb_mt_fd_combine", gd.data_ptr(), gad.data_ptr(), gbd.data_ptr(), stride, G, P, ed.data_ptr(), 0.025)
    ref = g[:, :P] + 0.025 * ((ga[:, :P] - gb[:, :P]) / eps_n[:, None])
    assert torch.allclose(gd[:, :P].cpu(), ref, rtol=1e-6, atol=1e-6)
    assert float(gd[:, P:].abs().max()) == 0                       # the alignment padding of the arena rows is not touched
    # per-chunk clip: scale rows so that about half of them exceed the clip; one row sits exactly AT the clip norm (not clipped: strict >)
    rows = ref.clone()
    norms = rows.double().norm(dim=1)
    srt = norms.sort().values
    clip = float(0.5 * (srt[G // 2 - 1] + srt[G // 2]))           # between two rows' norms: the decision does not hang on the last bit of a square root
    gd2 = torch.zeros(G, stride)
    gd2[:, :P] = rows
    gd2 = gd2.cuda()
    sq = torch.zeros(G, device="cuda")
    ws = torch.zeros(max(G, 2) * lib.MT_BLOCKS, device="cuda")
    lib.call("fb_mt_sqnorm", gd2.data_ptr(), stride, G, P, 1.0, None, 0.0, sq.data_ptr(), ws.data_ptr())
    clipped = torch.full((G,), -1.0, device="cuda")
    lib.call("fb_mt_chunk_clip", gd2.data_ptr(), stride, G, P, sq.data_ptr(), clip, clipped.data_ptr())
    norm32 = sq.cpu().sqrt()
    want_flag = (norm32 > clip).float()
    assert torch.equal(clipped.cpu(), want_flag) and 0 < int(want_flag.sum()) < G
    coef = torch.where(norm32 > clip, clip / (norm32 + 1e-6), torch.ones(G))
    assert torch.allclose(gd2[:, :P].cpu(), rows * coef[:, None], rtol=1e-6, atol=1e-8)
    # against torch's own clip on the same rows (the reference's call): clipped rows end up at the clip norm
    after = gd2[:, :P].cpu().double().norm(dim=1)
    assert torch.allclose(after[want_flag.bool()], torch.full((int(want_flag.sum()),), clip, dtype=torch.float64), rtol=1e-5)
    # `clipped` is optional (the pre-pass passes NULL)
    lib.call("fb_mt_chunk_clip", gd2.data_ptr(), stride, G, P, sq.data_ptr(), 1e9, None)


@pytest.mark.parametrize("C", [64, 2048, 100])
def test_bn_eval_coeffs_vs_torch_batchnorm_eval(C):
    """fb_bn_eval_coeffs: y = x * scale + shift must be torch's BatchNorm2d in eval mode (reference training.py:343-347 model.eval())."""
    lib = _lib()
    torch.manual_seed(C)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C)), bn.bias.copy_(torch.randn(C))
        bn.running_mean.copy_(torch.randn(C)), bn.running_var.copy_(torch.rand(C) * 3 + 1e-3)
    bn.eval()
    scale, shift = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dev = [t.detach().cuda() for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]         # (kept alive across the call)
    lib.call("fb_bn_eval_coeffs", dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), bn.eps, scale.data_ptr(), shift.data_ptr(), C)
    torch.cuda.synchronize()
    x = torch.randn(5, C, 3, 3)
    got = x * scale.cpu()[None, :, None, None] + shift.cpu()[None, :, None, None]
    with torch.no_grad():
        want = bn(x)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    ref_scale = bn.weight.detach().double() / (bn.running_var.double() + bn.eps).sqrt()
    assert torch.allclose(scale.cpu().double(), ref_scale, rtol=2e-6)
    assert torch.allclose(shift.cpu().double(), bn.bias.detach().double() - bn.running_mean.double() * ref_scale, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n,classes", [(257, 10), (64, 1000), (1, 10)])
def test_head_tta_vs_reference_formula(n, classes):
    """fb_head_tta: the reference's test-time-flip epilogue (training.py:370-373): outputs = softmax(z_a) + softmax(z_b); the loss function is
    applied to those SUMMED PROBABILITIES as if they were logits (CrossEntropyLoss(outputs, labels)), predictions = argmax(outputs)."""
    lib = _lib()
    torch.manual_seed(n + classes)
    za, zb = torch.randn(n, classes) * 3, torch.randn(n, classes) * 3
    y = torch.randint(0, classes, (n,))
    zb[: n // 2] = za[: n // 2] + 0.1 * torch.randn(n // 2, classes)        # (the mirror of an image mostly agrees with it)
    ws = torch.zeros(2 * n + 2, device="cuda")
    zad, zbd, yd = za.cuda(), zb.cuda(), y.cuda()
    lib.call("fb_head_tta", zad.data_ptr(), zbd.data_ptr(), yd.data_ptr(), n, classes, ws.data_ptr(), ws.data_ptr() + 8 * n, ws.data_ptr() + 8 * n + 4)
    outputs = torch.softmax(za.double(), 1) + torch.softmax(zb.double(), 1)
    loss_sum = torch.nn.functional.cross_entropy(outputs, y, reduction="sum")
    correct = (outputs.argmax(1) == y).sum()
    got_loss, got_correct = ws[2 * n:].tolist()
    assert abs(got_loss - float(loss_sum)) < 1e-5 * max(1.0, float(loss_sum)), (got_loss, float(loss_sum))
    assert got_correct == float(correct)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,hw,imgs,G", [(64, 16, 32, 5), (128, 8, 64, 3), (512, 2, 128, 4), (64, 32, 128, 3)])
def test_bn_bwd_fused_equals_the_two_pass_form(dtype, C, hw, imgs, G):
    """fb_bn_bwd_fused (one pass over (dout, x): a resident cluster of workgroups keeps a statistics group's operands in registers between the
    reduction and the apply step; off by default -- it measured slower, profiles/r4_notes.md) against fb_bn_bwd_reduce -> finalize -> apply on
    the same inputs: same masked gradient bit for bit, dx / dgamma / dbeta to fp32 rounding (the pixel partition of the sums differs), and
    against torch's native_batch_norm_backward through autograd."""
    lib = _lib()
    handle = lib.load()
    dtc = lib.dtype_code(dtype)
    n = G * imgs
    px, ppg = n * hw * hw, imgs * hw * hw
    if not handle.fb_bn_bwd_fused_supported(px, C, ppg, dtc):
        pytest.skip("shape not for the cluster kernel")
    torch.manual_seed(C + hw)
    eb = 4 if dtype == torch.float32 else 2
    x = torch.randn(n, hw, hw, C, device="cuda").to(dtype)
    dout = torch.randn(n, hw, hw, C, device="cuda").to(dtype)
    gamma = torch.rand(C, device="cuda") + 0.5
    xg = x.float().view(G, imgs * hw * hw, C)
    mean, var = xg.mean(1), xg.var(1, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    scale = gamma[None] * invstd
    y = (xg - mean[:, None]) * scale[:, None] + 0.1                                  # the forward output whose ReLU mask the backward uses
    keep = (y > 0)
    bits = (keep.reshape(-1, 8).to(torch.int32) * (2 ** torch.arange(8, device="cuda", dtype=torch.int32))).sum(1).to(torch.uint8) if eb == 2 else \
        (keep.reshape(-1, 4).to(torch.int32) * (2 ** torch.arange(4, device="cuda", dtype=torch.int32))).sum(1).to(torch.uint8)
    out = {}
    for mode in ("two-pass", "fused"):
        dx, dy = torch.empty_like(x), torch.empty_like(x)
        gout = torch.zeros(G, 2 * C, device="cuda")
        coef = torch.zeros(G, C, 3, device="cuda")
        if mode == "two-pass":
            rows = handle.fb_bn_bwd_reduce_rows(px, ppg)
            ws = torch.zeros(int(handle.fb_ws_bn_partial_floats(px, C)), device="cuda")
            lib.call("fb_bn_bwd_reduce", dout.data_ptr(), None, bits.data_ptr(), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), C, 0, ws.data_ptr(), px, C, ppg, dtc)
            lib.call("fb_bn_bwd_finalize", ws.data_ptr(), rows, G, C, float(ppg), scale.data_ptr(), mean.data_ptr(), invstd.data_ptr(), C, 0,
                     gout.data_ptr(), gout.data_ptr() + 4 * C, 2 * C, coef.data_ptr(), 0)
            lib.call("fb_bn_bwd_apply", dout.data_ptr(), None, bits.data_ptr(), x.data_ptr(), coef.data_ptr(), dx.data_ptr(), dy.data_ptr(), px, C, ppg, dtc, None, None)
        else:
            ws = torch.zeros(int(handle.fb_ws_bn_bwd_fused_floats(px, C, ppg, dtc)), device="cuda")
            sync = torch.zeros(int(handle.fb_ws_bn_bwd_fused_ints(G)), device="cuda", dtype=torch.int32)
            for _ in range(2):                                                       # (a second launch on the same workspace: counters and flags are re-armed)
                lib.call("fb_bn_bwd_fused", dout.data_ptr(), bits.data_ptr(), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), C, 0,
                         gout.data_ptr(), gout.data_ptr() + 4 * C, 2 * C, coef.data_ptr(), dx.data_ptr(), dy.data_ptr(), px, C, ppg, float(ppg), dtc,
                         ws.data_ptr(), sync.data_ptr())
            torch.cuda.synchronize()
            assert int(sync[-1]) == 0, "a cluster timed out"
        torch.cuda.synchronize()
        out[mode] = (dx.float(), dy.float(), gout.clone())
    a, b = out["fused"], out["two-pass"]
    assert torch.equal(a[1], b[1])
    tol = 1e-5 if dtype == torch.float32 else 6e-3                                   # bf16: one output rounding where a coefficient differs in its last bit
    assert rel(a[0], b[0]) < tol and rel(a[2], b[2]) < 1e-5, (rel(a[0], b[0]), rel(a[2], b[2]))
    # torch: d/dx and d/dgamma, d/dbeta of  relu-masked BatchNorm  with the same upstream gradient
    xt = x.float().view(G, -1, C).clone().requires_grad_(True)
    gm, bt = gamma.clone().requires_grad_(True), torch.full((C,), 0.1, device="cuda", requires_grad=True)
    for g in range(G):
        yt = torch.nn.functional.batch_norm(xt[g].t().reshape(1, C, -1), None, None, gm, bt, True, 0.0, 1e-5)
        (yt * (dout.float().view(G, -1, C)[g] * keep[g]).t().reshape(1, C, -1)).sum().backward()
    want = xt.grad.view_as(a[0])
    assert rel(a[0], want) < (1e-4 if dtype == torch.float32 else 8e-3), rel(a[0], want)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cin_pad,hw,imgs,G,split", [(32, 32, 16, 3, 4), (160, 16, 8, 2, 2), (32, 8, 128, 2, 1)])
def test_wgrad_with_the_batchnorm_apply_in_its_loader(dt, cin_pad, hw, imgs, G, split):
    """fb_wgrad_args.bn_x: the stem's weight gradient computes dx = c_dy * (dout masked) + c_x * x + c_0 (fb_bn_bwd_apply) in its operand loader
    instead of reading a materialised dx: the same slabs as fb_bn_bwd_apply -> fb_conv2d_wgrad, bit for bit (same fp32 expression, same bf16
    rounding; fp32 storage: the bf16x6 split of the computed values), with and without a ReLU bitmask."""
    lib = _lib()
    handle = lib.load()
    dtc, C = lib.dtype_code(dt), 64
    n = G * imgs
    px, ppg = n * hw * hw, imgs * hw * hw
    torch.manual_seed(cin_pad + hw)
    patches = torch.randn(n, hw, hw, cin_pad, device="cuda").to(dt)
    x = torch.randn(n, hw, hw, C, device="cuda").to(dt)
    dout = torch.randn(n, hw, hw, C, device="cuda").to(dt)
    bits = torch.randint(0, 256, (x.numel() * x.element_size() // 16,), device="cuda", dtype=torch.uint8)
    coef = torch.randn(G, C, 3, device="cuda")
    for mask in (bits, None):
        dx = torch.empty_like(x)
        lib.call("fb_bn_bwd_apply", dout.data_ptr(), None, mask.data_ptr() if mask is not None else None, x.data_ptr(), coef.data_ptr(), dx.data_ptr(), None, px, C, ppg,
                 dtc, None, None)
        slabs = []
        for fused in (False, True):
            slab = torch.zeros(G * split * C * cin_pad, device="cuda")
            a = lib.WgradArgs(patches.data_ptr(), (dout if fused else dx).data_ptr(), slab.data_ptr(), n, hw, hw, cin_pad, hw, hw, C, 1, 1, 1, 0, imgs, split, dtc, 0,
                              None, None, x.data_ptr() if fused else None, (mask.data_ptr() if mask is not None else None) if fused else None,
                              coef.data_ptr() if fused else None)
            if fused:
                assert handle.fb_wgrad_bn_fused_supported(lib.C.byref(a))
            lib.call("fb_conv2d_wgrad", lib.C.byref(a))
            torch.cuda.synchronize()
            slabs.append(slab)
        assert bool(torch.isfinite(slabs[1]).all()) and float(slabs[0].abs().max()) > 0
        assert torch.equal(slabs[0], slabs[1]), float((slabs[0] - slabs[1]).abs().max() / slabs[0].abs().max())
    # a layer the loader form is not for is refused, not silently computed otherwise
    a = lib.WgradArgs(x.data_ptr(), dout.data_ptr(), slabs[0].data_ptr(), n, hw, hw, 64, hw, hw, C, 1, 1, 1, 0, imgs, 1, dtc, 0, None, None, x.data_ptr(), None, coef.data_ptr())
    assert not handle.fb_wgrad_bn_fused_supported(lib.C.byref(a))
    with pytest.raises(lib.EngineError, match="bn_x"):
        lib.call("fb_conv2d_wgrad", lib.C.byref(a))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,hw,imgs,G", [(128, 16, 32, 3), (512, 4, 128, 2), (256, 8, 16, 5)])
def test_bn_bwd_of_two_batchnorms_sharing_the_gradient(dtype, C, hw, imgs, G):
    """fb_bn_bwd_reduce2 / fb_bn_bwd_apply2: the BatchNorm of conv2 and the one of the shortcut convolution of a downsampling block both start from the
    gradient of the block output (through its ReLU mask): one read of dout per pass serves both, bit for bit what the separate calls give."""
    lib = _lib()
    handle = lib.load()
    dtc = lib.dtype_code(dtype)
    n = G * imgs
    px, ppg = n * hw * hw, imgs * hw * hw
    torch.manual_seed(C)
    eb = 4 if dtype == torch.float32 else 2
    xa, xb, dout = (torch.randn(n, hw, hw, C, device="cuda").to(dtype) for _ in range(3))
    bits = torch.randint(0, 256, (dout.numel() * eb // 16,), device="cuda", dtype=torch.uint8)
    mean_tab = torch.randn(G, 2 * C, device="cuda") * 0.1                               # layer a: columns [0, C), layer b: [C, 2C)
    inv = [torch.rand(G, C, device="cuda") + 0.5 for _ in range(2)]
    scale = [torch.rand(G, C, device="cuda") + 0.5 for _ in range(2)]
    rows = handle.fb_bn_bwd_reduce_rows(px, ppg)

    def finalize(part, k, gout, coef):
        lib.call("fb_bn_bwd_finalize", part.data_ptr(), rows, G, C, float(ppg), scale[k].data_ptr(), mean_tab.data_ptr(), inv[k].data_ptr(), 2 * C, k * C,
                 gout.data_ptr(), gout.data_ptr() + 4 * C, 2 * C, coef.data_ptr(), 0)

    out = {}
    for mode in ("separate", "dual"):
        parts = [torch.zeros(2 * rows * C, device="cuda") for _ in range(2)]
        gouts = [torch.zeros(G, 2 * C, device="cuda") for _ in range(2)]
        coefs = [torch.zeros(G, C, 3, device="cuda") for _ in range(2)]
        dxs = [torch.empty_like(dout) for _ in range(2)]
        if mode == "separate":
            for k, x in enumerate((xa, xb)):
                lib.call("fb_bn_bwd_reduce", dout.data_ptr(), None, bits.data_ptr(), x.data_ptr(), mean_tab.data_ptr(), inv[k].data_ptr(), 2 * C, k * C,
                         parts[k].data_ptr(), px, C, ppg, dtc)
                finalize(parts[k], k, gouts[k], coefs[k])
                lib.call("fb_bn_bwd_apply", dout.data_ptr(), None, bits.data_ptr(), x.data_ptr(), coefs[k].data_ptr(), dxs[k].data_ptr(), None, px, C, ppg, dtc, None, None)
        else:
            lib.call("fb_bn_bwd_reduce2", dout.data_ptr(), bits.data_ptr(), xa.data_ptr(), inv[0].data_ptr(), 0, parts[0].data_ptr(), xb.data_ptr(), inv[1].data_ptr(), C,
                     parts[1].data_ptr(), mean_tab.data_ptr(), 2 * C, px, C, ppg, dtc)
            for k in range(2):
                finalize(parts[k], k, gouts[k], coefs[k])
            lib.call("fb_bn_bwd_apply2", dout.data_ptr(), bits.data_ptr(), xa.data_ptr(), coefs[0].data_ptr(), dxs[0].data_ptr(), xb.data_ptr(), coefs[1].data_ptr(),
                     dxs[1].data_ptr(), px, C, ppg, dtc)
        torch.cuda.synchronize()
        out[mode] = (parts, gouts, coefs, dxs)
    for group in range(4):
        for k in range(2):
            assert torch.equal(out["separate"][group][k], out["dual"][group][k]), (group, k)
    assert float(out["dual"][3][0].float().abs().max()) > 0 and not torch.equal(out["dual"][3][0], out["dual"][3][1])
