"""fp32 convolutions with pre-split operands (FB_F32P planes, csrc/conv_igemm_planes.hip) against the in-kernel split (conv_igemm_v3_kernel<f32s_tag>):
same bits, time per launch.  ResNet-152 @224 shapes, N images."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fullbatchtraining_amd import lib
from fullbatchtraining_amd.lib import call

N = int(os.environ.get("N", "512"))
lib.load()
FB_F32, FB_F32P = 0, 2


def planes(t, rows, C):
    out = torch.empty(rows, 3 * C, device="cuda", dtype=torch.bfloat16)
    call("fb_planes_from_f32", t.data_ptr(), out.data_ptr(), rows, C)
    return out


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000


shapes = [  # (cin, cout, k, stride, hw_in, mode, addend)
    (64, 256, 1, 1, 56, 0, 0), (256, 64, 1, 1, 56, 0, 0), (128, 512, 1, 1, 28, 0, 0), (512, 128, 1, 1, 28, 0, 0), (256, 1024, 1, 1, 14, 0, 0), (1024, 256, 1, 1, 14, 0, 0),
    (512, 2048, 1, 1, 7, 0, 0), (2048, 512, 1, 1, 7, 0, 0), (64, 64, 3, 1, 56, 0, 0), (128, 128, 3, 1, 28, 0, 0), (256, 256, 3, 1, 14, 0, 0), (512, 512, 3, 1, 7, 0, 0),
    (128, 128, 3, 2, 56, 0, 0), (256, 1024, 1, 1, 14, 1, 0), (1024, 256, 1, 1, 14, 1, 1), (256, 256, 3, 1, 14, 1, 0), (256, 256, 3, 2, 28, 1, 2)]
print(f"N = {N} images; us per launch: in-kernel split (f32s) / planes; TFLOP/s fp32-equivalent")
tot_a = tot_b = 0.0
for cin, cout, k, stride, hw, mode, addend in shapes:
    torch.manual_seed(cin + cout + k)
    pad = k // 2
    ho = (hw + 2 * pad - k) // stride + 1
    if mode == 0:
        src = torch.randn(N, hw, hw, cin, device="cuda")
        w = torch.randn(cout, k * k, cin, device="cuda") * 0.05
        dst_shape, cs, cd, hs, hd = (N, ho, ho, cout), cin, cout, hw, ho
    else:   # input gradient: src = dY [ho x ho x cout] -> dst = dX [hw x hw x cin]; weights [ci][tap][co]
        src = torch.randn(N, ho, ho, cout, device="cuda")
        w = torch.randn(cin, k * k, cout, device="cuda") * 0.05
        dst_shape, cs, cd, hs, hd = (N, hw, hw, cin), cout, cin, ho, hw
    add = None
    if addend == 1:
        add = torch.randn(*dst_shape, device="cuda")
    elif addend == 2:
        add = torch.randn(N, hw // 2, hw // 2, cin, device="cuda")
    stat = torch.zeros(2, (N * dst_shape[1] * dst_shape[2] + 127) // 128, cd, device="cuda") if mode == 0 else None
    a, b = torch.empty(*dst_shape, device="cuda"), torch.full(dst_shape, float("nan"), device="cuda")
    src_p = planes(src, src.numel() // cs, cs)
    w_p = planes(w, w.shape[0] * w.shape[1], cs)
    stat_b = torch.zeros_like(stat) if stat is not None else None

    def run(s, wt, d, st, dt):
        args = lib.ConvArgs(s.data_ptr(), wt.data_ptr(), d.data_ptr(), add.data_ptr() if add is not None else None, st.data_ptr() if st is not None else None,
                            N, hs, hs, cs, hd, hd, cd, k, k, stride, pad, mode, 0, 0, addend, dt, None, None, None, None, None, 0)
        call("fb_conv2d", lib.C.byref(args))

    ta = timeit(lambda: run(src, w, a, stat, FB_F32))
    tb = timeit(lambda: run(src_p, w_p, b, stat_b, FB_F32P))
    same = torch.equal(a, b) and (stat is None or torch.equal(stat, stat_b))
    flop = 2.0 * N * ho * ho * cout * k * k * cin / (4 if (mode == 1 and stride == 2) else 1) * (4 if (mode == 1 and stride == 2) else 1)
    tot_a += ta; tot_b += tb
    print(f"{'fwd' if mode == 0 else 'dgrad'} {cin}->{cout} k{k} s{stride} @{hw}{' +add' + str(addend) if addend else ''}: {ta:8.1f} / {tb:8.1f} us  ({flop / ta / 1e6:6.1f} / {flop / tb / 1e6:6.1f} TF/s)  "
          f"x{ta / tb:.2f}  same bits: {same}", flush=True)
    if not same:
        print("   max abs diff", float((a - b).abs().max()), "rel", float((a - b).norm() / a.norm()))
print(f"sum: {tot_a:.0f} / {tot_b:.0f} us  x{tot_a / tot_b:.2f}")
