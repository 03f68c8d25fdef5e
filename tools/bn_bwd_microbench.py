"""BatchNorm backward: the two-pass form (fb_bn_bwd_reduce -> finalize -> apply) against the one-pass cluster kernel (fb_bn_bwd_fused) on the
benchmark's shapes: time per layer and agreement of dx / dgamma / dbeta.  GPU box:  python tools/bn_bwd_microbench.py [G]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fullbatchtraining_amd import lib


def bench(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 98
    dt = torch.float32 if os.environ.get("DT") == "f32" else torch.bfloat16
    dtc = lib.dtype_code(dt)
    eb = 4 if dt == torch.float32 else 2
    handle = lib.load()
    print(f"| C | map | two-pass us (reduce + finalize + apply) | fused us | GB/s fused (3 passes) | dx rel | dgamma rel | dbeta rel |\n|---|---|---|---|---|---|---|---|")
    for C, hw in ((64, 32), (128, 16), (256, 8), (512, 4)):
        n = G * 128
        px, ppg = n * hw * hw, 128 * hw * hw
        torch.manual_seed(C)
        x = torch.randn(n, hw, hw, C, device="cuda").to(dt)
        dout = torch.randn(n, hw, hw, C, device="cuda").to(dt)
        bits = torch.randint(0, 256, (x.numel() * eb // 16,), device="cuda", dtype=torch.uint8)
        mean_tab, invstd, scale = torch.randn(G, C, device="cuda") * 0.1, torch.rand(G, C, device="cuda") + 0.5, torch.rand(G, C, device="cuda") + 0.5
        out = {}
        for mode in ("two-pass", "fused"):
            dx, dy = torch.empty_like(x), torch.empty_like(x)
            gout = torch.zeros(G, 2 * C, device="cuda")
            coef = torch.zeros(G, C, 3, device="cuda")
            if mode == "two-pass":
                rows = handle.fb_bn_bwd_reduce_rows(px, ppg)
                ws = torch.zeros(int(handle.fb_ws_bn_partial_floats(px, C)), device="cuda")

                def run(dy_out=None):
                    lib.call("fb_bn_bwd_reduce", dout.data_ptr(), None, bits.data_ptr(), x.data_ptr(), mean_tab.data_ptr(), invstd.data_ptr(), C, 0, ws.data_ptr(),
                             px, C, ppg, dtc)
                    lib.call("fb_bn_bwd_finalize", ws.data_ptr(), rows, G, C, float(ppg), scale.data_ptr(), mean_tab.data_ptr(), invstd.data_ptr(), C, 0,
                             gout.data_ptr(), gout.data_ptr() + 4 * C, 2 * C, coef.data_ptr(), 0)
                    lib.call("fb_bn_bwd_apply", dout.data_ptr(), None, bits.data_ptr(), x.data_ptr(), coef.data_ptr(), dx.data_ptr(), dy_out, px, C, ppg, dtc, None, None)
            else:
                assert handle.fb_bn_bwd_fused_supported(px, C, ppg, dtc), (C, hw)
                ws = torch.zeros(int(handle.fb_ws_bn_bwd_fused_floats(px, C, ppg, dtc)), device="cuda")
                sync = torch.zeros(int(handle.fb_ws_bn_bwd_fused_ints(G)), device="cuda", dtype=torch.int32)

                def run(dy_out=None):
                    lib.call("fb_bn_bwd_fused", dout.data_ptr(), bits.data_ptr(), x.data_ptr(), mean_tab.data_ptr(), invstd.data_ptr(), scale.data_ptr(), C, 0,
                             gout.data_ptr(), gout.data_ptr() + 4 * C, 2 * C, coef.data_ptr(), dx.data_ptr(), dy_out, px, C, ppg, float(ppg), dtc, ws.data_ptr(),
                             sync.data_ptr())
            t = bench(run)
            run(dy.data_ptr())
            torch.cuda.synchronize()
            if mode == "fused":
                assert int(sync[-1]) == 0, "cluster time-out"
                if os.environ.get("TRACE"):
                    import ctypes
                    ncw = int(handle.fb_ws_bn_bwd_fused_floats(px, C, ppg, dtc)) // G // (2 * C) - 0
                    ncw = (int(handle.fb_ws_bn_bwd_fused_floats(px, C, ppg, dtc)) // G - 4 * C) // (2 * C)
                    tr = torch.zeros(G, ncw, 6, device="cuda", dtype=torch.int64)
                    handle.fb_bn_bwd_fused_trace.argtypes = [ctypes.c_void_p]
                    handle.fb_bn_bwd_fused_trace(tr.data_ptr())
                    run()
                    torch.cuda.synchronize()
                    handle.fb_bn_bwd_fused_trace(None)
                    tc = tr.cpu().double() / 100.0                      # us
                    t0 = tc[..., 0].min()
                    ph = [(tc[..., k + 1] - tc[..., k]).mean().item() for k in range(4)]
                    print(f"    C={C}: {ncw} workgroups per chunk; mean us per workgroup and chunk: load+sum {ph[0]:.1f}, arrive {ph[1]:.1f}, wait/reduce {ph[2]:.1f}, store {ph[3]:.1f}; "
                          f"kernel span {(tc[..., 4].max() - t0).item():.0f} us")
                    red = tc[..., 5] > 0
                    if red.any():
                        print(f"    reducers: arrive->coefficients {(tc[..., 3] - tc[..., 2])[red].mean().item():.1f} us; "
                              f"chunk timeline of chunk 0/1/2 (first load start, last load end, last arrival, coefficients, last store) us: "
                              + "; ".join(", ".join(f"{v:.0f}" for v in (tc[g, :, 0].min() - t0, tc[g, :, 1].max() - t0, tc[g, :, 2].max() - t0, tc[g, :, 3].min() - t0, tc[g, :, 4].max() - t0)) for g in range(3)))
            out[mode] = (t, dx.float().clone(), gout.clone(), dy.float().clone())
        (t2, dx2, g2, dy2), (t1, dx1, g1, dy1) = out["two-pass"], out["fused"]
        rel = lambda a, b: float((a - b).norm() / b.norm())
        assert torch.equal(dy1, dy2) or os.environ.get('FB_BNF_POLL') == '-1'
        print(f"| {C} | {hw}x{hw} | {t2:.0f} | {t1:.0f} | {3 * x.numel() * eb / t1 / 1e3:.0f} | {rel(dx1, dx2):.1e} | {rel(g1[:, :C], g2[:, :C]):.1e} | {rel(g1[:, C:], g2[:, C:]):.1e} |",
              flush=True)


if __name__ == "__main__":
    main()
