// Foreign-LDS-write detector (development tool, built on the GPU box against the in-tree library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/lds_canary.hip -Lfullbatchtraining_amd/csrc -lfbengine -Wl,-rpath,$PWD/fullbatchtraining_amd/csrc -o /tmp/lds_canary
//   /tmp/lds_canary <iters> <use_dgrad> <use_wgrad> [canary_kib] [n_img]
// Stream A: input-gradient convolution (64 -> 64, 32x32: the resident-filter kernel) followed by a "canary" kernel whose workgroups fill their
// LDS with a pattern, wait a few microseconds and verify it.  Stream B: the 3x3 weight-gradient kernel of the same layer, free-running.
// Any LDS word that changes under a canary workgroup was written by somebody else (e.g. an LDS-DMA request of a workgroup that has ended).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fb_engine.h"

struct Report { unsigned long long bad_words; unsigned first_off, first_val, first_expected, first_block; unsigned long long checks; };

__global__ __launch_bounds__(256) void canary_kernel(Report* rep, int words, int spins, unsigned salt) {
    extern __shared__ unsigned lds[];
    const unsigned tag = salt * 2654435761u + blockIdx.x * 40503u;
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = tag ^ (unsigned)i ^ 0xA5A50000u;
    __syncthreads();
    for (int s = 0; s < spins; ++s) {
        for (int i = threadIdx.x; i < words; i += 256) {
            const unsigned v = ((volatile unsigned*)lds)[i], e = tag ^ (unsigned)i ^ 0xA5A50000u;
            if (v != e) {
                if (atomicAdd(&rep->bad_words, 1ull) == 0) { rep->first_off = i * 4; rep->first_val = v; rep->first_expected = e; rep->first_block = blockIdx.x; }
                ((volatile unsigned*)lds)[i] = e;
            }
        }
        __builtin_amdgcn_s_sleep(20);
    }
    if (threadIdx.x == 0) atomicAdd(&rep->checks, 1ull);
}

// visibility check: counts the 16-byte vectors of `buf` that still hold the 0xFF fill written before the convolution
__global__ __launch_bounds__(256) void stale_kernel(const uint4* buf, long long n_vec, Report* rep) {
    unsigned long long bad = 0;
    long long first = -1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (long long)gridDim.x * 256) {
        const uint4 v = buf[i];
        if (v.x == 0xFFFFFFFFu || v.y == 0xFFFFFFFFu || v.z == 0xFFFFFFFFu || v.w == 0xFFFFFFFFu) { ++bad; if (first < 0) first = i; }
    }
    if (bad) {
        if (atomicAdd(&rep->bad_words, bad) == 0) { rep->first_off = (unsigned)(first & 0xffffffff); rep->first_val = (unsigned)(first >> 32); rep->first_block = blockIdx.x; }
    }
    if (threadIdx.x == 0) atomicAdd(&rep->checks, 1ull);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200, use_dgrad = argc > 2 ? atoi(argv[2]) : 1, use_wgrad = argc > 3 ? atoi(argv[3]) : 1;
    const int kib = argc > 4 ? atoi(argv[4]) : 16, n = argc > 5 ? atoi(argv[5]) : 384;
    const int W = 32, C = 64, ipg = 128, split = 64;
    const size_t act_b = (size_t)n * W * W * C * 2, w_b = (size_t)C * 9 * C * 2;
    void *dy, *w, *dx, *x2, *dy2; float* slab; Report* rep;
    hipMalloc(&dy, act_b); hipMalloc(&w, w_b); hipMalloc(&dx, act_b); hipMalloc(&x2, act_b); hipMalloc(&dy2, act_b);
    hipMalloc(&slab, (size_t)(n / ipg) * split * C * 9 * C * 4); hipMalloc(&rep, sizeof(Report));
    hipMemset(dy, 0x3c, act_b); hipMemset(w, 0x3c, w_b); hipMemset(x2, 0x3c, act_b); hipMemset(dy2, 0x3c, act_b); hipMemset(rep, 0, sizeof(Report));
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    const int null_a = argc > 6 ? atoi(argv[6]) : 0, events = argc > 7 ? atoi(argv[7]) : 0;
    if (null_a) sa = nullptr;                               // stream A = the legacy default stream (what the engine's main stream is under torch)
    hipEvent_t ev_a, ev_b;
    hipEventCreateWithFlags(&ev_a, hipEventDisableTiming); hipEventCreateWithFlags(&ev_b, hipEventDisableTiming);
    fb_conv_args a = {};
    a.src = dy; a.wgt = w; a.dst = dx; a.n_img = n; a.Hs = a.Ws = a.Hd = a.Wd = W; a.Cs = a.Cd = C; a.R = a.S = 3; a.stride = 1; a.pad = 1; a.mode = 1;
    a.dtype = FB_BF16;
    fb_wgrad_args g = {};
    g.x = x2; g.dy = dy2; g.dw_partial = slab; g.n_img = n; g.Hs = g.Ws = g.Hd = g.Wd = W; g.Cs = g.Cd = C; g.R = g.S = 3; g.stride = 1; g.pad = 1;
    g.imgs_per_group = ipg; g.split_k = split; g.dtype = FB_BF16;
    hipFuncSetAttribute((const void*)canary_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int it = 0; it < iters; ++it) {
        if (events) { hipEventRecord(ev_a, sa); hipStreamWaitEvent(sb, ev_a, 0); }      // (the engine: B waits for A's producer ...)
        if (use_wgrad && fb_conv2d_wgrad(&g, sb) != 0) { printf("wgrad failed: %s\n", fb_last_error_string()); return 1; }
        if (kib == 0) hipMemsetAsync(dx, 0xFF, act_b, sa);
        if (use_dgrad && fb_conv2d(&a, sa) != 0) { printf("conv failed: %s\n", fb_last_error_string()); return 1; }
        if (kib == 0) hipLaunchKernelGGL(stale_kernel, dim3(3072), dim3(256), 0, sa, (const uint4*)dx, (long long)(act_b / 16), rep);
        else hipLaunchKernelGGL(canary_kernel, dim3(3072), dim3(256), (size_t)kib * 1024, sa, rep, kib * 256, 6, (unsigned)it);
        if (events) { hipEventRecord(ev_b, sb); if (it % 3 == 2) hipStreamWaitEvent(sa, ev_b, 0); }      // (... and A joins B now and then)
    }
    hipDeviceSynchronize();
    Report r;
    hipMemcpy(&r, rep, sizeof(r), hipMemcpyDeviceToHost);
    printf("null_stream=%d events=%d ", null_a, events);
    printf("dgrad=%d wgrad=%d canary=%d KiB n=%d iters=%d: %llu corrupted LDS words in %llu canary workgroups", use_dgrad, use_wgrad, kib, n, iters, r.bad_words, r.checks);
    if (r.bad_words && kib == 0) printf("  [canary 0 KiB = visibility check: vectors of the convolution output still holding the pre-fill; first vector %llu of %llu]",
                                        ((unsigned long long)r.first_val << 32) | r.first_off, (unsigned long long)(act_b / 16));
    else if (r.bad_words) printf("  (first: block %u byte offset %u value 0x%08x expected 0x%08x)", r.first_block, r.first_off, r.first_val, r.first_expected);
    printf("\n");
    return 0;
}
