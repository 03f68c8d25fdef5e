// Phase timeline of the persistent halo4 convolution kernel (development tool, built on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFB_H4_TRACE -Ifullbatchtraining_amd/csrc -Iinclude tools/h4_trace.hip -o /tmp/h4_trace
//   /tmp/h4_trace <W> <Cin> <Cout> <n_img> [mode]
long long* g_h4_trace = nullptr;
#include "../fullbatchtraining_amd/csrc/conv3x3_halo4.hip"
thread_local char fb_err_buf[512] = "";                    // (the library's runtime.cpp is not linked into this tool)
bool fb_f32_split_enabled() { return true; }
int fb_persistent_cus() { return 256; }
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

int main(int argc, char** argv) {
    const int W = atoi(argv[1]), Cin = atoi(argv[2]), Cout = atoi(argv[3]), n = atoi(argv[4]), mode = argc > 5 ? atoi(argv[5]) : 0;
    const size_t in_b = (size_t)n * W * W * Cin * 2, out_b = (size_t)n * W * W * Cout * 2, w_b = (size_t)Cout * 9 * Cin * 2;
    void *x, *w, *y; float* stat;
    hipMalloc(&x, in_b); hipMalloc(&w, w_b); hipMalloc(&y, out_b); hipMalloc(&stat, (size_t)2 * (n * W * W / 128) * Cout * 4);
    hipMemset(x, 0x3c, in_b); hipMemset(w, 0x3c, w_b);
    const int n_blocks = n * W * W / 256 * (Cout / 64);   // = tiles: one trace row per tile
    hipMalloc(&g_h4_trace, (size_t)n_blocks * 64);
    fb_conv_args a = {};
    a.src = x; a.wgt = w; a.dst = y; a.stat_partial = mode == 0 ? stat : nullptr;
    a.n_img = n; a.Hs = a.Ws = a.Hd = a.Wd = W; a.Cs = Cin; a.Cd = Cout; a.R = a.S = 3; a.stride = 1; a.pad = 1; a.mode = mode;
    a.dtype = FB_BF16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0, 0);
        if (!fb_try_conv3x3_halo4(&a, 0)) { printf("not handled\n"); return 1; }
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> t((size_t)n_blocks * 8);
    hipMemcpy(t.data(), g_h4_trace, t.size() * 8, hipMemcpyDeviceToHost);
    long long tmin = t[0], tmax = 0;
    double ph[5] = {0, 0, 0, 0, 0}, life = 0;
    for (int b = 0; b < n_blocks; ++b) {
        const long long* r = &t[(size_t)b * 8];
        tmin = std::min(tmin, r[0]); tmax = std::max(tmax, r[5]);
        for (int k = 0; k < 5; ++k) ph[k] += (double)(r[k + 1] - r[k]);
        life += (double)(r[5] - r[0]);
    }
    const double span = (double)(tmax - tmin);
    const double tick_us = 0.01;                 // wall_clock64: 100 MHz
    printf("W=%d Cin=%d Cout=%d n=%d mode=%d: %d blocks, kernel %.1f us, span %.0f ticks (%.4f us/tick)\n", W, Cin, Cout, n, mode, n_blocks, ms * 1e3, span, tick_us);
    const char* names[5] = {"prologue (addresses, issue)", "wait halo + weights", "tap loop", "stores", "stats"};
    for (int k = 0; k < 5; ++k) printf("  %-28s %8.2f us\n", names[k], ph[k] / n_blocks * tick_us);
    printf("  %-28s %8.2f us   (sum of lifetimes / span = %.1f blocks in flight)\n", "block lifetime", life / n_blocks * tick_us, life / span);
    {   // shader clock during the kernel: s_memtime ticks per 100 MHz wall tick
        double cyc = 0, wall = 0;
        for (int b = 0; b < n_blocks; ++b) { const long long* r = &t[(size_t)b * 8]; cyc += (double)(r[7] - r[6]); wall += (double)(r[5] - r[0]); }
        printf("  shader clock while tiles run: %.0f MHz\n", cyc / wall * 100.0);
    }
    // gaps between consecutive blocks on the same CU slot are not visible here; report dispatch spread
    std::vector<long long> starts(n_blocks);
    for (int b = 0; b < n_blocks; ++b) starts[b] = t[(size_t)b * 8];
    std::sort(starts.begin(), starts.end());
    printf("  first-wave start times: 10%%=%.1f us 50%%=%.1f us 90%%=%.1f us of %.1f\n", (starts[n_blocks / 10] - tmin) * tick_us,
           (starts[n_blocks / 2] - tmin) * tick_us, (starts[n_blocks * 9 / 10] - tmin) * tick_us, span * tick_us);
    return 0;
}
