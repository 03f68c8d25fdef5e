// Step timeline of the ping-pong 1x1 GEMM kernel (development tool):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFB_C1G_TRACE -Ifullbatchtraining_amd/csrc -Iinclude tools/g1_trace.hip -o tools/scratch/g1_trace.bin
//   FB_C1G=1 tools/scratch/g1_trace.bin <K> <Cd> <pixels>
long long* g_g1_trace = nullptr;
#include "../fullbatchtraining_amd/csrc/conv1x1_gemm.hip"
thread_local char fb_err_buf[512] = "";
int fb_persistent_cus() { return 256; }
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    const int K = atoi(argv[1]), Cd = atoi(argv[2]); const long long M = atoll(argv[3]);
    void *x, *w, *y;
    hipMalloc(&x, (size_t)M * K * 2); hipMalloc(&w, (size_t)K * Cd * 2); hipMalloc(&y, (size_t)M * Cd * 2);
    hipMemset(x, 0x3c, (size_t)M * K * 2); hipMemset(w, 0x3c, (size_t)K * Cd * 2);
    hipMalloc(&g_g1_trace, 2 * 16 * 8 * 8);
    hipMemset(g_g1_trace, 0, 2 * 16 * 8 * 8);
    fb_conv_args a = {};
    a.src = x; a.wgt = w; a.dst = y; a.n_img = 1; a.Hs = a.Hd = 1; a.Ws = a.Wd = (int)M; a.Cs = K; a.Cd = Cd; a.R = a.S = 1; a.stride = 1; a.pad = 0; a.mode = 0; a.dtype = FB_BF16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0, 0);
        if (!fb_try_conv1x1_gemm(&a, 0)) { printf("not handled (FB_C1G=1?)\n"); return 1; }
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> t(2 * 16 * 8);
    hipMemcpy(t.data(), g_g1_trace, t.size() * 8, hipMemcpyDeviceToHost);
    printf("K=%d Cd=%d M=%lld: kernel %.1f us\n", K, Cd, M, ms * 1e3);
    const long long t0 = t[0];
    const char* names[7] = {"top", "waited", "barrier", "dma issued", "reads+half0", "mid barrier", "half1 issued"};
    for (int g = 0; g < 2; ++g) {
        printf("group %c (cycles since E's first stamp; per step: ", g ? 'L' : 'E');
        for (int k = 0; k < 7; ++k) printf("%s%s", names[k], k < 6 ? " | " : ")\n");
        for (int s = 0; s < 16; ++s) {
            printf("  step %2d:", s);
            for (int k = 0; k < 7; ++k) printf(" %7lld", t[(g * 16 + s) * 8 + k] - t0);
            printf("\n");
        }
    }
    return 0;
}
