// On-chip load bandwidth probe (development tool): what a CU can pull from L2 / Infinity Cache / HBM with LDS-DMA or register loads.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_probe.hip -o /tmp/l2_probe && /tmp/l2_probe
// Every workgroup (NW waves) sweeps a region of `region` bytes `iters` times with 16-byte-per-lane loads (1 KiB per wave instruction):
//   mode 0: buffer_load_dwordx4 ... lds (LDS-DMA) into a ring in LDS          mode 1: buffer_load_dwordx4 into registers (xor-reduced)
// `shared` = 0: every workgroup its own region (working set = workgroups x region); 1: the workgroups of an XCD-slot share one region.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE, int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void probe_kernel(const char* __restrict__ src, long long region, long long stride, int iters, unsigned* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) char lds[NW * DEPTH * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (long long)blockIdx.x * stride;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)region, 0x00020000);
    const int per_sweep = (int)(region / (NW * 1024));                      // wave instructions per wave and sweep
    u32x4 acc = {0u, 0u, 0u, 0u};
    if constexpr (MODE == 0) {
        for (int it = 0; it < iters; ++it)
            for (int i = 0; i < per_sweep; i += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (wave * DEPTH + d) * 1024), 16,
                                                             (unsigned)(((i + d) * NW + wave) * 1024 + lane * 16), 0, 0, 0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");      // (half a batch stays in flight)
            }
    } else {
        u32x4 va[DEPTH], vb[DEPTH];                  // two batches in flight alternately
        auto issue = [&](u32x4 (&v)[DEPTH], int i) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const unsigned off = (unsigned)((((i + d) % per_sweep) * NW + wave) * 1024 + lane * 16);
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[d]) : "v"(off), "s"(rs) : "memory");
            }
        };
        auto consume = [&](u32x4 (&v)[DEPTH]) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v[d]) : "n"(DEPTH) : "memory"); acc ^= v[d]; }
        };
        issue(va, 0);
        for (int it = 0; it < iters; ++it)
            for (int i = 0; i < per_sweep; i += 2 * DEPTH) {
                issue(vb, i + DEPTH);
                consume(va);
                issue(va, i + 2 * DEPTH);
                consume(vb);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (MODE == 0) acc[0] = *(unsigned*)(lds + tid * 4);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[tid] = acc[0];
}

template <int MODE, int NW, int DEPTH>
static void run(const char* src, size_t buf_bytes, int wgs, long long region, bool shared, int iters, unsigned* sink) {
    const long long stride = shared ? 0 : region;
    if (!shared && (size_t)wgs * region > buf_bytes) { printf("skip\n"); return; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe_kernel<MODE, NW, DEPTH>), dim3(wgs), dim3(NW * 64), 0, 0, src, region, stride, iters, sink);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)wgs * region * iters;
    printf("mode %d waves %d depth %2d wgs %4d region %7lld KB %s set %8.1f MB : %7.1f us  %6.2f TB/s  %6.1f GB/s per WG\n", MODE, NW, DEPTH, wgs, region / 1024,
           shared ? "shared " : "private", (shared ? 1.0 : (double)wgs) * region / 1e6, best * 1e3, bytes / best / 1e9, bytes / wgs / best / 1e6);
}

int main() {
    const size_t buf = 2048ull << 20;
    char* src; unsigned* sink;
    hipMalloc(&src, buf); hipMalloc(&sink, 4096);
    hipMemset(src, 1, buf);
    for (int wgs : {256, 512}) {
        // L2-resident: 256 x 32 KB = 8 MB (1 MB per XCD); Infinity-Cache-resident: 256 x 512 KB = 128 MB; HBM: 256 x 4 MB = 1 GB
        for (long long region : {32ll << 10, 512ll << 10, 4096ll << 10}) {
            const int iters = (int)((64ll << 20) / region);                  // 64 MB per workgroup
            run<0, 8, 4>(src, buf, wgs, region, false, iters, sink);
            run<0, 8, 8>(src, buf, wgs, region, false, iters, sink);
            run<0, 4, 8>(src, buf, wgs, region, false, iters, sink);
            run<1, 8, 4>(src, buf, wgs, region, false, iters, sink);
            run<1, 8, 8>(src, buf, wgs, region, false, iters, sink);
            run<1, 4, 8>(src, buf, wgs, region, false, iters, sink);
        }
        run<0, 8, 8>(src, buf, wgs, 512ll << 10, true, 128, sink);           // everybody reads the same 512 KB (a filter)
        run<1, 8, 8>(src, buf, wgs, 512ll << 10, true, 128, sink);
    }
    return 0;
}
