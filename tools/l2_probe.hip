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

// mode 2: the access order of a GEMM's K-loop over a row-major [rows][row_bytes] operand: a tile = 256 rows; a K-step loads `piece` bytes of every row of the tile
// (one instruction = 1 KiB = 1024 / piece rows); the K-steps walk along the rows; then the next tile.  piece == row_bytes: whole rows (a streaming kernel).
template <int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void gemm_order_kernel(const char* __restrict__ src, long long region, int row_bytes, int piece, int shared_iters, unsigned* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) char lds[NW * DEPTH * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (shared_iters ? 0 : (long long)blockIdx.x * region);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)region, 0x00020000);
    const int tile_bytes = 256 * row_bytes, n_tiles = (int)(region / tile_bytes);
    const int lanes_per_row = piece / 16, rows_per_instr = 64 / lanes_per_row;
    const int instr_per_step = 256 / rows_per_instr;                         // per workgroup
    const unsigned lane_off = (unsigned)((lane / lanes_per_row) * row_bytes + (lane % lanes_per_row) * 16);
    for (int it = 0; it < (shared_iters ? shared_iters : 1); ++it)
    for (int t = 0; t < n_tiles; ++t)
        for (int ks = 0; ks < row_bytes / piece; ++ks)
            for (int i = wave * DEPTH; i < instr_per_step; i += NW * DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (wave * DEPTH + d) * 1024), 16,
                                                             lane_off + (unsigned)((i + d) * rows_per_instr * row_bytes), t * tile_bytes + ks * piece, 0, 0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
            }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (*(unsigned*)(lds + tid * 4) == 0x12345678u) sink[tid] = 1;
}

template <int NW, int DEPTH>
static void run_gemm_order(const char* src, int wgs, long long region, int row_bytes, int piece, unsigned* sink, int shared_iters = 0) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((gemm_order_kernel<NW, DEPTH>), dim3(wgs), dim3(NW * 64), 0, 0, src, region, row_bytes, piece, shared_iters, sink);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("gemm order: waves %d depth %d wgs %d region %lld KB %s rows of %4d B in pieces of %4d B : %7.1f us  %6.2f TB/s  %6.1f GB/s per WG\n", NW, DEPTH, wgs, region / 1024,
           shared_iters ? "shared " : "private", row_bytes, piece, best * 1e3, (double)wgs * region * (shared_iters ? shared_iters : 1) / best / 1e9,
           (double)region * (shared_iters ? shared_iters : 1) / best / 1e6);
}

// mode 3: the load structure of a GEMM K-loop with a ring of LDS stages: per step every wave waits for ITS pieces of the oldest round (counted wait), a
// workgroup barrier, then every wave issues its PW pieces of the round AHEAD steps ahead.  No compute.  Sequential sweep of a private region.
template <int NW, int PW, int AHEAD>
__global__ __launch_bounds__(NW * 64) void ring_kernel(const char* __restrict__ src, long long region, unsigned* __restrict__ sink) {
    constexpr int STAGE = NW * PW * 1024, NSTAGE = AHEAD + 1;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (long long)blockIdx.x * region;
    const int n_steps = (int)(region / STAGE);
    auto issue = [&](int step) {
        const long long off = (long long)(step % n_steps) * STAGE;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + off), 0, STAGE, 0x00020000);
        char* st = lds + (step % NSTAGE) * STAGE;
#pragma unroll
        for (int i = 0; i < PW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(st + (wave * PW + i) * 1024), 16, (unsigned)((wave * PW + i) * 1024 + lane * 16), 0, 0, 0);
    };
    for (int a = 0; a < AHEAD; ++a) issue(a);
    for (int step = 0; step < n_steps; ++step) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PW) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(step + AHEAD);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (*(unsigned*)(lds + tid * 4) == 0x12345678u) sink[tid] = 1;
}

template <int NW, int PW, int AHEAD>
static void run_ring(const char* src, int wgs, unsigned* sink) {
    constexpr int STAGE = NW * PW * 1024, LDS = (AHEAD + 1) * STAGE;
    if (LDS * (wgs / 256) > 160 * 1024) return;
    const long long region = (2048ll << 20) / wgs / (1 << 20) * (1 << 20);
    hipFuncSetAttribute((const void*)ring_kernel<NW, PW, AHEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((ring_kernel<NW, PW, AHEAD>), dim3(wgs), dim3(NW * 64), LDS, 0, src, region, sink);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("ring: waves %d stage %3d KB rounds ahead %d (LDS %3d KB) wgs %d : %7.1f us  %6.2f TB/s\n", NW, STAGE / 1024, AHEAD, LDS / 1024, wgs, best * 1e3, (double)wgs * region / best / 1e9);
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
    if (getenv("RING")) {
        for (int wgs : {256, 512}) {
            run_ring<8, 6, 2>(src, wgs, sink);   // the GEMM kernel: 48 KB stages, two rounds ahead
            run_ring<8, 4, 1>(src, wgs, sink);   // 32 KB stages
            run_ring<8, 4, 2>(src, wgs, sink);
            run_ring<8, 4, 3>(src, wgs, sink);
            run_ring<8, 4, 4>(src, wgs, sink);
            run_ring<8, 2, 1>(src, wgs, sink);   // 16 KB stages
            run_ring<8, 2, 2>(src, wgs, sink);
            run_ring<8, 2, 3>(src, wgs, sink);
            run_ring<8, 2, 4>(src, wgs, sink);
            run_ring<8, 2, 6>(src, wgs, sink);
            run_ring<8, 2, 8>(src, wgs, sink);
            run_ring<8, 1, 2>(src, wgs, sink);   // 8 KB stages
            run_ring<8, 1, 4>(src, wgs, sink);
            run_ring<8, 1, 8>(src, wgs, sink);
            run_ring<8, 1, 12>(src, wgs, sink);
            run_ring<4, 4, 1>(src, wgs, sink);   // 4 waves, 16 KB stages
            run_ring<4, 4, 2>(src, wgs, sink);
            run_ring<4, 4, 4>(src, wgs, sink);
            run_ring<4, 8, 1>(src, wgs, sink);   // 4 waves, 32 KB stages
            run_ring<4, 8, 2>(src, wgs, sink);
        }
        return 0;
    }
    if (getenv("GEMM_L2")) {                                                  // the same orders out of L2: everybody sweeps the same 1 MB
        for (int row_bytes : {512, 1024, 2048})
            for (int piece : {128, 256, 512, 1024}) {
                if (piece > row_bytes) continue;
                run_gemm_order<8, 4>(src, 256, 1 << 20, row_bytes, piece, sink, 64);
                run_gemm_order<4, 4>(src, 256, 1 << 20, row_bytes, piece, sink, 64);
                run_gemm_order<8, 2>(src, 256, 1 << 20, row_bytes, piece, sink, 64);
            }
        return 0;
    }
    if (getenv("GEMM_ORDER")) {
        for (int wgs : {256, 512, 768})
            for (int row_bytes : {512, 1024, 2048, 4096})
                for (int piece : {128, 256, 512, row_bytes}) {
                    if (piece > row_bytes || (piece == 512 && row_bytes == 512 && piece != row_bytes)) continue;
                    run_gemm_order<8, 4>(src, wgs, (2048ll << 20) / wgs / (256 * 4096) * (256 * 4096), row_bytes, piece, sink);
                }
        return 0;
    }
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
