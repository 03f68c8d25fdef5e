// 1x1 convolutions with K >= 512 input channels (the channel-reducing convolutions of the Bottleneck blocks and their input gradients,
// reference resnets.py:289-291: 1024 -> 256 @14x14, 512 -> 128 @28x28, 2048 -> 512 @7x7 ...), bf16, forward and input gradient.
//
// These went to the implicit GEMM (conv_igemm_glds.hip: 128 x 128 tiles, ONE 32 KiB stage, three workgroups per CU), where the counters show a
// latency-bound kernel (round 5, 1024 -> 256 @14x14, 1024 images, 155 us = 678 TFLOP/s: 53 % of the wave cycles in s_waitcnt / s_barrier, matrix
// pipe busy a third of the time): every K-step waits for the loads it has just issued, and a tile lives through 16 of them.  Here:
//   * persistent workgroups (one per CU, 8 waves) walk a list of 256-pixel x 128-channel tiles; the K-steps of ALL their tiles form one stream
//     through a RING OF THREE 48 KiB LDS stages (256 pixel rows + 128 filter rows of 128 bytes, `buffer_load ... lds`, XOR-swizzled on the
//     source side): the loads of step t + 2 are issued while step t is multiplied, across tile boundaries, and every wait is a counted
//     `s_waitcnt vmcnt(N)` -- the queue is never drained (all memory operations are issued unconditionally; rows past the tensor's end and
//     steps past the workgroup's last tile read zeros / store nothing through the range check of per-tile descriptors)
//   * a wave owns 128 pixels x 32 channels (16 accumulator fragments): 20 fragment reads per 32 MFMAs, a whole 128-pixel statistics block per
//     wave (BatchNorm partial sums by DPP row sums, no cross-wave reduction), 16-byte stores after `v_permlane16_swap`
//   * co-tiles of one pixel tile are neighbours in the tile list: they run at the same time on one XCD and share the pixel rows in its L2
#include "common.h"
#include "conv_params.h"

#include <type_traits>

namespace {
typedef __attribute__((ext_vector_type(4))) unsigned g1_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned g1_u32x2;
typedef __attribute__((ext_vector_type(4))) float g1_f32x4;
template <int N> __device__ __forceinline__ void g1_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void g1_wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int OFF> __device__ __forceinline__ uint4 g1_lds_read16(unsigned byte_addr) {
    g1_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
template <int I, int N, typename F> __device__ __forceinline__ void g1_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); g1_static_for<I + 1, N>(f); }
}
__device__ __forceinline__ int g1_xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, xcd = b & 7, slot = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
constexpr unsigned G1_OOB = 0x80000000u;
#ifndef G1_STAGGER
#define G1_STAGGER 1         /* the two waves of a SIMD issue their LDS-DMA pieces at different points of a step (A/B builds: tools/build_variant.py) */
#endif
// 256 pixels x 256 channels per tile: (256 + 256) rows of 128 bytes per K-step = 64 KiB through the CU's load path (64 bytes / clock) per 2048 matrix-pipe
// cycles -- half the path's capacity.  128 x 128 tiles (the implicit GEMM) need ALL of it, 256 x 128 three quarters: tools/l2_probe.hip.
constexpr int G1_BM = 256, G1_BN = 256, G1_A_STAGE = G1_BM * 128, G1_B_STAGE = G1_BN * 128, G1_NA = 3, G1_NB = 2;
constexpr int G1_PA = G1_BM / 8 / 8, G1_PB = G1_BN / 8 / 8;         // LDS-DMA pieces (8 rows) per wave and K-step: pixels / filter rows

struct G1Params {
    const char* src; const char* wgt; char* dst; float* stat;
    long long M; int K; int Cd; int n_co; int n_tiles; int n_workers; int n_mblocks;
    int exp;   // timing experiments, only in builds with -DFB_C1G_EXPERIMENTS (WRONG results): FB_C1G_EXP & 1 = pixel rows from the first 512 rows (L2), & 2 = stores into the first 256 rows, & 4 = no MFMAs, & 8 = no fragment reads either, & 16 = no LDS-DMA (stale stages are multiplied)
};
#ifdef FB_C1G_EXPERIMENTS
#define G1_EXP(p) ((p).exp)
#else
#define G1_EXP(p) 0
#endif
}  // namespace

template <bool STAT>
__global__ __launch_bounds__(512) void conv1x1_gemm_kernel(const G1Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int FI = 4, FJ = 8;                                   // a wave: 64 channels x 128 pixels (32 accumulator fragments)
    constexpr int NST = 2 * FJ + (STAT ? 2 * FI : 0);               // stores of one tile's epilogue per wave
    __shared__ __attribute__((aligned(16))) char lds[G1_NA * G1_A_STAGE + G1_NB * G1_B_STAGE];   // 160 KiB: pixel ring (3 stages), filter ring (2)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int worker = __builtin_amdgcn_readfirstlane(g1_xcd_remap(blockIdx.x, gridDim.x));
    const int wp = wave >> 2, wc = wave & 3;                        // pixel half / channel quarter of the tile
    const int KS = p.K >> 6;                                        // K-steps per tile
    const int rowA_b = p.K * 2, row_b = p.Cd * 2;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    // LDS-DMA piece q = 4 wave + i of a ring stage: rows 8 q .. 8 q + 7; lane -> row (lane >> 3), logical chunk (lane & 7) ^ (row & 7)
    const unsigned dma_lane = (unsigned)((lane >> 3) * rowA_b + (((lane & 7) ^ (lane >> 3)) * 16));
    // fragment reads: row (16 j + col) of the wave's pixel half / (16 i + col) of its channel quarter, logical chunk g + 4 h
    unsigned pa[2], pb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        pa[h] = lds0 + (wp * 128 + col) * 128 + (((g + 4 * h) ^ (col & 7)) * 16);
        pb[h] = lds0 + G1_NA * G1_A_STAGE + (wc * 64 + col) * 128 + (((g + 4 * h) ^ (col & 7)) * 16);
    }
    // stores: after the lane-row exchange of a fragment pair and the col / col ^ 8 exchange (conv1x1_k32.hip) a store instruction writes the wave's 64
    // channels = one whole 128-byte line of 8 pixels: pixel wp * 128 + 16 j + (col & 7) [+ 8], channels wc * 64 + (col >> 3) * 32 + {0, 16, 8, 24}[g] .. + 7
    const unsigned voffS = (unsigned)((wp * 128 + (col & 7)) * row_b + (wc * 64 + (col >> 3) * 32 + (g & 1) * 16 + (g >> 1) * 8) * 2);
    const unsigned voffT = col == 0 ? (unsigned)((wc * 64 + g * 4) * 4) : G1_OOB;

    const int n_my = worker < p.n_tiles ? (p.n_tiles - worker + p.n_workers - 1) / p.n_workers : 0;      // tiles of this workgroup
    if (n_my == 0) return;

    // LDS-DMA rounds of step (tile index ti of this workgroup, K-step ks); ti >= n_my: empty descriptors (zeros into a stage nobody multiplies)
    auto issue_a = [&](const int ti, const int ks, const int stage) {
        const int tile = worker + ti * p.n_workers;
        const bool live = ti < n_my;
        const int mt = live ? tile / p.n_co : 0;
        const long long m0 = (long long)mt * G1_BM;
        long long rows = p.M - m0;
        rows = !live || rows < 0 ? 0 : (rows > G1_BM ? G1_BM : rows);
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + ((G1_EXP(p) & 1) ? (m0 & 511) : m0) * rowA_b), 0, (int)(rows * rowA_b), 0x00020000);
        char* base = lds + stage * G1_A_STAGE;
#pragma unroll
        for (int i = 0; i < G1_PA; ++i) {
            const int q = wave * G1_PA + i;                         // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + q * 1024), 16, dma_lane + (unsigned)(q * 8 * rowA_b), ks * 128, 0, 0);
        }
    };
    auto issue_b = [&](const int ti, const int ks, const int stage) {
        const int tile = worker + ti * p.n_workers;
        const bool live = ti < n_my;
        const int co = live ? tile % p.n_co : 0;
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.wgt + (long long)co * G1_BN * rowA_b), 0, live ? G1_BN * rowA_b : 0, 0x00020000);
        char* base = lds + G1_NA * G1_A_STAGE + stage * G1_B_STAGE;
#pragma unroll
        for (int i = 0; i < G1_PB; ++i) {
            const int q = wave * G1_PB + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(base + q * 1024), 16, dma_lane + (unsigned)(q * 8 * rowA_b), ks * 128, 0, 0);
        }
    };

    f32x4_t acc[FI][FJ];
    // Queue order (per wave, all counts follow from it): step t issues [filter round t + 1] [pixel round t + 2]; the top of step t needs pixel round t (issued
    // in step t - 2) and filter round t (step t - 1) and leaves pixel round t + 1 (step t - 1, BEHIND filter round t) in flight: vmcnt(G1_PA).
    // Prologue: [pixels 0] [filter 0] [pixels 1]  (KS >= 8: all of the first tile)
    issue_a(0, 0, 0);
    issue_b(0, 0, 0);
    issue_a(0, 1, 1);
    int b_ti = 0, b_ks = 1, a_ti = 0, a_ks = 2;          // the next rounds to issue
    int sa = 0, sb = 0;                                  // ring stages of the step that is multiplied
    int after_epilogue = 0;                              // the first step of a tile: the previous tile's stores sit behind pixel round 1 in the queue (one step: the
                                                         // second step's filter round was issued behind them, so its wait retires them anyway)
    for (int ti = 0; ti < n_my; ++ti) {
        for (int ks = 0; ks < KS; ++ks) {
            if (after_epilogue > 0) { g1_wait_vmcnt<G1_PA + NST>(); --after_epilogue; } else g1_wait_vmcnt<G1_PA>();
            __builtin_amdgcn_s_barrier();                // ... everybody's share has landed; everybody has left the stages the next rounds go into
            // An LDS-DMA instruction holds its wave at issue while the CU's load path works the queue off (64 of them per step: ~2000 cycles with all eight
            // waves issuing together and the matrix pipe idle -- as long as the step's MFMAs).  So the two waves of a SIMD take turns: waves 0-3 (one per
            // SIMD) issue their pieces here, waves 4-7 between the two halves of their MFMAs -- while one is held, the other one multiplies.
            const bool early = !G1_STAGGER || wave < 4;
            if (early) {
                issue_b((G1_EXP(p) & 16) ? n_my : b_ti, b_ks, sb ^ 1);
                issue_a((G1_EXP(p) & 16) ? n_my : a_ti, a_ks, sa == 0 ? 2 : sa - 1);
            }
#ifdef FB_C1G_EXPERIMENTS
            if (G1_EXP(p) & 8) { if (ks == 0) for (int i = 0; i < FI; ++i) for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; sa = sa == 2 ? 0 : sa + 1; sb ^= 1; if (!early) { issue_b(b_ti, b_ks, sb); issue_a(a_ti, a_ks, sa == 2 ? 0 : sa + 1); } if (++b_ks == KS) { b_ks = 0; ++b_ti; } if (++a_ks == KS) { a_ks = 0; ++a_ti; } continue; }
#endif
            const unsigned soa = sa * G1_A_STAGE, sob = sb * G1_B_STAGE;
            uint4 wb[2][FI], px[2][FJ];
            g1_static_for<0, 2>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                g1_static_for<0, FI>([&](auto ic) { constexpr int i = decltype(ic)::value; wb[h][i] = g1_lds_read16<i * 2048>(pb[h] + sob); });
                g1_static_for<0, FJ>([&](auto jc) { constexpr int j = decltype(jc)::value; px[h][j] = g1_lds_read16<j * 2048>(pa[h] + soa); });
            });
            g1_wait_lgkmcnt<FI + FJ>();
#ifdef FB_C1G_EXPERIMENTS
            if (G1_EXP(p) & 4) {
                g1_wait_lgkmcnt<0>();
                if (ks == 0) for (int i = 0; i < FI; ++i) for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                unsigned x = 0;
                for (int h = 0; h < 2; ++h) { for (int i = 0; i < FI; ++i) x ^= wb[h][i].x ^ wb[h][i].w; for (int j = 0; j < FJ; ++j) x ^= px[h][j].y ^ px[h][j].z; }
                acc[0][0][0] += __uint_as_float(x);
                if (!early) { issue_b(b_ti, b_ks, sb ^ 1); issue_a(a_ti, a_ks, sa == 0 ? 2 : sa - 1); }
                if (++b_ks == KS) { b_ks = 0; ++b_ti; }
                if (++a_ks == KS) { a_ks = 0; ++a_ti; }
                sa = sa == 2 ? 0 : sa + 1; sb ^= 1;
                continue;
            }
#endif
            if (ks == 0) {
#pragma unroll
                for (int i = 0; i < FI; ++i)
#pragma unroll
                    for (int j = 0; j < FJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wb[0][i]), __builtin_bit_cast(bf16x8_t, px[0][j]), (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < FI; ++i)
#pragma unroll
                    for (int j = 0; j < FJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wb[0][i]), __builtin_bit_cast(bf16x8_t, px[0][j]), acc[i][j], 0, 0, 0);
            }
            if (!early) {
                __builtin_amdgcn_sched_barrier(0);
                issue_b((G1_EXP(p) & 16) ? n_my : b_ti, b_ks, sb ^ 1);
                issue_a((G1_EXP(p) & 16) ? n_my : a_ti, a_ks, sa == 0 ? 2 : sa - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (++b_ks == KS) { b_ks = 0; ++b_ti; }
            if (++a_ks == KS) { a_ks = 0; ++a_ti; }
            g1_wait_lgkmcnt<0>();
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wb[1][i]), __builtin_bit_cast(bf16x8_t, px[1][j]), acc[i][j], 0, 0, 0);
            sa = sa == 2 ? 0 : sa + 1;
            sb ^= 1;
        }
        // ---- epilogue of the tile: bf16 outputs (whole 128-byte lines per store instruction), BatchNorm partial sums of the wave's 128-pixel block ----
        const int tile = worker + ti * p.n_workers;
        const int mt = tile / p.n_co, co = tile - mt * p.n_co;
        const long long m0 = (long long)mt * G1_BM;
        long long rows = p.M - m0;
        rows = rows > G1_BM ? G1_BM : rows;
        const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dst + ((G1_EXP(p) & 2) ? (m0 & 255) : m0) * row_b + co * G1_BN * 2), 0, (int)(rows * row_b), 0x00020000);
        float ssum[FI][4], ssq[FI][4];
        const bool upper = col >= 8;
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            unsigned q[FI][2];
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                q[i][0] = pack_bf16x2(v[0], v[1]); q[i][1] = pack_bf16x2(v[2], v[3]);
                if constexpr (STAT) {
                    if (j == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ssum[i][r] = v[r]; ssq[i][r] = v[r] * v[r]; }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ssum[i][r] += v[r]; ssq[i][r] = fmaf(v[r], v[r], ssq[i][r]); }
                    }
                }
            }
            // fragment pairs (0, 1) / (2, 3): 8 consecutive channels per lane ({0, 16, 8, 24}[g] of the pair's 32); lanes col and col ^ 8 then trade one
            // piece each, so that s1 / s2 are the two 32-channel halves' pieces of pixel (col & 7) / (col & 7) + 8 and col >> 3 picks the half
            g1_u32x4 a0, a2;
            {
                const g1_u32x2 lo = __builtin_amdgcn_permlane16_swap(q[0][0], q[1][0], false, false);
                const g1_u32x2 hi = __builtin_amdgcn_permlane16_swap(q[0][1], q[1][1], false, false);
                a0 = (g1_u32x4){lo[0], hi[0], lo[1], hi[1]};
                const g1_u32x2 lo2 = __builtin_amdgcn_permlane16_swap(q[2][0], q[3][0], false, false);
                const g1_u32x2 hi2 = __builtin_amdgcn_permlane16_swap(q[2][1], q[3][1], false, false);
                a2 = (g1_u32x4){lo2[0], hi2[0], lo2[1], hi2[1]};
            }
            g1_u32x4 give, got, s1, s2;
#pragma unroll
            for (int e = 0; e < 4; ++e) give[e] = upper ? a0[e] : a2[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) got[e] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)give[e], 0x128, 0xf, 0xf, false);   // row_ror:8
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[e] = upper ? got[e] : a0[e]; s2[e] = upper ? a2[e] : got[e]; }
            __builtin_amdgcn_raw_buffer_store_b128(s1, rsD, voffS + (unsigned)(j * 16 * row_b), 0, 0);
            store_b128_guard(s1);
            __builtin_amdgcn_raw_buffer_store_b128(s2, rsD, voffS + (unsigned)((j * 16 + 8) * row_b), 0, 0);
            store_b128_guard(s2);
        }
        if constexpr (STAT) {
            const long long blk = (m0 >> 7) + wp;
            const bool ok = blk < p.n_mblocks;
            const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc((void*)(p.stat + (ok ? blk : 0) * p.Cd + co * G1_BN), 0,
                                                                                ok ? (int)((p.n_mblocks + 1LL) * p.Cd * 4) : 0, 0x00020000);
            const int plane = p.n_mblocks * p.Cd * 4;
#pragma unroll
            for (int i = 0; i < FI; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { ssum[i][r] = row16_sum(ssum[i][r]); ssq[i][r] = row16_sum(ssq[i][r]); }
                { const g1_u32x4 sv = __builtin_bit_cast(g1_u32x4, (g1_f32x4){ssum[i][0], ssum[i][1], ssum[i][2], ssum[i][3]}); __builtin_amdgcn_raw_buffer_store_b128(sv, rsT, voffT + i * 64, 0, 0); store_b128_guard(sv); }
                { const g1_u32x4 sv = __builtin_bit_cast(g1_u32x4, (g1_f32x4){ssq[i][0], ssq[i][1], ssq[i][2], ssq[i][3]}); __builtin_amdgcn_raw_buffer_store_b128(sv, rsT, voffT + i * 64, plane, 0); store_b128_guard(sv); }
            }
        }
        after_epilogue = 1;
    }
    g1_wait_vmcnt<0>();                                  // (the rounds past the end: nothing may land in LDS after the workgroup has left)
#endif
}

// returns 1 if the kernel handled the call: bf16 1x1 convolution (forward or input gradient), K a multiple of 64 and >= 512, output channels a multiple of
// 128, one shared weight set, no addend, optional BatchNorm partial sums (forward).  FB_C1G=0: the implicit GEMM takes these calls.
int fb_try_conv1x1_gemm(const fb_conv_args* a, hipStream_t st) {
    // OPT-IN (FB_C1G=1: forward calls, FB_C1G=2: input gradients too; read per call: the tests compare the two kernels inside one process).  Built, bit-identical to
    // the implicit GEMM, and measured WITHOUT effect where it counts: ResNet-152 @224, 2048 images per step, same box: 5775 / 5809 images/s with it,
    // 5789 / 5780 without, 5790 with the input gradients too.
    const char* sw = getenv("FB_C1G");
    if (sw == nullptr || atoi(sw) == 0) return 0;
    if (a->R != 1 || a->S != 1 || a->stride != 1 || a->pad != 0 || a->dtype != FB_BF16) return 0;
    if (a->Hs != a->Hd || a->Ws != a->Wd) return 0;
    if (a->Cs < 512 || a->Cs % 64 != 0 || a->Cs > 4096 || a->Cd % G1_BN != 0) return 0;
    if (a->wset_stride != 0 && a->imgs_per_wset > 0 && a->imgs_per_wset < a->n_img) return 0;
    if (a->addend || a->addend_mask || a->bst_x) return 0;
    if (a->mode == 1 && a->stat_partial) return 0;
    // Measured against the implicit GEMM (1024 images, same box, us): forward 1024 -> 256 @14x14 147 / 158, 512 -> 2048 @7x7 149 / 177, 512 -> 128 @28x28
    // 210 / 219, 2048 -> 512 @7x7 140 / 138; input gradients 154 / 154, 137 / 132, 211 / 215, 129 / 133: the ring removes the waits but the 48 LDS-DMA
    // instructions a K-step needs cost a wave ~100 cycles each to ISSUE -- as much as its 32 MFMAs -- so the matrix pipe is no busier than before.
    // (alone on the device; inside the step the difference disappears, see above)
    if (a->mode == 1 && !(sw != nullptr && atoi(sw) == 2)) return 0;
    const long long M = (long long)a->n_img * a->Hd * a->Wd;
    if ((M + G1_BM - 1) / G1_BM * (a->Cd / G1_BN) >= (1LL << 31) || (M / 128 + 2) * a->Cd * 8 >= (1LL << 31)) return 0;
    G1Params p;
    p.src = (const char*)a->src; p.wgt = (const char*)a->wgt; p.dst = (char*)a->dst; p.stat = a->stat_partial;
    p.M = M; p.K = a->Cs; p.Cd = a->Cd;
    p.n_co = a->Cd / G1_BN;
    p.n_tiles = (int)((M + G1_BM - 1) / G1_BM) * p.n_co;
    p.n_mblocks = (int)((M + 127) / 128);
    p.exp = 0;
#ifdef FB_C1G_EXPERIMENTS
    p.exp = getenv("FB_C1G_EXP") ? atoi(getenv("FB_C1G_EXP")) : 0;
#endif
    const int n_cu = fb_persistent_cus();
    p.n_workers = p.n_tiles < n_cu ? p.n_tiles : n_cu;
    if (a->stat_partial) hipLaunchKernelGGL((conv1x1_gemm_kernel<true>), dim3(p.n_workers), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv1x1_gemm_kernel<false>), dim3(p.n_workers), dim3(512), 0, st, p);
    return 1;
}
