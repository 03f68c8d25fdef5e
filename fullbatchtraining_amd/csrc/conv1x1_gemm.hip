// 1x1 convolutions with K >= 512 input channels and a multiple of 256 output channels (the channel-reducing convolutions of the Bottleneck blocks and
// their input gradients, reference resnets.py:289-291: 1024 -> 256 @14x14, 2048 -> 512 @7x7 ...), bf16, forward and input gradient.
//
// These went to the implicit GEMM (conv_igemm_glds.hip: 128 x 128 tiles, ONE 32 KiB stage, three workgroups per CU: 1024 -> 256 @14x14, 1024 images,
// 155-160 us = 3.2 TB/s, matrix pipe busy a third of the time).  Two things hold that form down (round 5, tools/l2_probe.hip and the timing experiments
// of this file): a 128 x 128 tile stages 32 KiB per 512 matrix-pipe cycles -- ALL of the 64 bytes / clock a CU's load path moves -- and its phases run
// one after the other (an LDS-DMA instruction holds its wave at issue while the load path works the queue off, and all waves of a workgroup issue
// together).  Here:
//   * 256-pixel x 256-channel tiles (8 waves, one persistent workgroup per CU; a wave: 128 pixels x 64 channels = 32 accumulator fragments, 24
//     fragment reads per 64 MFMAs): 64 KiB through the load path per 2048 matrix-pipe cycles, half its capacity
//   * the K-steps of ALL tiles of a workgroup form one stream through two LDS rings -- pixel rows: three 32 KiB stages, filter rows: two (160 KiB,
//     `buffer_load ... lds`, XOR-swizzled on the source side); every wait is a counted `s_waitcnt vmcnt(N)`, nothing drains the queue (all memory
//     operations are issued unconditionally; rows past the tensor's end and steps past the last tile read zeros / store nothing: per-tile descriptors)
//   * PING-PONG: the two waves of every SIMD run half a step apart and split the loading -- see the comment at the loops
//   * whole 128-byte lines per store instruction (lane-row exchange + the col / col ^ 8 exchange of conv1x1_k32.hip): half-line stores of two waves
//     reach memory as 1.25x the bytes (round 5, tools/pmc_hbm_case.sh); a whole 128-pixel statistics block per wave (DPP row sums, no cross-wave reduction)
//   * co-tiles of one pixel tile are neighbours in the tile list: they run at the same time on one XCD and share the pixel rows in its L2
// A step's timeline: tools/g1_trace.hip.
#include "common.h"
#include "conv_params.h"

#include <type_traits>
#ifdef FB_C1G_TRACE
extern long long* g_g1_trace;        // tools/g1_trace.hip
#endif

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
#ifndef G1_DMA_FIRST
#define G1_DMA_FIRST 0       /* 1: a step's LDS-DMA pieces are issued before its fragment reads (A/B builds) */
#endif
#ifndef G1_IQ
#define G1_IQ 2              /* 16-channel fragment rows (8 MFMAs each) of a step multiplied BEFORE its middle barrier (A/B builds: tools/build_variant.py) */
#endif
// 256 pixels x 256 channels per tile: (256 + 256) rows of 128 bytes per K-step = 64 KiB through the CU's load path (64 bytes / clock) per 2048 matrix-pipe
// cycles -- half the path's capacity.  128 x 128 tiles (the implicit GEMM) need ALL of it, 256 x 128 three quarters: tools/l2_probe.hip.
constexpr int G1_BM = 256, G1_BN = 256, G1_A_STAGE = G1_BM * 128, G1_B_STAGE = G1_BN * 128, G1_NA = 3, G1_NB = 2;
constexpr int G1_PA = G1_BM / 8 / 4, G1_PB = G1_BN / 8 / 4;         // LDS-DMA pieces (8 rows) per wave of the loading group and K-step: pixels / filter rows

struct G1Params {
    const char* src; const char* wgt; char* dst; float* stat;
    const char* addend; const unsigned char* addend_mask;     // ADDM 1 / 2: same-shape addend of an input gradient; ... through its ReLU bitmask (1 byte per 8 channels)
    long long M; int K; int Cd; int n_co; int n_tiles; int n_workers; int n_mblocks;
#ifdef FB_C1G_TRACE
    long long* trace;
#endif
    int exp;   // timing experiments, only in builds with -DFB_C1G_EXPERIMENTS (WRONG results): FB_C1G_EXP & 1 = pixel rows from the first 512 rows (L2), & 2 = stores into the first 256 rows
};
#ifdef FB_C1G_TRACE
// tools/g1_trace.hip: waves 0 and 4 of workgroup 0 stamp the shader clock (s_memtime) at 8 points of steps G1_T0 .. G1_T0 + 15
#define G1_T0 20
#define G1_STAMP(grp, k) do { if (worker == 0 && gw == 0 && tstep >= G1_T0 && tstep < G1_T0 + 16) { const long long t_ = __builtin_amdgcn_s_memtime(); \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) p.trace[(((grp) * 16 + (tstep - G1_T0)) * 8) + (k)] = t_; } } while (0)
#else
#define G1_STAMP(grp, k) do { } while (0)
#endif
#ifdef FB_C1G_EXPERIMENTS
#define G1_EXP(p) ((p).exp)
#else
#define G1_EXP(p) 0
#endif
}  // namespace

// ADDM: 0 no addend, 1 same-shape addend, 2 the addend where its ReLU bit is set (fb_conv_args.addend_mask) -- input gradients of the convolution behind a residual branch
template <bool STAT, int ADDM = 0>
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

    // LDS-DMA rounds of step (tile index ti of this workgroup, K-step ks); ti >= n_my: empty descriptors (zeros into a stage nobody multiplies).
    // The four waves of a group issue a whole round: 8 pieces each.
    const int gw = wave & 3;
    // (the descriptor of a round depends on its tile only: rebuilt when the tile index moves on, the K-step rides in the scalar offset)
    auto desc_a = [&](const int ti) {
        const int tile = worker + ti * p.n_workers;
        const bool live = ti < n_my;
        const int mt = live ? tile / p.n_co : 0;
        const long long m0 = (long long)mt * G1_BM;
        long long rows = p.M - m0;
        rows = !live || rows < 0 ? 0 : (rows > G1_BM ? G1_BM : rows);
        return __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + ((G1_EXP(p) & 1) ? (m0 & 511) : m0) * rowA_b), 0, (int)(rows * rowA_b), 0x00020000);
    };
    auto desc_b = [&](const int ti) {
        const int tile = worker + ti * p.n_workers;
        const bool live = ti < n_my;
        const int co = live ? tile % p.n_co : 0;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(p.wgt + (long long)co * G1_BN * rowA_b), 0, live ? G1_BN * rowA_b : 0, 0x00020000);
    };
    auto issue_a = [&](const __amdgpu_buffer_rsrc_t rsA, const int ks, const int stage) {
        char* base = lds + stage * G1_A_STAGE;
#pragma unroll
        for (int i = 0; i < G1_PA; ++i) {
            const int q = gw * G1_PA + i;                           // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + q * 1024), 16, dma_lane + (unsigned)(q * 8 * rowA_b), ks * 128, 0, 0);
        }
    };
    auto issue_b = [&](const __amdgpu_buffer_rsrc_t rsB, const int ks, const int stage) {
        char* base = lds + G1_NA * G1_A_STAGE + stage * G1_B_STAGE;
#pragma unroll
        for (int i = 0; i < G1_PB; ++i) {
            const int q = gw * G1_PB + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(base + q * 1024), 16, dma_lane + (unsigned)(q * 8 * rowA_b), ks * 128, 0, 0);
        }
    };

    f32x4_t acc[FI][FJ];
    uint4 wb[2][FI], px[2][FJ];
    // A step's work of a wave: all fragment reads (24), then 64 MFMAs in two PARTS of 8 G1_IQ / 64 - 8 G1_IQ around the middle barrier: the group that is busy
    // with the memory side (reads, LDS-DMA issue) multiplies little, the other one much -- both parts of an interval end together.
    auto frag_reads = [&](const int sa, const int sb) {
        const unsigned soa = sa * G1_A_STAGE, sob = sb * G1_B_STAGE;
        g1_static_for<0, 2>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            g1_static_for<0, FI>([&](auto ic) { constexpr int i = decltype(ic)::value; wb[h][i] = g1_lds_read16<i * 2048>(pb[h] + sob); });
            g1_static_for<0, FJ>([&](auto jc) { constexpr int j = decltype(jc)::value; px[h][j] = g1_lds_read16<j * 2048>(pa[h] + soa); });
        });
    };
    auto mma_rows = [&](auto hc, auto i0c, auto i1c, auto firstc) {
        constexpr int h = decltype(hc)::value, i0 = decltype(i0c)::value, i1 = decltype(i1c)::value;
#pragma unroll
        for (int i = i0; i < i1; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                const f32x4_t c = decltype(firstc)::value ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : acc[i][j];
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wb[h][i]), __builtin_bit_cast(bf16x8_t, px[h][j]), c, 0, 0, 0);
            }
    };
    using c0 = std::integral_constant<int, 0>; using c1 = std::integral_constant<int, 1>; using cq = std::integral_constant<int, G1_IQ>; using cf = std::integral_constant<int, FI>;
    auto part1 = [&](const bool first) {
        g1_wait_lgkmcnt<FI + FJ>();                      // the fragments of the first 32 channels
        if (first) mma_rows(c0{}, c0{}, cq{}, std::true_type{}); else mma_rows(c0{}, c0{}, cq{}, std::false_type{});
        g1_wait_lgkmcnt<0>();                            // (every fragment of the step is in registers: the stages may be overwritten behind the next barrier)
    };
    auto part2 = [&](const bool first) {
        if (first) mma_rows(c0{}, cq{}, cf{}, std::true_type{}); else mma_rows(c0{}, cq{}, cf{}, std::false_type{});
        mma_rows(c1{}, c0{}, cf{}, std::false_type{});
    };
    // epilogue of a tile: bf16 outputs (whole 128-byte lines per store instruction), BatchNorm partial sums of the wave's 128-pixel block
    auto epilogue = [&](const int ti) {
        const int tile = worker + ti * p.n_workers;
        const int mt = tile / p.n_co, co = tile - mt * p.n_co;
        const long long m0 = (long long)mt * G1_BM;
        long long rows = p.M - m0;
        rows = rows > G1_BM ? G1_BM : rows;
        const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dst + ((G1_EXP(p) & 2) ? (m0 & 255) : m0) * row_b + co * G1_BN * 2), 0, (int)(rows * row_b), 0x00020000);
        float ssum[FI][4], ssq[FI][4];
        const bool upper = col >= 8;
        // ADDM: the addend in the ACCUMULATORS' layout (a lane: channels 16 i + 4 g .. + 3 of pixel 16 j + col: 8 bytes per fragment), ordinary loads -- hipcc
        // waits for them with vmcnt(0), which also lands this wave's ring rounds: the ones its next step's wait would have waited for anyway (E: filter round
        // t + 1, L: pixel round t + 2), so the counted waits behind the epilogue only become conservative.  The mask: 8 bytes per pixel (this wave's 64 channels).
        const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc((void*)(ADDM ? p.addend + m0 * row_b + co * G1_BN * 2 : p.dst), 0, ADDM ? (int)(rows * row_b) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void*)(ADDM == 2 ? (const char*)p.addend_mask + ((m0 * row_b + co * G1_BN * 2) >> 4) : p.dst), 0,
                                                                              ADDM == 2 ? (int)((rows * row_b) >> 4) : 0, 0x00020000);
        const unsigned voffE = (unsigned)((wp * 128 + col) * row_b + (wc * 64 + g * 4) * 2), voffM = (unsigned)(((wp * 128 + col) * row_b + wc * 64 * 2) >> 4);
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            unsigned q[FI][2];
            g1_u32x2 am = {0xffffffffu, 0xffffffffu};
            if constexpr (ADDM == 2) am = __builtin_amdgcn_raw_buffer_load_b64(rsM, voffM + (unsigned)((j * 16 * row_b) >> 4), 0, 0);
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if constexpr (ADDM != 0) {
                    g1_u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rsE, voffE + (unsigned)(j * 16 * row_b) + i * 32, 0, 0);
                    if constexpr (ADDM == 2) {       // mask byte 2 i + (g >> 1) of the pixel's eight: this lane's channels are its low (g even) or high nibble
                        const unsigned bits = am[i >> 1] >> ((((2 * i + (g >> 1)) & 3) * 8) + (g & 1) * 4);
                        const unsigned m0_ = (unsigned)__builtin_amdgcn_sbfe((int)bits, 0, 1), m1_ = (unsigned)__builtin_amdgcn_sbfe((int)bits, 1, 1);
                        const unsigned m2_ = (unsigned)__builtin_amdgcn_sbfe((int)bits, 2, 1), m3_ = (unsigned)__builtin_amdgcn_sbfe((int)bits, 3, 1);
                        a[0] &= (m0_ & 0x0000ffffu) | (m1_ & 0xffff0000u);
                        a[1] &= (m2_ & 0x0000ffffu) | (m3_ & 0xffff0000u);
                    }
                    v[0] += __uint_as_float(a[0] << 16); v[1] += __uint_as_float(a[0] & 0xffff0000u);
                    v[2] += __uint_as_float(a[1] << 16); v[3] += __uint_as_float(a[1] & 0xffff0000u);
                }
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
    };

    // PING-PONG.  What the timing experiments said about the lock-step form (all eight waves: barrier, 64 LDS-DMA instructions, 192 fragment reads,
    // 512 MFMAs): the phases add up -- an LDS-DMA instruction holds its wave at issue while the CU's load path works the queue off (~2000 cycles per
    // step with nothing else running), the fragment reads of eight waves arrive together, and the matrix pipe waits for both.  Here the two waves of
    // every SIMD run HALF A STEP APART: waves 0-3 (group E, one per SIMD) and waves 4-7 (group L) alternate between [LDS-DMA issue, fragment reads,
    // first 32 MFMAs] and [second 32 MFMAs] -- in every barrier interval one group feeds the matrix pipe from registers while the other one is busy
    // with the memory side.  Barriers b = 0, 1, 2 ...: E's step t runs from barrier 2t (middle: 2t + 1), L's from 2t + 1 (middle: 2t + 2).
    //   E loads the FILTER rounds (two stages): round t + 1 behind barrier 2t, awaited before barrier 2t + 2 (its own next top)
    //   L loads the PIXEL rounds (three stages): round t + 2 behind barrier 2t + 1, round t + 1 awaited before barrier 2t + 2 (E reads it next)
    // Stage reuse: a stage's last fragment read is complete (lgkmcnt(0)) before the reader's middle barrier, and the next round into it is issued at least one
    // barrier later.  L passes one extra barrier at the start (that is the offset), E one at the end.
    int tstep = 0;                                       // (tools/g1_trace.hip stamps steps by this count)
    (void)tstep;
    if (wave < 4) {
        __amdgpu_buffer_rsrc_t rsB = desc_b(0);
        issue_b(rsB, 0, 0);
        int b_ti = 0, b_ks = 1, sa = 0, sb = 0;
        bool after_epilogue = false;
        for (int ti = 0; ti < n_my; ++ti) {
            for (int ks = 0; ks < KS; ++ks) {
                G1_STAMP(0, 0);
                if (after_epilogue) { g1_wait_vmcnt<NST>(); after_epilogue = false; } else g1_wait_vmcnt<0>();     // filter round t (the newest loads of this wave)
                G1_STAMP(0, 1);
                __builtin_amdgcn_s_barrier();            // 2t
                G1_STAMP(0, 2);
                if (G1_DMA_FIRST) issue_b(rsB, b_ks, sb ^ 1);
                frag_reads(sa, sb);
                if (!G1_DMA_FIRST) issue_b(rsB, b_ks, sb ^ 1);      // (behind the reads: while the wave is held at issue its fragments arrive)
                if (++b_ks == KS) { b_ks = 0; ++b_ti; rsB = desc_b(b_ti); }
                G1_STAMP(0, 3);
                part1(ks == 0);
                G1_STAMP(0, 4);
                __builtin_amdgcn_s_barrier();            // 2t + 1
                G1_STAMP(0, 5);
                part2(ks == 0);
                G1_STAMP(0, 6);
                ++tstep;
                sa = sa == 2 ? 0 : sa + 1;
                sb ^= 1;
            }
            epilogue(ti);
            after_epilogue = true;
        }
        __builtin_amdgcn_s_barrier();                    // 2 n_steps: L's last middle barrier
    } else {
        __amdgpu_buffer_rsrc_t rsA = desc_a(0);
        issue_a(rsA, 0, 0);
        issue_a(rsA, 1, 1);
        // the next round to issue is round 2 = (tile 2 / KS, K-step 2 % KS): with K = 128 (KS = 2) that is already the SECOND tile's first K-step (round 5 started
        // at (0, 2) for every K: a K-step past the row's end, and a tile index that never moved on -- wrong pixel rows for every tile behind a workgroup's first)
        int a_ti = 2 / KS, a_ks = 2 % KS, sa = 0, sb = 0;
        if (a_ti != 0) rsA = desc_a(a_ti);
        bool after_epilogue = false;
        g1_wait_vmcnt<G1_PA>();                          // pixel round 0
        __builtin_amdgcn_s_barrier();                    // 0
        for (int ti = 0; ti < n_my; ++ti) {
            for (int ks = 0; ks < KS; ++ks) {
                G1_STAMP(1, 0);
                __builtin_amdgcn_s_barrier();            // 2t + 1
                G1_STAMP(1, 2);
                if (G1_DMA_FIRST) issue_a(rsA, a_ks, sa == 0 ? 2 : sa - 1);
                frag_reads(sa, sb);
                if (!G1_DMA_FIRST) issue_a(rsA, a_ks, sa == 0 ? 2 : sa - 1);
                if (++a_ks == KS) { a_ks = 0; ++a_ti; rsA = desc_a(a_ti); }
                G1_STAMP(1, 3);
                part1(ks == 0);
                G1_STAMP(1, 4);
                // pixel round t + 1 (issued a step ago; behind it in the queue: round t + 2, and once per tile the previous tile's stores between them)
                if (after_epilogue) { g1_wait_vmcnt<G1_PA + NST>(); after_epilogue = false; } else g1_wait_vmcnt<G1_PA>();
                G1_STAMP(1, 1);
                __builtin_amdgcn_s_barrier();            // 2t + 2
                G1_STAMP(1, 5);
                part2(ks == 0);
                G1_STAMP(1, 6);
                ++tstep;
                sa = sa == 2 ? 0 : sa + 1;
                sb ^= 1;
            }
            epilogue(ti);
            after_epilogue = true;
        }
    }
    g1_wait_vmcnt<0>();                                  // (the rounds past the end: nothing may land in LDS after the workgroup has left)
#endif
}

// returns 1 if the kernel handled the call: bf16 1x1 convolution (forward or input gradient), K a multiple of 64 and >= 512, output channels a multiple of
// 256, one shared weight set, no addend, optional BatchNorm partial sums (forward).
// FB_C1G=0: never, 1: every forward call it can take, 2: input gradients too (read per call: the tests compare the two kernels inside one process; same bits -- both
// add the K-steps up in the same order).  Alone on the device it beats the implicit GEMM by 15-25 % (1024 / 2048 images, same box, us: forward 1024 -> 256
// @14x14 136 / 159 and 242-253 / 326-334, its input-gradient twin 141 / 173 and 256-262 / 304-307, 2048 -> 512 @7x7 122 / 146, 107 / 121), and rocprofv3 sees
// those 4.4 ms of kernel time go (ResNet-152 @224, one stream: 356.7 -> 351.8 ms of kernels per step) -- but with groups of 1024 images the STEP did not get
// shorter (5848-5890 images/s with it against 5878-5912 without; a workgroup takes its CU's whole LDS, so nothing overlaps its start or its tail).  With the
// 2048-image groups of the end of round 5 the forward calls pay (6255-6266 against 6214-6228), the input gradients do not (6210-6233): the default.
int fb_conv1x1_gemm_takes(const fb_conv_args* a) {
    const char* sw = getenv("FB_C1G");
    if (sw != nullptr && atoi(sw) == 0) return 0;
    if (a->R != 1 || a->S != 1 || a->stride != 1 || a->pad != 0 || a->dtype != FB_BF16) return 0;
    if (a->Hs != a->Hd || a->Ws != a->Wd) return 0;
    // K = 128 / 256 (the Bottleneck blocks' conv3: 128 -> 512, 256 -> 1024; K = 128: +0.45 % more): forward calls only -- alone on the device the streaming kernel is faster (244 against 320 us, 2048 images), inside
    // the ResNet-152 step this one is (6383-6385 against 6344-6347 images/s: whole-line stores, 822 MB written instead of 1028)
    // (K < 512 only where the output is four times the input -- the expanding conv3 of a Bottleneck block, whose bytes are its stores; the 2x shortcuts of a BasicBlock net
    // run beside a weight-gradient stream that this kernel's 160 KiB of LDS shut out: ResNet-18 headline 233.5-233.9 ms with it against 232.9-233.6)
    if (sw == nullptr && a->Cs < 512 && a->Cd < 4 * a->Cs) return 0;
    if (a->Cs < ((a->mode == 0 || a->addend) ? 128 : 512) || a->Cs % 64 != 0 || a->Cs > 4096 || a->Cd % G1_BN != 0) return 0;
    if (a->wset_stride != 0 && a->imgs_per_wset > 0 && a->imgs_per_wset < a->n_img) return 0;
    if (a->bst_x) return 0;
    if (a->addend && (a->mode != 1 || a->addend_mode != 1)) return 0;
    if (a->addend_mask && !a->addend) return 0;
    if (a->mode == 1 && a->stat_partial) return 0;
    // input gradients only with FB_C1G=2: without an addend (K >= 512) no step-level gain; with the same-shape addend of an identity block (plain or through its ReLU
    // bitmask, K >= 128) the epilogue's 8-byte loads in the accumulators' layout are waited for with vmcnt(0) and the kernel spills: ResNet-152 @224 6082-6092
    // images/s with it against 6421-6442 on the streaming kernel (round 5) -- built, bit-identical, off
    if (a->mode == 1 && !(sw != nullptr && atoi(sw) == 2 && fb_experimental())) return 0;      // (FB_C1G=2 acts only with FB_EXPERIMENTAL=1: the input-gradient forms lost their A/B)
    const long long M = (long long)a->n_img * a->Hd * a->Wd;
    if ((M + G1_BM - 1) / G1_BM * (a->Cd / G1_BN) >= (1LL << 31) || (M / 128 + 2) * a->Cd * 8 >= (1LL << 31)) return 0;
    const long long n_tiles = (M + G1_BM - 1) / G1_BM * (a->Cd / G1_BN);
    // DEFAULT: calls whose workgroups walk five tiles or more (a 2048-image ResNet-152 group: 1568 tiles; a 1024-image group's 3.06 tiles per workgroup
    // round up to 4 -- a quarter of the kernel's span is a tail on 16 CUs, and the step gains nothing there)
    if (sw == nullptr && n_tiles < 5 * fb_persistent_cus()) return 0;
    return 1;
}

int fb_try_conv1x1_gemm(const fb_conv_args* a, hipStream_t st) {
    if (!fb_conv1x1_gemm_takes(a)) return 0;
    const long long M = (long long)a->n_img * a->Hd * a->Wd;
    G1Params p;
    p.src = (const char*)a->src; p.wgt = (const char*)a->wgt; p.dst = (char*)a->dst; p.stat = a->stat_partial;
    p.addend = (const char*)a->addend; p.addend_mask = (const unsigned char*)a->addend_mask;
    p.M = M; p.K = a->Cs; p.Cd = a->Cd;
    p.n_co = a->Cd / G1_BN;
    p.n_tiles = (int)((M + G1_BM - 1) / G1_BM) * p.n_co;
    p.n_mblocks = (int)((M + 127) / 128);
    p.exp = 0;
#ifdef FB_C1G_TRACE
    p.trace = ::g_g1_trace;
#endif
#ifdef FB_C1G_EXPERIMENTS
    p.exp = getenv("FB_C1G_EXP") ? atoi(getenv("FB_C1G_EXP")) : 0;
#endif
    const int n_cu = fb_persistent_cus();
    p.n_workers = p.n_tiles < n_cu ? p.n_tiles : n_cu;
    if (a->stat_partial) hipLaunchKernelGGL((conv1x1_gemm_kernel<true>), dim3(p.n_workers), dim3(512), 0, st, p);
    else if (a->addend && a->addend_mask) hipLaunchKernelGGL((conv1x1_gemm_kernel<false, 2>), dim3(p.n_workers), dim3(512), 0, st, p);
    else if (a->addend) hipLaunchKernelGGL((conv1x1_gemm_kernel<false, 1>), dim3(p.n_workers), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv1x1_gemm_kernel<false>), dim3(p.n_workers), dim3(512), 0, st, p);
    return 1;
}
