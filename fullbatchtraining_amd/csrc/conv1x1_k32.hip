// The stem convolution as a streaming kernel.  The 3x3x3 (or 7x7x3) stem runs on pre-gathered patches (fb_stem_patches) as a 1x1
// convolution with K = 27 -> 32 input "channels" and 64 outputs: 64 bytes in, 128 bytes out per pixel, 21 FLOP per byte -- HBM-bound by
// a factor of 20.  The implicit-GEMM tile kernel treats it as a GEMM with ONE K-step: LDS staging, two barriers and the tile set-up per
// 128 pixels for a single MFMA deep (875 us per 12.8 M pixels = 2.8 TB/s of algorithmic traffic, 47 % of a streaming copy).
// Here nothing goes through LDS:
//   * the whole filter (64 x 32 bf16 = 4 KiB) lives in 16 registers per lane as the A operands of four 16x16x32 MFMAs
//   * a wave streams 32 pixels per trip: two 16-byte global loads per lane are exactly the B operands (pixel = lane & 15, k-group =
//     lane >> 4; a wave-load covers 16 whole 64-byte rows), requested one trip ahead; 8 MFMAs; outputs leave as four 16-byte stores per
//     lane after the v_permlane16_swap row exchange the other kernels use + a DPP exchange between lanes col and col ^ 8, so that every
//     wave-store writes 8 whole 128-byte pixels (plain stores: non-temporal ones cost 13 % here)
//   * 12 544 images: 638 us with the block split over four waves, half-line non-temporal stores; 596 with a wave per block; 529 with plain
//     stores; 482 with whole-line stores (5.1 TB/s of 2.46 GB)
//   * BatchNorm partial sums (per 128-pixel block, from the fp32 accumulators): a wave owns a block, DPP row sums, no LDS
//   * persistent workgroups, waves grid-stride over 128-pixel blocks
#include "common.h"
#include "conv_params.h"

namespace {
typedef __attribute__((ext_vector_type(4))) unsigned k32_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned k32_u32x2;
}

__global__ __launch_bounds__(256) void conv1x1_k32_kernel(const uint4* __restrict__ src, const uint4* __restrict__ wgt, char* __restrict__ dst,
                                                          float* __restrict__ stat, long long M, int n_blocks, int n_mblocks) {
#if defined(__HIP_DEVICE_COMPILE__)
    // A WAVE owns a whole 128-pixel statistics block (four trips of 32 pixels): its partial sums stay in registers across the trips and leave
    // with DPP row sums -- no LDS, no workgroup barrier (the first version split a block over the four waves of a workgroup and paid two
    // barriers per 128 pixels: 638 us per 12.8 M pixels where the bytes alone take 400)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, g = lane >> 4;
    // A operands: wgt [64 co][32 ci] bf16 = 4 x 16 bytes per row; fragment i holds co 16i + col, k-group g
    uint4 wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = wgt[(16 * i + col) * 4 + g];
    uint4 cur[2], nxt[2];
    auto fetch = [&](long long blk, int trip, uint4 (&pf)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long m = blk * 128 + trip * 32 + j * 16 + col;
            if (m < M) {
                const k32_u32x4 v = __builtin_nontemporal_load((const k32_u32x4*)(src + m * 4 + g));
                pf[j] = make_uint4(v[0], v[1], v[2], v[3]);
            } else {
                pf[j] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    const long long stride = (long long)gridDim.x * 4;
    long long blk = (long long)blockIdx.x * 4 + wave;
    if (blk < n_blocks) fetch(blk, 0, cur);
    for (; blk < n_blocks; blk += stride) {
        float ssum[4][4], ssq[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
#pragma unroll 1
        for (int trip = 0; trip < 4; ++trip) {
            if (trip < 3) fetch(blk, trip + 1, nxt);
            else if (blk + stride < n_blocks) fetch(blk + stride, 0, nxt);
            f32x4_t acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[i]), __builtin_bit_cast(bf16x8_t, cur[j]),
                                                                        (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned pk[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pk[i][0] = pack_bf16x2(acc[i][j][0], acc[i][j][1]); pk[i][1] = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ssum[i][r] += acc[i][j][r]; ssq[i][r] += acc[i][j][r] * acc[i][j][r]; }
                }
                // a lane holds channels 4g..4g+3 of each 16-channel fragment; the row swap of a fragment pair leaves it with 8 consecutive
                // channels ({0, 16, 8, 24}[g] of the pair's 32): 16 bytes per pair.  Stored like that, an instruction writes the first or the
                // second 64 bytes of 16 pixels -- half cache lines.  Lanes col and col ^ 8 trade one of their two pieces (DPP row rotate by 8), so that
                // an instruction writes BOTH halves of 8 pixels: 1 KiB contiguous per wave-store
                k32_u32x4 a0, a2;
                {
                    const k32_u32x2 lo = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                    const k32_u32x2 hi = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                    a0 = (k32_u32x4){lo[0], hi[0], lo[1], hi[1]};
                    const k32_u32x2 lo2 = __builtin_amdgcn_permlane16_swap(pk[2][0], pk[3][0], false, false);
                    const k32_u32x2 hi2 = __builtin_amdgcn_permlane16_swap(pk[2][1], pk[3][1], false, false);
                    a2 = (k32_u32x4){lo2[0], hi2[0], lo2[1], hi2[1]};
                }
                const bool upper = col >= 8;
                k32_u32x4 give, got;
#pragma unroll
                for (int e = 0; e < 4; ++e) give[e] = upper ? a0[e] : a2[e];
#pragma unroll
                for (int e = 0; e < 4; ++e) got[e] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)give[e], 0x128, 0xf, 0xf, false);   // row_ror:8
                k32_u32x4 s1, s2;                             // pixel (col & 7) and (col & 7) + 8 of this 16-pixel fragment; half = col >> 3
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[e] = upper ? got[e] : a0[e]; s2[e] = upper ? a2[e] : got[e]; }
                const long long m1 = blk * 128 + trip * 32 + j * 16 + (col & 7), m2 = m1 + 8;
                const int co = (col >> 3) * 32 + (g & 1) * 16 + (g >> 1) * 8;
                if (m1 < M) *(k32_u32x4*)(dst + (m1 * 64 + co) * 2) = s1;
                if (m2 < M) *(k32_u32x4*)(dst + (m2 * 64 + co) * 2) = s2;
            }
            cur[0] = nxt[0]; cur[1] = nxt[1];
        }
        if (stat != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { ssum[i][r] = row16_sum(ssum[i][r]); ssq[i][r] = row16_sum(ssq[i][r]); }
            if (col == 0) {                                  // four lanes (g): channels 16 i + 4 g .. + 3 of every fragment
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *(float4*)(stat + blk * 64 + i * 16 + g * 4) = make_float4(ssum[i][0], ssum[i][1], ssum[i][2], ssum[i][3]);
                    *(float4*)(stat + ((long long)n_mblocks + blk) * 64 + i * 16 + g * 4) = make_float4(ssq[i][0], ssq[i][1], ssq[i][2], ssq[i][3]);
                }
            }
        }
    }
#endif
}

// returns 1 if the kernel handled the call: forward 1x1, K = 32 bf16 inputs, 64 outputs, one shared weight set, no addend
int fb_try_conv1x1_k32(const fb_conv_args* a, hipStream_t st) {
    static const bool disabled = getenv("FB_DISABLE_STEM_STREAM") != nullptr;
    if (disabled) return 0;
    if (a->mode != 0 || a->R != 1 || a->S != 1 || a->stride != 1 || a->pad != 0 || a->dtype != FB_BF16) return 0;
    if (a->Cs != 32 || a->Cd != 64 || a->addend || a->Hs != a->Hd || a->Ws != a->Wd) return 0;
    if (a->wset_stride != 0 && a->imgs_per_wset > 0 && a->imgs_per_wset < a->n_img) return 0;
    const long long M = (long long)a->n_img * a->Hd * a->Wd;
    const long long n_blocks = (M + 127) / 128;
    if (n_blocks >= (1LL << 31)) return 0;
    const int n_cu = fb_persistent_cus();
    const long long n_wg = (n_blocks + 3) / 4;                  // a wave per 128-pixel block
    const int grid = (int)(n_wg < 8LL * n_cu ? n_wg : 8LL * n_cu);
    hipLaunchKernelGGL(conv1x1_k32_kernel, dim3(grid), dim3(256), 0, st, (const uint4*)a->src, (const uint4*)a->wgt, (char*)a->dst, a->stat_partial, M,
                       (int)n_blocks, (int)n_blocks);
    return 1;
}
