// The stem convolution as a streaming kernel.  The 3x3x3 (or 7x7x3) stem runs on pre-gathered patches (fb_stem_patches) as a 1x1
// convolution with K = 27 -> 32 input "channels" and 64 outputs: 64 bytes in, 128 bytes out per pixel, 21 FLOP per byte -- HBM-bound by
// a factor of 20.  The implicit-GEMM tile kernel treats it as a GEMM with ONE K-step: LDS staging, two barriers and the tile set-up per
// 128 pixels for a single MFMA deep (875 us per 12.8 M pixels = 2.8 TB/s of algorithmic traffic, 47 % of a streaming copy).
// Here nothing goes through LDS but the 4-wave statistics hand-over:
//   * the whole filter (64 x 32 bf16 = 4 KiB) lives in 16 registers per lane as the A operands of four 16x16x32 MFMAs
//   * a wave streams 32 pixels per trip: two 16-byte global loads per lane are exactly the B operands (pixel = lane & 15, k-group =
//     lane >> 4; a wave-load covers 16 whole 64-byte rows), requested one trip ahead; 8 MFMAs; outputs leave as four 16-byte stores per
//     lane after the v_permlane16_swap row exchange the other kernels use
//   * BatchNorm partial sums (per 128-pixel block, from the fp32 accumulators) by DPP row sums + one LDS hand-over between the 4 waves
//   * persistent workgroups, grid-stride over 128-pixel blocks
#include "common.h"
#include "conv_params.h"

namespace {
typedef __attribute__((ext_vector_type(4))) unsigned k32_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned k32_u32x2;
}

__global__ __launch_bounds__(256) void conv1x1_k32_kernel(const uint4* __restrict__ src, const uint4* __restrict__ wgt, char* __restrict__ dst,
                                                          float* __restrict__ stat, long long M, int n_blocks, int n_mblocks) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ float red[4][64][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, g = lane >> 4;
    // A operands: wgt [64 co][32 ci] bf16 = 4 x 16 bytes per row; fragment i holds co 16i + col, k-group g
    uint4 wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = wgt[(16 * i + col) * 4 + g];
    uint4 cur[2], nxt[2];
    auto fetch = [&](long long blk, uint4 (&pf)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long m = blk * 128 + wave * 32 + j * 16 + col;
            if (m < M) {
                const k32_u32x4 v = __builtin_nontemporal_load((const k32_u32x4*)(src + m * 4 + g));
                pf[j] = make_uint4(v[0], v[1], v[2], v[3]);
            } else {
                pf[j] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    long long blk = blockIdx.x;
    if (blk < n_blocks) fetch(blk, cur);
    for (; blk < n_blocks; blk += gridDim.x) {
        const long long nb = blk + gridDim.x;
        if (nb < n_blocks) fetch(nb, nxt);
        f32x4_t acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[i]), __builtin_bit_cast(bf16x8_t, cur[j]),
                                                                    (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        float ssum[4][4], ssq[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long m = blk * 128 + wave * 32 + j * 16 + col;
            unsigned pk[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pk[i][0] = pack_bf16x2(acc[i][j][0], acc[i][j][1]); pk[i][1] = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) { ssum[i][r] += acc[i][j][r]; ssq[i][r] += acc[i][j][r] * acc[i][j][r]; }
            }
            // a lane holds channels 4g..4g+3 of each 16-channel fragment; the row swap of a fragment pair leaves it with 8 consecutive
            // channels ({0, 16, 8, 24}[g] of the pair's 32): one 16-byte store per pair
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const k32_u32x2 lo = __builtin_amdgcn_permlane16_swap(pk[i][0], pk[i + 1][0], false, false);
                const k32_u32x2 hi = __builtin_amdgcn_permlane16_swap(pk[i][1], pk[i + 1][1], false, false);
                const int co = i * 16 + (g & 1) * 16 + (g >> 1) * 8;
                if (m < M) __builtin_nontemporal_store((k32_u32x4){lo[0], hi[0], lo[1], hi[1]}, (k32_u32x4*)(dst + (m * 64 + co) * 2));
            }
        }
        if (stat != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = row16_sum(ssum[i][r]), b = row16_sum(ssq[i][r]);
                    if (col == 0) { red[wave][i * 16 + g * 4 + r][0] = a; red[wave][i * 16 + g * 4 + r][1] = b; }
                }
            __syncthreads();
            if (tid < 64) {
                const float a = ((red[0][tid][0] + red[1][tid][0]) + red[2][tid][0]) + red[3][tid][0];
                const float b = ((red[0][tid][1] + red[1][tid][1]) + red[2][tid][1]) + red[3][tid][1];
                stat[blk * 64 + tid] = a;
                stat[((long long)n_mblocks + blk) * 64 + tid] = b;
            }
            __syncthreads();
        }
        cur[0] = nxt[0]; cur[1] = nxt[1];
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
    const int grid = (int)(n_blocks < 8LL * n_cu ? n_blocks : 8LL * n_cu);
    hipLaunchKernelGGL(conv1x1_k32_kernel, dim3(grid), dim3(256), 0, st, (const uint4*)a->src, (const uint4*)a->wgt, (char*)a->dst, a->stat_partial, M,
                       (int)n_blocks, (int)n_blocks);
    return 1;
}
