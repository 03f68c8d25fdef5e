// Weight-gradient convolution on MFMA for gfx950:  dW[co][tap][ci] = sum_p dY[p][co] * X[src(p,tap)][ci]  per chunk.
//
// GEMM view: M = co, N = ci, K = pixels.  Both operands arrive pixel-major (NHWC rows), i.e. K is the *strided*
// dimension, so the MFMA fragments are read transposed out of LDS:
//   bf16: ds_read_b64_tr_b16 (hardware 4x16 transpose read) -- two reads give the 8 k-values a lane needs
//   f32 : ds_read_b32 per element (v_mfma_f32_16x16x4_f32 takes one f32 per lane)
// Every wave owns a 64x64 (co x ci) accumulator tile; the template chooses how the 4 waves are arranged:
//   <WM=1,WN=1,WK=4>: one 64x64 tile per block, waves split the pixel range (reduced through LDS in fixed order)
//   <WM=2,WN=2,WK=1>: 128x128 tile per block
// Grid: x = tile, y = tap, z = group * split_k.  Output: fp32 slabs [group][split][co][tap][ci], no atomics, so the
// sum order is fixed and identical for the two passes of the finite-difference regulariser.
#include "common.h"
#include "profile.h"

struct WgradParams {
    const char* x; const char* dy; float* out;
    int n_img, Hs, Ws, Cs, Hd, Wd, Cd;
    int R, S, stride, pad;
    int imgs_per_group, split_k, px_per_group, px_per_split;
    long long group_stride;                                 // floats between the slabs of consecutive groups
    const float* amax_x; const float* amax_dy;              // f32h (fp16x2 split): largest magnitudes of the two operand tensors
    const char* bn_x; const unsigned char* bn_mask; const float* bn_coef;   // BNF: BatchNorm backward apply inside the dy loader
};

template <typename T> struct WG;
template <> struct WG<bf16_tag> { static constexpr int PAD = 16; };
template <> struct WG<float> { static constexpr int PAD = 64; };
template <> struct WG<f32s_tag> : WG<float> {};
template <> struct WG<f32h_tag> : WG<float> {};

// transposed fragment: for channels c0..c0+15 (lane&15) and pixels pb + 8*(lane>>4) .. +7 (bf16) -> one 16-byte chunk
template <typename T> __device__ __forceinline__ void load_frag_t(const char* tile, int row_bytes, int pb, int c0, int lane, uint4 (&out)[2]);
template <> __device__ __forceinline__ void load_frag_t<bf16_tag>(const char* tile, int row_bytes, int pb, int c0, int lane, uint4 (&out)[2]) {
    // 32 pixels x 16 channels -> bf16x8 per lane: k = 8*(lane>>4)+e, e=0..7 ; out[1] unused
    const int t = lane & 15, g = lane >> 4;
    const char* a0 = tile + (pb + g * 8 + (t >> 2)) * row_bytes + (c0 + (t & 3) * 4) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0 + 4 * row_bytes));
    out[0] = make_uint4(((unsigned)(unsigned short)lo[0]) | ((unsigned)(unsigned short)lo[1] << 16),
                        ((unsigned)(unsigned short)lo[2]) | ((unsigned)(unsigned short)lo[3] << 16),
                        ((unsigned)(unsigned short)hi[0]) | ((unsigned)(unsigned short)hi[1] << 16),
                        ((unsigned)(unsigned short)hi[2]) | ((unsigned)(unsigned short)hi[3] << 16));
}
template <> __device__ __forceinline__ void load_frag_t<float>(const char* tile, int row_bytes, int pb, int c0, int lane, uint4 (&out)[2]) {
    // 32 pixels x 16 channels -> 8 floats per lane: out[h] element e <-> pixel pb + 16h + 4e + (lane>>4)
    const int t = lane & 15, g = lane >> 4;
    const char* a0 = tile + (pb + g) * row_bytes + (c0 + t) * 4;
    unsigned v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = *(const unsigned*)(a0 + 4 * e * row_bytes);
    out[0] = make_uint4(v[0], v[1], v[2], v[3]);
    out[1] = make_uint4(v[4], v[5], v[6], v[7]);
}

// BNF (bf16): the A operand is computed in the loader from (dout, x, ReLU bitmask, coefficients) -- the dx of fb_bn_bwd_apply, same fp32
// expression, same rounding to bf16 -- instead of being read from memory (fb_wgrad_args.bn_x)
template <typename T, int WM, int WN, int WK, int KREP, int NJ, bool BNF = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int EB = ET<T>::EB;
    constexpr int TM = 64 * WM, TN = 16 * NJ * WN;     // block tile (co x ci); wave tile 64 x 16*NJ
    constexpr int KPX = 32 * WK * KREP;                // pixels per block K-step
    constexpr int ROW_A = TM * EB + WG<T>::PAD, ROW_B = TN * EB + WG<T>::PAD;
    constexpr int CH_A = TM * EB / 16, CH_B = TN * EB / 16;   // 16-byte chunks per row
    constexpr int LD_A = KPX * CH_A / 256, LD_B = KPX * CH_B / 256;   // chunks per thread
    static_assert(KPX * CH_A % 256 == 0 && KPX * CH_B % 256 == 0, "tile/threads");
    // split path (fp32 storage, bf16x6 arithmetic): a tile is stored to LDS as THREE bf16 planes (h, m, l: every loaded element is
    // split once, 22 VALU per 16-byte chunk) and the fragments come from the transposed bf16 reads of the bf16 kernel -- no per-fragment
    // split (44 VALU per fragment and wave before), 6 ds_read_b64_tr_b16 instead of 8 ds_read_b32 per fragment
    // f32h: TWO scaled fp16 planes and three MFMAs per fragment pair (common.h); the accumulators hold the scaled sums, the epilogue
    // multiplies by 2^-(e_x + e_dy)
    constexpr bool HSPLIT = is_hsplit<T>::value;
    constexpr bool SPLIT = is_split<T>::value || HSPLIT;
    constexpr int NPL = HSPLIT ? 2 : 3;
    constexpr int PROW_A = TM * 2 + WG<bf16_tag>::PAD, PROW_B = TN * 2 + WG<bf16_tag>::PAD;      // plane rows (bf16 / fp16)
    constexpr int PLANE_A = KPX * PROW_A, PLANE_B = KPX * PROW_B;
    constexpr int TILE_BYTES = SPLIT ? NPL * (PLANE_A + PLANE_B) : KPX * (ROW_A + ROW_B);
    static_assert(WK == 1 || WM == 1, "the cross-wave K reduction covers one 64-row block tile");
    constexpr int RED_BYTES = (WK > 1) ? WK * 64 * TN * 4 : 0;
    constexpr int LDS_BYTES = TILE_BYTES > RED_BYTES ? TILE_BYTES : RED_BYTES;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    char* tileA = lds;
    char* tileB = lds + (SPLIT ? NPL * PLANE_A : KPX * ROW_A);
    float hs_a = 1.f, hs_b = 1.f;
    if constexpr (HSPLIT) { hs_a = fb_pow2_scale(p.amax_dy[blockIdx.z / p.split_k]); hs_b = fb_pow2_scale(p.amax_x[blockIdx.z / p.split_k]); }   // per chunk

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    const int tiles_n = p.Cs / TN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int tap = blockIdx.y, r = tap / p.S, s = tap % p.S;
    const int group = blockIdx.z / p.split_k, split = blockIdx.z % p.split_k;
    const int HWd = p.Hd * p.Wd;
    const long long gp0 = (long long)group * p.px_per_group;       // first output pixel of the group
    const int k_begin = split * p.px_per_split;
    const int k_end = min(k_begin + p.px_per_split, p.px_per_group);

    f32x4_t acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    uint4 ra[LD_A], rb[LD_B];
    uint4 rx[BNF ? LD_A : 1];                               // BNF: the BatchNorm input at the same positions
    unsigned rm[BNF ? LD_A : 1];
    constexpr int BV = 16 / EB;                             // channels of one 16-byte chunk
    float cf[BNF ? 3 * BV : 1];                             // BNF: three coefficients for each channel of this thread's chunk (tid % CH_A)
    if constexpr (BNF) {
        static_assert(256 % CH_A == 0 && !HSPLIT, "BNF: fixed channel chunk per thread; not for the fp16x2 planes (their scale needs the materialised dx)");
        const float* c = p.bn_coef + ((long long)group * p.Cd + tile_m * TM + (tid % CH_A) * BV) * 3;
#pragma unroll
        for (int q = 0; q < 3 * BV; ++q) cf[q] = c[q];
    }
    // dx = c_dy * dy + c_x * x + c_0 (fb_bn_bwd_apply: same expression, same rounding) of chunk i, in place
    auto bn_dx_chunk = [&](int i) {
        if constexpr (BNF) {
            float d[BV], xv[BV], o[BV];
            ET<T>::unpack(ra[i], d); ET<T>::unpack(rx[i], xv);
#pragma unroll
            for (int q = 0; q < BV; ++q) {
                const float dy = ((rm[i] >> q) & 1u) ? d[q] : 0.f;
                o[q] = fb_bn_dx(cf[3 * q], cf[3 * q + 1], cf[3 * q + 2], dy, xv[q]);
                if (rm[i] & 0x100u) o[q] = 0.f;
            }
            ra[i] = ET<T>::pack(o);
        }
    };
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LD_A; ++i) {
            const int id = tid + 256 * i, row = id / CH_A, ch = id % CH_A;
            const int k = k0 + row;
            ra[i] = make_uint4(0, 0, 0, 0);
            if constexpr (BNF) { rx[i] = make_uint4(0, 0, 0, 0); rm[i] = 0x100u; }       // (bit 8: a row beyond the slice contributes exactly zero)
            if (k < k_end) {
                const long long e = ((gp0 + k) * p.Cd + tile_m * TM) * EB + ch * 16;
                ra[i] = *(const uint4*)(p.dy + e);
                if constexpr (BNF) { rx[i] = *(const uint4*)(p.bn_x + e); rm[i] = p.bn_mask ? p.bn_mask[e >> 4] : 0xffu; }
            }
        }
#pragma unroll
        for (int i = 0; i < LD_B; ++i) {
            const int id = tid + 256 * i, row = id / CH_B, ch = id % CH_B;
            const int k = k0 + row;
            rb[i] = make_uint4(0, 0, 0, 0);
            if (k < k_end) {
                const long long gp = gp0 + k;
                const int n = (int)(gp / HWd), rem = (int)(gp - (long long)n * HWd), oy = rem / p.Wd, ox = rem - oy * p.Wd;
                const int sy = oy * p.stride - p.pad + r, sx = ox * p.stride - p.pad + s;
                if ((unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws)
                    rb[i] = *(const uint4*)(p.x + ((((long long)n * p.Hs + sy) * p.Ws + sx) * p.Cs + tile_n * TN) * EB + ch * 16);
            }
        }
    };
    auto lstore = [&]() {
        if constexpr (SPLIT) {
#pragma unroll
            for (int i = 0; i < LD_A; ++i) {
                const int id = tid + 256 * i, row = id / CH_A, ch = id % CH_A;
                uint2 h, m, l;
                bn_dx_chunk(i);
                char* d = tileA + row * PROW_A + ch * 8;
                if constexpr (HSPLIT) { split_h2x4(ra[i], hs_a, h, l); *(uint2*)d = h; *(uint2*)(d + PLANE_A) = l; }
                else { split_f32x4(ra[i], h, m, l); *(uint2*)d = h; *(uint2*)(d + PLANE_A) = m; *(uint2*)(d + 2 * PLANE_A) = l; }
            }
#pragma unroll
            for (int i = 0; i < LD_B; ++i) {
                const int id = tid + 256 * i, row = id / CH_B, ch = id % CH_B;
                uint2 h, m, l;
                char* d = tileB + row * PROW_B + ch * 8;
                if constexpr (HSPLIT) { split_h2x4(rb[i], hs_b, h, l); *(uint2*)d = h; *(uint2*)(d + PLANE_B) = l; }
                else { split_f32x4(rb[i], h, m, l); *(uint2*)d = h; *(uint2*)(d + PLANE_B) = m; *(uint2*)(d + 2 * PLANE_B) = l; }
            }
        } else {
#pragma unroll
        for (int i = 0; i < LD_A; ++i) {
            const int id = tid + 256 * i, row = id / CH_A, ch = id % CH_A;
            bn_dx_chunk(i);
            *(uint4*)(tileA + row * ROW_A + ch * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < LD_B; ++i) { const int id = tid + 256 * i, row = id / CH_B, ch = id % CH_B; *(uint4*)(tileB + row * ROW_B + ch * 16) = rb[i]; }
        }
    };

    if (k_begin < k_end) gload(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += KPX) {
        __syncthreads();           // previous step's fragment reads are done
        lstore();
        __syncthreads();
        if (k0 + KPX < k_end) gload(k0 + KPX);
#pragma unroll
        for (int rep = 0; rep < KREP; ++rep) {
            const int pb = (wk * KREP + rep) * 32;
            if constexpr (SPLIT) {
                uint4 ap[4][NPL], bp[NJ][NPL], tmp[2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) { load_frag_t<bf16_tag>(tileA + pl * PLANE_A, PROW_A, pb, wm * 64 + i * 16, lane, tmp); ap[i][pl] = tmp[0]; }
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) { load_frag_t<bf16_tag>(tileB + pl * PLANE_B, PROW_B, pb, wn * 16 * NJ + j * 16, lane, tmp); bp[j][pl] = tmp[0]; }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if constexpr (HSPLIT) acc[i][j] = mma_planes3h_acc(ap[i], bp[j], acc[i][j]);
                        else acc[i][j] = mma_planes6(ap[i], bp[j], acc[i][j]);
                    }
            } else {
            uint4 af[4][2], bf[NJ][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) load_frag_t<T>(tileA, ROW_A, pb, wm * 64 + i * 16, lane, af[i]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) load_frag_t<T>(tileB, ROW_B, pb, wn * 16 * NJ + j * 16, lane, bf[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[i][j] = mma_chunk<T>(af[i][0], bf[j][0], acc[i][j]);
                    if constexpr (EB == 4) acc[i][j] = mma_chunk<T>(af[i][1], bf[j][1], acc[i][j]);
                }
            }
        }
    }

    if constexpr (HSPLIT) {                                  // the accumulators hold sums of (2^e_dy dy) * (2^e_x x)
        const float inv = 1.f / (hs_a * hs_b);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][j][q] *= inv;
    }
    // ---- output: [group][split][co][tap][ci] -------------------------------------------------------------------------
    const int taps = p.R * p.S;
    float* out = p.out + group * p.group_stride + ((long long)split * p.Cd) * taps * p.Cs;
    if constexpr (WK > 1) {
        float* red = (float*)lds;   // [WK][64][TN]
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[(wk * 64 + i * 16 + (lane >> 4) * 4 + q) * TN + wn * 16 * NJ + j * 16 + (lane & 15)] = acc[i][j][q];
        __syncthreads();
        for (int e = tid; e < 64 * TN; e += 256) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WK; ++w) v += red[w * 64 * TN + e];
            const int co = tile_m * TM + e / TN, ci = tile_n * TN + e % TN;
            out[((long long)co * taps + tap) * p.Cs + ci] = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = tile_m * TM + wm * 64 + i * 16 + (lane >> 4) * 4 + q;
                    const int ci = tile_n * TN + wn * 16 * NJ + j * 16 + (lane & 15);
                    out[((long long)co * taps + tap) * p.Cs + ci] = acc[i][j][q];
                }
    }
}

int fb_try_wgrad3x3(const fb_wgrad_args* a, hipStream_t st);      // conv_wgrad3x3.hip
int fb_try_wgrad3x3_v2(const fb_wgrad_args* a, hipStream_t st);   // conv_wgrad3x3_v2.hip
int fb_try_wgrad1x1(const fb_wgrad_args* a, hipStream_t st);      // conv_wgrad1x1.hip

// BatchNorm backward apply inside the loader: bf16, 1x1, the 64 x 32-channel tiles of the stem (register-staged operands)
extern "C" int32_t fb_wgrad_bn_fused_supported(const fb_wgrad_args* a) {
    static const bool disabled = getenv("FB_DISABLE_WGRAD_BNF") != nullptr;
    return !disabled && a && !a->amax_x && !a->amax_dy && a->R == 1 && a->S == 1 && a->stride == 1 && a->pad == 0 && a->Cs % 32 == 0 && a->Cs % 64 != 0
           && a->Cd % 64 == 0 && a->Hs == a->Hd && a->Ws == a->Wd;
}

extern "C" int fb_conv2d_wgrad(const fb_wgrad_args* a, void* stream) {
    if (!a || !a->x || !a->dy || !a->dw_partial) FB_FAIL(FB_ERR_ARG, "fb_conv2d_wgrad: null pointer");
    if (a->bn_x && (!a->bn_coef || !fb_wgrad_bn_fused_supported(a))) FB_FAIL(FB_ERR_UNSUPPORTED, "fb_conv2d_wgrad: bn_x (BatchNorm apply in the loader) is for 1x1 layers with Cs < 64 (no fp16x2 planes) and needs bn_coef");
    if (a->Cs % 32 != 0 || a->Cd % 64 != 0) FB_FAIL(FB_ERR_SHAPE, "fb_conv2d_wgrad: Cs=%d must be a multiple of 32, Cd=%d of 64", a->Cs, a->Cd);
    if (a->n_img % a->imgs_per_group != 0) FB_FAIL(FB_ERR_SHAPE, "fb_conv2d_wgrad: n_img %% imgs_per_group != 0");
    if (a->split_k < 1) FB_FAIL(FB_ERR_ARG, "fb_conv2d_wgrad: split_k < 1");
    WgradParams p;
    p.x = (const char*)a->x; p.dy = (const char*)a->dy; p.out = a->dw_partial;
    p.n_img = a->n_img; p.Hs = a->Hs; p.Ws = a->Ws; p.Cs = a->Cs; p.Hd = a->Hd; p.Wd = a->Wd; p.Cd = a->Cd;
    p.R = a->R; p.S = a->S; p.stride = a->stride; p.pad = a->pad;
    p.imgs_per_group = a->imgs_per_group; p.split_k = a->split_k;
    p.group_stride = a->group_stride ? a->group_stride : (long long)a->split_k * a->Cd * a->R * a->S * a->Cs;
    p.px_per_group = a->imgs_per_group * a->Hd * a->Wd;
    p.amax_x = a->dtype == FB_F32 ? a->amax_x : nullptr; p.amax_dy = a->dtype == FB_F32 ? a->amax_dy : nullptr;
    p.bn_x = (const char*)a->bn_x; p.bn_mask = (const unsigned char*)a->bn_mask; p.bn_coef = a->bn_coef;
    if ((p.amax_x == nullptr) != (p.amax_dy == nullptr)) FB_FAIL(FB_ERR_ARG, "fb_conv2d_wgrad: amax_x and amax_dy go together");
    const bool hsplit = p.amax_x != nullptr;
    const int n_groups = a->n_img / a->imgs_per_group;
    hipStream_t st = (hipStream_t)stream;
    // 128 x 128 tiles (four waves of 64 x 64) where both channel counts allow it and there are enough tiles; 3x3 layers of 128 channels too (round 5: on
    // 64 x 64 tiles every operand is streamed 18 times through L2 -- 62 fp32-TFLOP/s on the 28 x 28 maps of ResNet-152 against 152 for the 256-channel layers;
    // bf16: the one stride-2 layer on 56 x 56 maps that the all-taps kernels do not cover, 178 TFLOP/s).
    // fullbatchtraining_amd/engine.py::_choose_split mirrors this choice (K-slice counts).
    const bool big = (a->Cs % 128 == 0) && (a->Cd % 128 == 0) && (a->Cs >= 256 || a->Cd >= 256 || a->R == 3);
    const int kstep = big ? (a->dtype == FB_F32 ? 32 : 64) : 128;
    p.px_per_split = (int)(ceil_div64(ceil_div64(p.px_per_group, a->split_k), kstep) * kstep);
    const int taps = a->R * a->S;
    const bool split = fb_f32_split_enabled();
    const int32_t info[FB_PROF_INFO] = {a->n_img, a->Hs, a->Ws, a->Cs, a->Hd, a->Wd, a->Cd, a->R, a->stride, a->split_k | (a->dtype << 16), FB_K_WGRAD_GENERIC};
    const int prof = fb_prof_begin(FB_PROF_WGRAD, st, info);
    // the ImageNet stem's 7 x 7 x 3 patches (147 -> 160 "channels"), bf16: ONE 64 x 160 tile per workgroup instead of five 64 x 32 tiles that each read dy (and, with
    // the BatchNorm apply in the loader, x and the mask) again -- the launch is a streaming reduction over 12 544 pixels per image (round 5: 1.83 ms
    // per 512 images at 1.6 TB/s of algorithmic bytes before)
    static const bool wide160 = getenv("FB_DISABLE_WGRAD_WIDE160") == nullptr;
    if (wide160 && a->dtype == FB_BF16 && a->R == 1 && a->S == 1 && a->Cs % 160 == 0 && a->Cs % 64 != 0 && a->Cd % 64 == 0) {
        dim3 grid((a->Cd / 64) * (a->Cs / 160), taps, n_groups * a->split_k);
        if (a->bn_x) hipLaunchKernelGGL((conv_wgrad_kernel<bf16_tag, 1, 2, 2, 2, 5, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad_kernel<bf16_tag, 1, 2, 2, 2, 5, false>), grid, dim3(256), 0, st, p);
    } else if (a->bn_x) {
        dim3 grid((a->Cd / 64) * (a->Cs / 32), taps, n_groups * a->split_k);
        if (a->dtype == FB_F32 && split) hipLaunchKernelGGL((conv_wgrad_kernel<f32s_tag, 1, 1, 4, 1, 2, true>), grid, dim3(256), 0, st, p);
        else if (a->dtype == FB_F32) hipLaunchKernelGGL((conv_wgrad_kernel<float, 1, 1, 4, 1, 2, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad_kernel<bf16_tag, 1, 1, 4, 1, 2, true>), grid, dim3(256), 0, st, p);
    } else if (fb_try_wgrad3x3_v2(a, st)) { fb_prof_kernel(prof, FB_K_WGRAD3X3_V2);
    } else if (fb_try_wgrad3x3(a, st)) { fb_prof_kernel(prof, FB_K_WGRAD3X3_V1);
    } else if (fb_try_wgrad1x1(a, st)) { fb_prof_kernel(prof, FB_K_WGRAD1X1);
    } else if (big) {
        dim3 grid((a->Cd / 128) * (a->Cs / 128), taps, n_groups * a->split_k);
        if (hsplit) hipLaunchKernelGGL((conv_wgrad_kernel<f32h_tag, 2, 2, 1, 1, 4>), grid, dim3(256), 0, st, p);
        else if (a->dtype == FB_F32 && split) hipLaunchKernelGGL((conv_wgrad_kernel<f32s_tag, 2, 2, 1, 1, 4>), grid, dim3(256), 0, st, p);
        else if (a->dtype == FB_F32) hipLaunchKernelGGL((conv_wgrad_kernel<float, 2, 2, 1, 1, 4>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad_kernel<bf16_tag, 2, 2, 1, 2, 4>), grid, dim3(256), 0, st, p);
    } else if (a->Cs % 64 == 0) {
        dim3 grid((a->Cd / 64) * (a->Cs / 64), taps, n_groups * a->split_k);
        if (hsplit) {
            static const bool k4h = getenv("FB_WGRAD_F32S_K4") != nullptr;
            if (k4h) hipLaunchKernelGGL((conv_wgrad_kernel<f32h_tag, 1, 1, 4, 1, 4>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((conv_wgrad_kernel<f32h_tag, 1, 2, 2, 1, 2>), grid, dim3(256), 0, st, p);      // (the 2 x 2 wave arrangement of the bf16x6 form below)
        }
        else if (a->dtype == FB_F32 && split) {
            // bf16x6 (three planes per operand in LDS): two waves across the 64 input channels and two across K, 64 pixels per K-step -- 54 KiB and two workgroups per CU
            // instead of four waves across K on 128-pixel steps (108 KiB, one workgroup per CU: 58 fp32-TFLOP/s on the 64-channel layers of ResNet-152 @224, 47 ms of
            // the regularised step); the four K partial sums of an output element become two (another summation order: FB_WGRAD_F32S_K4=1 restores the old form)
            static const bool k4 = getenv("FB_WGRAD_F32S_K4") != nullptr;
            if (k4) hipLaunchKernelGGL((conv_wgrad_kernel<f32s_tag, 1, 1, 4, 1, 4>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((conv_wgrad_kernel<f32s_tag, 1, 2, 2, 1, 2>), grid, dim3(256), 0, st, p);
        }
        else if (a->dtype == FB_F32) hipLaunchKernelGGL((conv_wgrad_kernel<float, 1, 1, 4, 1, 4>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad_kernel<bf16_tag, 1, 1, 4, 1, 4>), grid, dim3(256), 0, st, p);
    } else {   // Cs multiple of 32 only (pre-gathered stem patches): 64 x 32 tiles
        dim3 grid((a->Cd / 64) * (a->Cs / 32), taps, n_groups * a->split_k);
        if (hsplit) hipLaunchKernelGGL((conv_wgrad_kernel<f32h_tag, 1, 1, 4, 1, 2>), grid, dim3(256), 0, st, p);
        else if (a->dtype == FB_F32 && split) hipLaunchKernelGGL((conv_wgrad_kernel<f32s_tag, 1, 1, 4, 1, 2>), grid, dim3(256), 0, st, p);
        else if (a->dtype == FB_F32) hipLaunchKernelGGL((conv_wgrad_kernel<float, 1, 1, 4, 1, 2>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad_kernel<bf16_tag, 1, 1, 4, 1, 2>), grid, dim3(256), 0, st, p);
    }
    fb_prof_end(prof, st);
    FB_CHECK_LAUNCH("fb_conv2d_wgrad");
    return FB_OK;
}

// ---- split-K slab reduction + channel-padding drop -------------------------------------------------------------------
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long long out_group_stride,
                                    int split_k, long long rows, int Cs_pad, int Cs_real) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over rows*Cs_real
    const int g = blockIdx.y;
    if (idx >= rows * Cs_real) return;
    const long long row = idx / Cs_real; const int ci = (int)(idx - row * Cs_real);
    const float* src = part + ((long long)g * split_k) * rows * Cs_pad + row * Cs_pad + ci;
    float v = 0.f;
    for (int s = 0; s < split_k; ++s) v += src[(long long)s * rows * Cs_pad];
    out[(long long)g * out_group_stride + idx] = v;
}

extern "C" int fb_wgrad_reduce(const float* dw_partial, float* out, int64_t out_group_stride, int32_t n_groups, int32_t split_k,
                               int32_t Cd, int32_t taps, int32_t Cs_pad, int32_t Cs_real, void* stream) {
    if (!dw_partial || !out) FB_FAIL(FB_ERR_ARG, "fb_wgrad_reduce: null pointer");
    const long long rows = (long long)Cd * taps, total = rows * Cs_real;
    dim3 grid((unsigned)ceil_div64(total, 256), n_groups);
    hipLaunchKernelGGL(wgrad_reduce_kernel, grid, dim3(256), 0, (hipStream_t)stream, dw_partial, out, (long long)out_group_stride,
                       split_k, rows, Cs_pad, Cs_real);
    FB_CHECK_LAUNCH("fb_wgrad_reduce");
    return FB_OK;
}

// ---- master KRSC fp32 -> compute copies ------------------------------------------------------------------------------
template <typename T>
__global__ void weight_prep_kernel(const float* __restrict__ master, long long wset_stride_in, long long wset_stride_out, int Cout, int taps,
                                   int Cin_real, int Cin_pad, T* __restrict__ w_fwd, T* __restrict__ w_dgrad) {
    const long long per_set = (long long)Cout * taps * Cin_pad;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ws = blockIdx.y;
    if (idx >= per_set) return;
    const int ci = (int)(idx % Cin_pad); const long long rt = idx / Cin_pad; const int t = (int)(rt % taps); const int co = (int)(rt / taps);
    float v = 0.f;
    if (ci < Cin_real) v = master[(long long)ws * wset_stride_in + ((long long)co * taps + t) * Cin_real + ci];
    T o;
    if constexpr (sizeof(T) == 4) o = v; else o = f32_to_bf16(v);
    w_fwd[(long long)ws * wset_stride_out + idx] = o;
    if (w_dgrad) w_dgrad[(long long)ws * wset_stride_out + ((long long)ci * taps + t) * Cout + co] = o;
}

// fp16x2 planes (common.h, f32h_tag): every 128-byte group of 32 fp32 values becomes [32 fp16 high pieces | 32 fp16 low pieces] of the
// values scaled by the layer's power of two, in the K-slot order of the convolution kernels' fragment reads: the lane group g of an MFMA
// step reads 16-byte chunk g (its high pieces) and chunk g + 4 (its low pieces), and the fp32 activations it splits on the fly are the
// values {4g..4g+3, 16+4g..16+4g+3} of the group -- so value j sits at slot 8*(j%16/4) + 4*(j/16) + j%4.
// forward copy: one thread per (row = (co, tap), 32-channel group): 32 consecutive master values in, 64 + 64 contiguous bytes out
__global__ __launch_bounds__(256) void weight_prep_planes_fwd_kernel(const float* __restrict__ master, long long wset_stride_in, long long wset_stride_out,
                                                                     int Cout, int taps, int Cin_real, int Cin_pad, _Float16* __restrict__ w_fwd,
                                                                     const float* __restrict__ amax) {
    const int groups = Cin_pad / 32;
    const long long n = (long long)Cout * taps * groups;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ws = blockIdx.y;
    if (idx >= n) return;
    const int q = (int)(idx % groups);
    const long long rt = idx / groups;
    const float sc = fb_pow2_scale(amax[ws]);
    const float* src = master + (long long)ws * wset_stride_in + rt * Cin_real + q * 32;
    unsigned hi[16], lo[16];
#pragma unroll
    for (int j = 0; j < 32; j += 2) {
        // slot order of the planes: value j of the group sits at slot 8*(j%16/4) + 4*(j/16) + j%4 -> fill slot pairs (s, s+1)
        const int s0 = j;                                    // slots s0, s0+1 hold values v(s0), v(s0+1)
        const int v0 = 4 * (s0 >> 3) + (s0 & 3) + 16 * ((s0 >> 2) & 1), v1 = v0 + 1;
        const float a = (q * 32 + v0 < Cin_real ? src[v0] : 0.f) * sc, b = (q * 32 + v1 < Cin_real ? src[v1] : 0.f) * sc;
        hi[j >> 1] = pack_f16x2(a, b);
        const f16x2_t hv = __builtin_bit_cast(f16x2_t, hi[j >> 1]);
        lo[j >> 1] = pack_f16x2(a - (float)hv[0], b - (float)hv[1]);
    }
    uint4* dst = (uint4*)(w_fwd + ((long long)ws * wset_stride_out + rt * Cin_pad + q * 32) * 2);
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = make_uint4(hi[4 * k], hi[4 * k + 1], hi[4 * k + 2], hi[4 * k + 3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[4 + k] = make_uint4(lo[4 * k], lo[4 * k + 1], lo[4 * k + 2], lo[4 * k + 3]);
}
// transposed copy [ci][tap][co]: one thread per (ci, tap, 32-channel group of co); neighbouring threads = neighbouring ci, so the 32 strided
// master reads of a thread are coalesced across the wave, and the thread writes its own 64 + 64 contiguous bytes
__global__ __launch_bounds__(256) void weight_prep_planes_dgrad_kernel(const float* __restrict__ master, long long wset_stride_in, long long wset_stride_out,
                                                                       int Cout, int taps, int Cin_real, int Cin_pad, _Float16* __restrict__ w_dgrad,
                                                                       const float* __restrict__ amax) {
    const int groups = Cout / 32;
    const long long n = (long long)Cin_pad * taps * groups;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ws = blockIdx.y;
    if (idx >= n) return;
    const int ci = (int)(idx % Cin_pad);
    const long long r2 = idx / Cin_pad;
    const int t = (int)(r2 % taps), q = (int)(r2 / taps);
    const float sc = fb_pow2_scale(amax[ws]);
    const float* src = master + (long long)ws * wset_stride_in + ((long long)(q * 32) * taps + t) * Cin_real + ci;
    const long long cstride = (long long)taps * Cin_real;
    unsigned hi[16], lo[16];
#pragma unroll
    for (int j = 0; j < 32; j += 2) {
        const int v0 = 4 * (j >> 3) + (j & 3) + 16 * ((j >> 2) & 1), v1 = v0 + 1;
        const float a = (ci < Cin_real ? src[v0 * cstride] : 0.f) * sc, b = (ci < Cin_real ? src[v1 * cstride] : 0.f) * sc;
        hi[j >> 1] = pack_f16x2(a, b);
        const f16x2_t hv = __builtin_bit_cast(f16x2_t, hi[j >> 1]);
        lo[j >> 1] = pack_f16x2(a - (float)hv[0], b - (float)hv[1]);
    }
    uint4* dst = (uint4*)(w_dgrad + ((long long)ws * wset_stride_out + ((long long)ci * taps + t) * Cout + q * 32) * 2);
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = make_uint4(hi[4 * k], hi[4 * k + 1], hi[4 * k + 2], hi[4 * k + 3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[4 + k] = make_uint4(lo[4 * k], lo[4 * k + 1], lo[4 * k + 2], lo[4 * k + 3]);
}

extern "C" int fb_weight_prep(const float* master, int64_t wset_stride_in, int64_t wset_stride_out, int32_t n_wsets, int32_t Cout, int32_t taps,
                              int32_t Cin_real, int32_t Cin_pad, void* w_fwd, void* w_dgrad, int32_t dtype, const float* amax, void* stream) {
    if (!master || !w_fwd) FB_FAIL(FB_ERR_ARG, "fb_weight_prep: null pointer");
    const long long per_set = (long long)Cout * taps * Cin_pad;
    dim3 grid((unsigned)ceil_div64(per_set, 256), n_wsets);
    if (amax) {
        if (dtype != FB_F32 || Cin_pad % 32 != 0 || Cout % 32 != 0) FB_FAIL(FB_ERR_ARG, "fb_weight_prep: fp16x2 planes need fp32 copies and channels in multiples of 32");
        const dim3 gf((unsigned)ceil_div64((long long)Cout * taps * (Cin_pad / 32), 256), n_wsets);
        hipLaunchKernelGGL(weight_prep_planes_fwd_kernel, gf, dim3(256), 0, (hipStream_t)stream, master, (long long)wset_stride_in, (long long)wset_stride_out,
                           Cout, taps, Cin_real, Cin_pad, (_Float16*)w_fwd, amax);
        if (w_dgrad) {
            const dim3 gd((unsigned)ceil_div64((long long)Cin_pad * taps * (Cout / 32), 256), n_wsets);
            hipLaunchKernelGGL(weight_prep_planes_dgrad_kernel, gd, dim3(256), 0, (hipStream_t)stream, master, (long long)wset_stride_in,
                               (long long)wset_stride_out, Cout, taps, Cin_real, Cin_pad, (_Float16*)w_dgrad, amax);
        }
        FB_CHECK_LAUNCH("fb_weight_prep");
        return FB_OK;
    }
    if (dtype == FB_F32)
        hipLaunchKernelGGL((weight_prep_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, master, (long long)wset_stride_in,
                           (long long)wset_stride_out, Cout, taps, Cin_real, Cin_pad, (float*)w_fwd, (float*)w_dgrad);
    else
        hipLaunchKernelGGL((weight_prep_kernel<unsigned short>), grid, dim3(256), 0, (hipStream_t)stream, master,
                           (long long)wset_stride_in, (long long)wset_stride_out, Cout, taps, Cin_real, Cin_pad, (unsigned short*)w_fwd,
                           (unsigned short*)w_dgrad);
    FB_CHECK_LAUNCH("fb_weight_prep");
    return FB_OK;
}
