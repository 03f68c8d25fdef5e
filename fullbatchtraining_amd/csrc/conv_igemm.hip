// Implicit-GEMM convolution (forward and input-gradient) on MFMA for gfx950.
//
//   D[co][pixel] += sum_{tap, ci} W[co][tap][ci] * SRC[src_pixel(pixel, tap)][ci]
//
// MFMA "A" operand = weight rows (co), "B" operand = gathered pixel rows, both K(ci)-contiguous 128-byte rows in LDS,
// so the accumulator fragment holds 4 consecutive output channels of one pixel per lane -> 8/16-byte NHWC stores.
// Block tile: 128 pixels x BN_CO channels, 256 threads (2x2 waves), K-step = 128 bytes of channels of one tap
// (64 bf16 / 32 f32), register-prefetched global loads + double-buffered LDS, XOR-swizzled 16-byte chunks.
// Zero padding and stride-2 dgrad parity classes are resolved in the gather (invalid rows load zeros; taps that are
// invalid for a whole parity class are skipped, so no MFMA work is wasted on structural zeros).
#include "common.h"
#include "profile.h"

#include "conv_params.h"

template <typename T, int BN_CO>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    constexpr int EB = ET<T>::EB;
    constexpr int BKE = 128 / EB;           // channels per K-step
    constexpr int WROWS = BN_CO / 32;       // weight rows per thread
    constexpr int FI = BN_CO / 32;          // co fragments per wave (2 waves along co)
    constexpr int FJ = 4;                   // pixel fragments per wave (2 waves along pixels, 64 pixels each)
    constexpr int TILE_CHUNKS = (128 + BN_CO) * 8;
    __shared__ uint4 lds[2 * TILE_CHUNKS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_co = wave >> 1, wave_px = wave & 1;
    const int mblk = blockIdx.x, co_blk = blockIdx.y, cls = blockIdx.z;
    const int cpy = (p.os == 2) ? (cls >> 1) : 0, cpx = (p.os == 2) ? (cls & 1) : 0;
    const int chunk = tid & 7, lrow = tid >> 3;

    // ---- per-thread gather rows -------------------------------------------------------------------------------------
    int a_img[4], a_y[4], a_x[4];
    const int qHW = p.qH * p.qW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = mblk * 128 + lrow + 32 * i;
        if (m < p.M) {
            const int n = m / qHW, rem = m - n * qHW, qy = rem / p.qW, qx = rem - qy * p.qW;
            a_img[i] = n * p.Hs * p.Ws; a_y[i] = qy * p.ss; a_x[i] = qx * p.ss;
        } else {
            a_img[i] = 0; a_y[i] = -(1 << 28); a_x[i] = 0;   // always out of range -> zeros
        }
    }
    const int first_img = (mblk * 128) / qHW;
    const char* wbase = p.wgt + (long long)(first_img / p.imgs_per_wset) * p.wset_stride_bytes;
    const int taps = p.R * p.S;
    const long long wrow_bytes = (long long)taps * p.Cs * EB;
    const char* wptr[WROWS];
#pragma unroll
    for (int i = 0; i < WROWS; ++i) wptr[i] = wbase + (long long)(co_blk * BN_CO + lrow + 32 * i) * wrow_bytes + chunk * 16;

    // ---- tap enumeration (block-uniform) ----------------------------------------------------------------------------
    auto tap_valid = [&](int t, int& dy, int& dx) -> bool {
        const int r = t / p.S, s = t - r * p.S;
        if (p.mode == 0) { dy = r - p.pad; dx = s - p.pad; return true; }
        if (p.os == 1) { dy = p.pad - r; dx = p.pad - s; return true; }
        const int vy = cpy + p.pad - r, vx = cpx + p.pad - s;
        if ((vy & 1) || (vx & 1)) return false;
        dy = vy >> 1; dx = vx >> 1;
        return true;
    };
    const int kc = (p.Cs + BKE - 1) / BKE;
    int n_valid = 0;
    for (int t = 0; t < taps; ++t) { int dy, dx; n_valid += tap_valid(t, dy, dx) ? 1 : 0; }
    const int n_iter = n_valid * kc;

    uint4 ra[4], rw[WROWS];
    int cur_t = -1, cur_c = kc, cur_dy = 0, cur_dx = 0;   // iteration cursor
    auto advance = [&]() {
        if (++cur_c >= kc) {
            cur_c = 0;
            do { ++cur_t; } while (cur_t < taps && !tap_valid(cur_t, cur_dy, cur_dx));
        }
    };
    auto gload = [&]() {
        const int c0 = cur_c * BKE;
        const bool cvalid = c0 + chunk * ET<T>::VEC < p.Cs;   // Cs may be a multiple of 32 only (bf16 half K-step)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sy = a_y[i] + cur_dy, sx = a_x[i] + cur_dx;
            const bool ok = cvalid && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
            ra[i] = make_uint4(0, 0, 0, 0);
            if (ok) ra[i] = *(const uint4*)(p.src + ((long long)(a_img[i] + sy * p.Ws + sx) * p.Cs + c0) * EB + chunk * 16);
        }
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            rw[i] = make_uint4(0, 0, 0, 0);
            if (cvalid) rw[i] = *(const uint4*)(wptr[i] + ((long long)cur_t * p.Cs + c0) * EB);
        }
    };
    auto lstore = [&](int buf) {
        uint4* base = lds + buf * TILE_CHUNKS;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int row = lrow + 32 * i; base[row * 8 + (chunk ^ (row & 7))] = ra[i]; }
#pragma unroll
        for (int i = 0; i < WROWS; ++i) { const int row = 128 + lrow + 32 * i; base[row * 8 + (chunk ^ (row & 7))] = rw[i]; }
    };

    f32x4_t acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    if (n_iter > 0) {
        advance(); gload(); lstore(0);
        __syncthreads();
        for (int it = 0; it < n_iter; ++it) {
            const int buf = it & 1;
            const bool more = it + 1 < n_iter;
            if (more) { advance(); gload(); }
            const uint4* base = lds + buf * TILE_CHUNKS;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pc = ((lane >> 4) + 4 * h) ^ (lane & 7);
                uint4 wf[FI], pf[FJ];
#pragma unroll
                for (int i = 0; i < FI; ++i) wf[i] = base[(128 + wave_co * (BN_CO / 2) + i * 16 + (lane & 15)) * 8 + pc];
#pragma unroll
                for (int j = 0; j < FJ; ++j) pf[j] = base[(wave_px * 64 + j * 16 + (lane & 15)) * 8 + pc];
#pragma unroll
                for (int i = 0; i < FI; ++i)
#pragma unroll
                    for (int j = 0; j < FJ; ++j) acc[i][j] = mma_chunk<T>(wf[i], pf[j], acc[i][j]);
            }
            if (more) lstore(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: (+addend) -> dst, per-channel partial statistics ---------------------------------------------------
    float ssum[FI][4], ssq[FI][4];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
        const int m = mblk * 128 + wave_px * 64 + j * 16 + (lane & 15);
        if (m < p.M) {
            const int n = m / qHW, rem = m - n * qHW, qy = rem / p.qW, qx = rem - qy * p.qW;
            const int oy = qy * p.os + cpy, ox = qx * p.os + cpx;
            const long long pix = ((long long)n * p.Hd + oy) * p.Wd + ox;
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int co = co_blk * BN_CO + wave_co * (BN_CO / 2) + i * 16 + (lane >> 4) * 4;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (p.addend_mode == 1) {
                    const char* ap = p.addend + (pix * p.Cd + co) * EB;
                    if constexpr (EB == 4) { const float4 a = *(const float4*)ap; v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; }
                    else { const uint2 a = *(const uint2*)ap; v[0] += __uint_as_float(a.x << 16); v[1] += __uint_as_float(a.x & 0xffff0000u);
                           v[2] += __uint_as_float(a.y << 16); v[3] += __uint_as_float(a.y & 0xffff0000u); }
                } else if (p.addend_mode == 2) {
                    const long long apix = ((long long)n * (p.Hd >> 1) + (oy >> 1)) * (p.Wd >> 1) + (ox >> 1);
                    const char* ap = p.addend + (apix * p.Cd + co) * EB;
                    if constexpr (EB == 4) { const float4 a = *(const float4*)ap; v[0] += 0.25f * a.x; v[1] += 0.25f * a.y; v[2] += 0.25f * a.z; v[3] += 0.25f * a.w; }
                    else { const uint2 a = *(const uint2*)ap; v[0] += 0.25f * __uint_as_float(a.x << 16); v[1] += 0.25f * __uint_as_float(a.x & 0xffff0000u);
                           v[2] += 0.25f * __uint_as_float(a.y << 16); v[3] += 0.25f * __uint_as_float(a.y & 0xffff0000u); }
                }
                char* dp = p.dst + (pix * p.Cd + co) * EB;
                if constexpr (EB == 4) *(float4*)dp = make_float4(v[0], v[1], v[2], v[3]);
                else *(uint2*)dp = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
#pragma unroll
                for (int r = 0; r < 4; ++r) { ssum[i][r] += v[r]; ssq[i][r] += v[r] * v[r]; }
            }
        }
    }
    if (p.stat != nullptr) {
        // reduce over the 16 pixel lanes (lane&15), then over the two pixel-waves through LDS; fixed order -> deterministic
        float* red = (float*)lds;   // [2 px-waves][BN_CO][2]
        __syncthreads();
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = ssum[i][r], b = ssq[i][r];
                a = row16_sum(a); b = row16_sum(b);
                if ((lane & 15) == 0) {
                    const int col = wave_co * (BN_CO / 2) + i * 16 + (lane >> 4) * 4 + r;
                    red[(wave_px * BN_CO + col) * 2] = a; red[(wave_px * BN_CO + col) * 2 + 1] = b;
                }
            }
        __syncthreads();
        if (tid < BN_CO) {
            const float a = red[tid * 2] + red[(BN_CO + tid) * 2];
            const float b = red[tid * 2 + 1] + red[(BN_CO + tid) * 2 + 1];
            const long long blk = (long long)cls * gridDim.x + mblk;
            p.stat[blk * p.Cd + co_blk * BN_CO + tid] = a;
            p.stat[((long long)p.n_mblocks + blk) * p.Cd + co_blk * BN_CO + tid] = b;
        }
    }
}

int fb_try_conv3x3_halo4(const fb_conv_args* a, hipStream_t st);   // conv3x3_halo4.hip
int fb_try_conv3x3_halo5(const fb_conv_args* a, hipStream_t st);   // conv3x3_halo5.hip
int fb_conv3x3_halo5_takes(const fb_conv_args* a);
int fb_try_conv3x3s2_dgrad_quad(const fb_conv_args* a, hipStream_t st);   // conv3x3s2_dgrad_quad.hip
int fb_try_conv1x1_k32(const fb_conv_args* a, hipStream_t st);     // conv1x1_k32.hip (the stem on pre-gathered patches)
int fb_try_conv1x1_stream(const fb_conv_args* a, hipStream_t st);  // conv1x1_stream.hip (short-K 1x1 convolutions)
int fb_try_conv1x1_gemm(const fb_conv_args* a, hipStream_t st);    // conv1x1_gemm.hip (K >= 512: persistent tiles, three-stage LDS ring)
int fb_try_conv1x1_pipe(const fb_conv_args* a, hipStream_t st);    // conv1x1_pipe.hip (the same with the epilogue threaded through the next group's MFMAs)

// 1 if fb_conv2d accepts `addend_mask` for these arguments (only the resident-filter 64-channel kernel applies the mask so far)
int fb_conv1x1_pipe_takes(const fb_conv_args* a);     // conv1x1_pipe.hip
int fb_conv3x3_halo4_takes(const fb_conv_args* a);    // conv3x3_halo4.hip
int fb_conv1x1_gemm_takes(const fb_conv_args* a);     // conv1x1_gemm.hip
// (round 6: + the implicit GEMM, for every stride-1 input gradient with a same-shape addend that none of the specialised kernels takes -- all Bottleneck identity
// blocks in fp32 storage, the 512-channel ones in bf16; FB_IGEMM_NO_MASK: the engine materialises d * (out > 0) for those again)
static int igemm_masked_addend_ok(const fb_conv_args* a) {
    if (getenv("FB_IGEMM_NO_MASK") != nullptr || a->mode != 1 || a->stride != 1 || a->bst_x) return 0;
    if (a->Cs % 32 != 0 || a->Cd % 64 != 0) return 0;
    static const bool v1 = getenv("FB_IGEMM_V1") != nullptr;
    if (v1) return 0;
    ConvParams p{};
    p.n_img = a->n_img; p.Hs = a->Hs; p.Ws = a->Ws; p.Cs = a->Cs; p.Hd = a->Hd; p.Wd = a->Wd; p.Cd = a->Cd; p.R = a->R; p.S = a->S;
    p.qH = a->Hd; p.qW = a->Wd;
    return fb_igemm_glds_fits(p, a->dtype);
}
extern "C" int32_t fb_conv_masked_addend_supported(const fb_conv_args* a) {
    if (!a || !a->addend || !a->addend_mask || a->addend_mode != 1) return 0;
    return fb_conv3x3_halo5_takes(a) || fb_conv1x1_pipe_takes(a) || fb_conv3x3_halo4_takes(a) || fb_conv1x1_gemm_takes(a) || igemm_masked_addend_ok(a);
}

// 1 if fb_conv2d implements the fused BatchNorm-backward reduction for these arguments: the resident-filter and the persistent halo kernels
extern "C" int32_t fb_conv_bwd_stat_supported(const fb_conv_args* a) {
    static const bool disabled = getenv("FB_DISABLE_FUSED_BWD_STAT") != nullptr;
    if (disabled || !a || !a->bst_x || !a->bst_mask || !a->stat_partial || a->mode != 1 || a->dtype != FB_BF16) return 0;
    if ((long long)a->n_img * a->Hd * a->Wd % 128 != 0) return 0;
    static const char* widths = getenv("FB_FUSED_BWD_STAT_W");          // A/B switch: list of map widths, e.g. "4,8"
    if (widths) {
        char tag[8];
        snprintf(tag, sizeof(tag), "%d,", a->Wd);
        char list[64];
        snprintf(list, sizeof(list), "%s,", widths);
        if (!strstr(list, tag) || (a->Wd < 10 && list != strstr(list, tag) && *(strstr(list, tag) - 1) != ',')) return 0;
    }
    return fb_conv3x3_halo5_takes(a) || fb_conv3x3_halo4_takes(a);
}

template <typename T> static int launch_conv(const ConvParams& p, int classes, hipStream_t st) {
    const int mblocks = (p.M + 127) / 128;
    if (p.Cd % 128 == 0 && (long long)mblocks * (p.Cd / 128) * classes >= 512) {
        dim3 grid(mblocks, p.Cd / 128, classes);
        hipLaunchKernelGGL((conv_igemm_kernel<T, 128>), grid, dim3(256), 0, st, p);
    } else {
        dim3 grid(mblocks, p.Cd / 64, classes);
        hipLaunchKernelGGL((conv_igemm_kernel<T, 64>), grid, dim3(256), 0, st, p);
    }
    return 0;
}

extern "C" int fb_conv2d(const fb_conv_args* a, void* stream) {
    if (!a || !a->src || !a->wgt || !a->dst) FB_FAIL(FB_ERR_ARG, "fb_conv2d: null pointer");
    const int EB = a->dtype == FB_F32 ? 4 : 2;
    if (a->Cs % 32 != 0) FB_FAIL(FB_ERR_SHAPE, "fb_conv2d: Cs=%d must be a multiple of 32", a->Cs);
    if (a->Cd % 64 != 0) FB_FAIL(FB_ERR_SHAPE, "fb_conv2d: Cd=%d must be a multiple of 64", a->Cd);
    if (a->mode != 0 && a->mode != 1) FB_FAIL(FB_ERR_ARG, "fb_conv2d: mode %d", a->mode);
    if (a->addend_mask && !fb_conv_masked_addend_supported(a))
        FB_FAIL(FB_ERR_UNSUPPORTED, "fb_conv2d: addend_mask is not implemented for this shape (ask fb_conv_masked_addend_supported)");
    if (a->bst_x && !fb_conv_bwd_stat_supported(a))
        FB_FAIL(FB_ERR_UNSUPPORTED, "fb_conv2d: the fused BatchNorm-backward reduction is not implemented for this shape (ask fb_conv_bwd_stat_supported)");
    if (a->stride != 1 && a->stride != 2) FB_FAIL(FB_ERR_UNSUPPORTED, "fb_conv2d: stride %d", a->stride);
    ConvParams p;
    p.src = (const char*)a->src; p.wgt = (const char*)a->wgt; p.dst = (char*)a->dst; p.addend = (const char*)a->addend;
    p.stat = a->stat_partial;
    p.n_img = a->n_img; p.Hs = a->Hs; p.Ws = a->Ws; p.Cs = a->Cs; p.Hd = a->Hd; p.Wd = a->Wd; p.Cd = a->Cd;
    p.R = a->R; p.S = a->S; p.stride = a->stride; p.pad = a->pad; p.mode = a->mode;
    p.imgs_per_wset = a->imgs_per_wset > 0 ? a->imgs_per_wset : a->n_img;
    p.wset_stride_bytes = a->wset_stride * EB;
    p.addend_mode = a->addend ? a->addend_mode : 0;
    p.addend_mask = a->addend ? (const unsigned char*)a->addend_mask : nullptr;
    p.amax_src = a->dtype == FB_F32 ? a->amax_src : nullptr; p.amax_wgt = a->dtype == FB_F32 ? a->amax_wgt : nullptr;
    if ((p.amax_src == nullptr) != (p.amax_wgt == nullptr)) FB_FAIL(FB_ERR_ARG, "fb_conv2d: amax_src and amax_wgt go together");
    p.amax_imgs = a->amax_imgs > 0 ? a->amax_imgs : a->n_img;
    int classes = 1;
    if (a->mode == 0) {
        if (a->Hd != (a->Hs + 2 * a->pad - a->R) / a->stride + 1 || a->Wd != (a->Ws + 2 * a->pad - a->S) / a->stride + 1)
            FB_FAIL(FB_ERR_SHAPE, "fb_conv2d: fwd output %dx%d inconsistent with input %dx%d", a->Hd, a->Wd, a->Hs, a->Ws);
        p.qH = a->Hd; p.qW = a->Wd; p.os = 1; p.ss = a->stride;
    } else {
        // dst = d_input [Hd x Wd], src = d_output [Hs x Ws]
        if (a->Hs != (a->Hd + 2 * a->pad - a->R) / a->stride + 1 || a->Ws != (a->Wd + 2 * a->pad - a->S) / a->stride + 1)
            FB_FAIL(FB_ERR_SHAPE, "fb_conv2d: dgrad shapes inconsistent");
        if (a->stat_partial && !a->bst_x) FB_FAIL(FB_ERR_ARG, "fb_conv2d: statistics in mode 1 need bst_x / bst_mask");
        if (a->stride == 1) { p.qH = a->Hd; p.qW = a->Wd; p.os = 1; }
        else {
            if ((a->Hd & 1) || (a->Wd & 1)) FB_FAIL(FB_ERR_SHAPE, "fb_conv2d: stride-2 dgrad needs even input dims");
            p.qH = a->Hd / 2; p.qW = a->Wd / 2; p.os = 2; classes = 4;
        }
        p.ss = 1;
    }
    p.M = a->n_img * p.qH * p.qW;
    p.n_mblocks = ((p.M + 127) / 128) * classes;
    hipStream_t st = (hipStream_t)stream;
    const int32_t info[FB_PROF_INFO] = {a->n_img, a->Hs, a->Ws, a->Cs, a->Hd, a->Wd, a->Cd, a->R, a->stride,
                                        (a->addend ? a->addend_mode : 0) | (a->addend_mask ? 4 : 0) | (a->bst_x ? 8 : 0) | (a->dtype << 4), 0};
    const int prof = fb_prof_begin(a->mode == 0 ? FB_PROF_IGEMM_FWD : FB_PROF_IGEMM_DGRAD, st, info);
    static const bool v1 = getenv("FB_IGEMM_V1") != nullptr;
    int kernel = 0;
    if (fb_try_conv1x1_k32(a, st)) kernel = FB_K_CONV1X1_K32;
    else if ((a->mode == 0 || a->addend) && (a->Cs == 256 || a->Cs == 128) && fb_try_conv1x1_gemm(a, st)) kernel = FB_K_CONV1X1_GEMM;      // (mode 1: only with FB_C1G=2)      // (large forward calls with K = 128 / 256: conv1x1_gemm.hip)
    else if (fb_try_conv1x1_pipe(a, st)) kernel = FB_K_CONV1X1_PIPE;
    else if (fb_try_conv1x1_stream(a, st)) kernel = FB_K_CONV1X1_STREAM;
    else if (fb_try_conv1x1_gemm(a, st)) kernel = FB_K_CONV1X1_GEMM;
    else if (fb_try_conv3x3s2_dgrad_quad(a, st)) kernel = FB_K_S2_DGRAD_QUAD;
    else if (fb_try_conv3x3_halo5(a, st)) kernel = FB_K_HALO5;
    else if (fb_try_conv3x3_halo4(a, st)) kernel = FB_K_HALO4;
    else {
        p.zeros = nullptr;
        kernel = FB_K_IGEMM_GLDS;
        if (v1 || !fb_launch_igemm_glds(p, classes, a->dtype, st)) {
            if (p.addend_mask) FB_FAIL(FB_ERR_UNSUPPORTED, "fb_conv2d: the register-staged implicit GEMM does not apply addend_mask (tensor beyond 2^31 bytes per tile or FB_IGEMM_V1)");
            if (p.amax_src) FB_FAIL(FB_ERR_UNSUPPORTED, "fb_conv2d: the register-staged implicit GEMM has no fp16x2 path (tensor beyond 2^31 bytes or FB_IGEMM_V1)");
            if (a->dtype == FB_F32) launch_conv<float>(p, classes, st); else launch_conv<bf16_tag>(p, classes, st);
            kernel = FB_K_IGEMM_V1;
        }
    }
    fb_prof_kernel(prof, kernel);
    fb_prof_end(prof, st);
    FB_CHECK_LAUNCH("fb_conv2d");
    return FB_OK;
}
