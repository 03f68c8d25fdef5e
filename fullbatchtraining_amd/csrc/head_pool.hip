// Pooling layers and the classifier head (global average pool, fc, log-softmax, NLL, argmax) with explicit backward.
// These are small, HBM/latency-bound kernels; the loss/accuracy reductions are sequential per chunk so their order is
// fixed (bit-reproducible).
#include "common.h"

// ---- AvgPool2d(2,2) forward (shortcut 'C', reference resnets.py:149); backward is fused into the dgrad epilogue ------
template <typename T>
__global__ void avgpool2_fwd_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int n_img, int H, int W, int cvec) {
    constexpr int V = ET<T>::VEC;
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)n_img * Ho * Wo * cvec;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvec); long long r = i / cvec;
        const int ox = (int)(r % Wo); r /= Wo; const int oy = (int)(r % Ho); const long long n = r / Ho;
        float a[V], acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                ET<T>::unpack(x[(((n * H + 2 * oy + dy) * W) + 2 * ox + dx) * cvec + cv], a);
#pragma unroll
                for (int k = 0; k < V; ++k) acc[k] += a[k];
            }
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] *= 0.25f;
        y[i] = ET<T>::pack(acc);
    }
}

extern "C" int fb_avgpool2_fwd(const void* x, void* y, int32_t n_img, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
    if (!x || !y) FB_FAIL(FB_ERR_ARG, "fb_avgpool2_fwd: null pointer");
    if ((H & 1) || (W & 1)) FB_FAIL(FB_ERR_SHAPE, "fb_avgpool2_fwd: odd spatial size");
    const int V = dtype == FB_F32 ? 4 : 8, cvec = C / V;
    const long long total = (long long)n_img * (H / 2) * (W / 2) * cvec;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == FB_F32) hipLaunchKernelGGL((avgpool2_fwd_kernel<float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, n_img, H, W, cvec);
    else hipLaunchKernelGGL((avgpool2_fwd_kernel<bf16_tag>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, n_img, H, W, cvec);
    FB_CHECK_LAUNCH("fb_avgpool2_fwd");
    return FB_OK;
}

// ---- MaxPool2d(3, stride 2, pad 1) of the 'standard' stem (reference resnets.py:78) ------------------------------------
template <typename T>
__global__ void maxpool3s2_fwd_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int n_img, int H, int W, int cvec) {
    constexpr int V = ET<T>::VEC;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long total = (long long)n_img * Ho * Wo * cvec;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvec); long long r = i / cvec;
        const int ox = (int)(r % Wo); r /= Wo; const int oy = (int)(r % Ho); const long long n = r / Ho;
        float a[V], m[V];
#pragma unroll
        for (int k = 0; k < V; ++k) m[k] = -INFINITY;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int sy = 2 * oy + dy, sx = 2 * ox + dx;
                if ((unsigned)sy >= (unsigned)H || (unsigned)sx >= (unsigned)W) continue;
                ET<T>::unpack(x[((n * H + sy) * W + sx) * cvec + cv], a);
#pragma unroll
                for (int k = 0; k < V; ++k) m[k] = fmaxf(m[k], a[k]);
            }
        y[i] = ET<T>::pack(m);
    }
}
// backward: dx[p] = sum over windows containing p whose argmax (first maximum in row-major window order, torch CPU
// max_pool2d semantics) is p.  One thread owns the 2x2 input quad {2a, 2a+1} x {2b, 2b+1} of one 16-byte channel vector: the quad is
// touched by the four windows (a..a+1) x (b..b+1) only, and every quad pixel sits at a FIXED position of each of them (window (a, b):
// positions 4, 5, 7, 8; (a, b+1): 3, 6; (a+1, b): 1, 2; (a+1, b+1): 0), so a window's argmax is computed once per quad instead of once per
// pixel it covers (the per-pixel gather walked 2.25 windows x 9 loads per pixel with index bookkeeping: 4.2 ms for 1024 images of the
// ImageNet stem, 7x the 0.6 ms its 3.6 GB take at the HBM roofline).  Sums are added in the per-pixel kernel's window order.
// IDX: the type of the flat item index -- 32-bit where the launch has fewer than 2^31 items (three 64-bit divisions by run-time values per item
// are several hundred instructions on this hardware)
template <typename T, typename IDX>
__global__ void maxpool3s2_bwd_kernel(const uint4* __restrict__ x, const uint4* __restrict__ dy, uint4* __restrict__ dx, int n_img, int H,
                                      int W, int cvec) {
    constexpr int V = ET<T>::VEC;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const IDX total = (IDX)n_img * Ho * Wo * cvec;
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
        const int cv = (int)(i % (IDX)cvec); IDX r = i / (IDX)cvec;
        const int b = (int)(r % (IDX)Wo); r /= (IDX)Wo; const int a = (int)(r % (IDX)Ho); const long long n = (long long)(r / (IDX)Ho);
        float acc[4][V];                              // quad pixels (0,0), (0,1), (1,0), (1,1)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < V; ++k) acc[q][k] = 0.f;
#pragma unroll
        for (int wy = 0; wy < 2; ++wy)
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int oy = a + wy, ox = b + wx;
                if (oy >= Ho || ox >= Wo) continue;
                float best[V]; int pos[V];
#pragma unroll
                for (int k = 0; k < V; ++k) { best[k] = -INFINITY; pos[k] = -1; }
#pragma unroll
                for (int dyy = 0; dyy < 3; ++dyy)
#pragma unroll
                    for (int dxx = 0; dxx < 3; ++dxx) {
                        const int sy = 2 * oy + dyy - 1, sx = 2 * ox + dxx - 1;
                        if ((unsigned)sy >= (unsigned)H || (unsigned)sx >= (unsigned)W) continue;
                        float v[V];
                        ET<T>::unpack(x[((n * H + sy) * W + sx) * cvec + cv], v);
#pragma unroll
                        for (int k = 0; k < V; ++k) if (v[k] > best[k] || pos[k] < 0) { best[k] = v[k]; pos[k] = dyy * 3 + dxx; }
                    }
                float g[V];
                ET<T>::unpack(dy[((n * Ho + oy) * Wo + ox) * cvec + cv], g);
                // window position of quad pixel (qy, qx): row 2a + qy - (2 oy - 1) = qy + 1 - 2 wy, column likewise
#pragma unroll
                for (int qy = 0; qy < 2; ++qy)
#pragma unroll
                    for (int qx = 0; qx < 2; ++qx) {
                        const int ry = qy + 1 - 2 * wy, rx = qx + 1 - 2 * wx;
                        if (ry < 0 || rx < 0) continue;
                        const int want = ry * 3 + rx;
#pragma unroll
                        for (int k = 0; k < V; ++k) if (pos[k] == want) acc[qy * 2 + qx][k] += g[k];
                    }
            }
#pragma unroll
        for (int qy = 0; qy < 2; ++qy)
#pragma unroll
            for (int qx = 0; qx < 2; ++qx) {
                const int py = 2 * a + qy, px = 2 * b + qx;
                if (py < H && px < W) dx[((n * H + py) * W + px) * cvec + cv] = ET<T>::pack(acc[qy * 2 + qx]);
            }
    }
}

// ---- the same pooling with the argmax REMEMBERED: the forward pass writes, per output element, the window position (0..8, row-major, first maximum) it
// took its value from -- one byte per element, 1/16 (bf16) of the pooled tensor -- and the backward pass reads that byte instead of finding the maximum of
// every window again.  The recomputing backward reads the whole pre-pool tensor (1.6 GB per 1024 images of the ImageNet stem) and spends ~300 compare /
// select instructions per item: 1.9-2.0 ms at 1.85 TB/s, three times its bytes' worth; this one reads dy + indices and writes dx.  Same bits: the position is
// chosen by the same comparison in the same order.
template <typename T>
__global__ void maxpool3s2_fwd_idx_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, unsigned char* __restrict__ idx, int n_img, int H, int W, int cvec) {
    constexpr int V = ET<T>::VEC;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long total = (long long)n_img * Ho * Wo * cvec;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvec); long long r = i / cvec;
        const int ox = (int)(r % Wo); r /= Wo; const int oy = (int)(r % Ho); const long long n = r / Ho;
        float best[V]; int pos[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { best[k] = -INFINITY; pos[k] = -1; }
#pragma unroll
        for (int dyy = 0; dyy < 3; ++dyy)
#pragma unroll
            for (int dxx = 0; dxx < 3; ++dxx) {
                const int sy = 2 * oy + dyy - 1, sx = 2 * ox + dxx - 1;
                if ((unsigned)sy >= (unsigned)H || (unsigned)sx >= (unsigned)W) continue;
                float v[V];
                ET<T>::unpack(x[((n * H + sy) * W + sx) * cvec + cv], v);
#pragma unroll
                for (int k = 0; k < V; ++k) if (v[k] > best[k] || pos[k] < 0) { best[k] = v[k]; pos[k] = dyy * 3 + dxx; }
            }
        y[i] = ET<T>::pack(best);
        unsigned w[V / 4];
#pragma unroll
        for (int q = 0; q < V / 4; ++q) w[q] = (unsigned)pos[4 * q] | ((unsigned)pos[4 * q + 1] << 8) | ((unsigned)pos[4 * q + 2] << 16) | ((unsigned)pos[4 * q + 3] << 24);
        if constexpr (V == 8) *(uint2*)(idx + i * 8) = make_uint2(w[0], w[1]); else *(unsigned*)(idx + i * 4) = w[0];
    }
}
template <typename T, typename IDX>
__global__ void maxpool3s2_bwd_idx_kernel(const unsigned char* __restrict__ idx, const uint4* __restrict__ dy, uint4* __restrict__ dx, int n_img, int H, int W, int cvec) {
    constexpr int V = ET<T>::VEC;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const IDX total = (IDX)n_img * Ho * Wo * cvec;
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
        const int cv = (int)(i % (IDX)cvec); IDX r = i / (IDX)cvec;
        const int b = (int)(r % (IDX)Wo); r /= (IDX)Wo; const int a = (int)(r % (IDX)Ho); const long long n = (long long)(r / (IDX)Ho);
        float acc[4][V];                              // quad pixels (0,0), (0,1), (1,0), (1,1): see maxpool3s2_bwd_kernel
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < V; ++k) acc[q][k] = 0.f;
#pragma unroll
        for (int wy = 0; wy < 2; ++wy)
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int oy = a + wy, ox = b + wx;
                if (oy >= Ho || ox >= Wo) continue;
                const long long o = ((n * Ho + oy) * Wo + ox) * cvec + cv;
                unsigned w[2];
                if constexpr (V == 8) { const uint2 t = *(const uint2*)(idx + o * 8); w[0] = t.x; w[1] = t.y; } else { w[0] = *(const unsigned*)(idx + o * 4); w[1] = 0; }
                float g[V];
                ET<T>::unpack(dy[o], g);
#pragma unroll
                for (int qy = 0; qy < 2; ++qy)
#pragma unroll
                    for (int qx = 0; qx < 2; ++qx) {
                        const int ry = qy + 1 - 2 * wy, rx = qx + 1 - 2 * wx;
                        if (ry < 0 || rx < 0) continue;
                        const unsigned want = (unsigned)(ry * 3 + rx);
#pragma unroll
                        for (int k = 0; k < V; ++k) if (((w[k >> 2] >> (8 * (k & 3))) & 0xffu) == want) acc[qy * 2 + qx][k] += g[k];
                    }
            }
#pragma unroll
        for (int qy = 0; qy < 2; ++qy)
#pragma unroll
            for (int qx = 0; qx < 2; ++qx) {
                const int py = 2 * a + qy, px = 2 * b + qx;
                if (py < H && px < W) dx[((n * H + py) * W + px) * cvec + cv] = ET<T>::pack(acc[qy * 2 + qx]);
            }
    }
}

extern "C" int fb_maxpool3s2_fwd_idx(const void* x, void* y, void* idx, int32_t n_img, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
    if (!x || !y || !idx) FB_FAIL(FB_ERR_ARG, "fb_maxpool3s2_fwd_idx: null pointer");
    const int V = dtype == FB_F32 ? 4 : 8, cvec = C / V;
    const long long total = (long long)n_img * ((H + 1) / 2) * ((W + 1) / 2) * cvec;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (dtype == FB_F32) hipLaunchKernelGGL((maxpool3s2_fwd_idx_kernel<float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, (unsigned char*)idx, n_img, H, W, cvec);
    else hipLaunchKernelGGL((maxpool3s2_fwd_idx_kernel<bf16_tag>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, (unsigned char*)idx, n_img, H, W, cvec);
    FB_CHECK_LAUNCH("fb_maxpool3s2_fwd_idx");
    return FB_OK;
}
extern "C" int fb_maxpool3s2_bwd_idx(const void* idx, const void* dy, void* dx, int32_t n_img, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
    if (!idx || !dy || !dx) FB_FAIL(FB_ERR_ARG, "fb_maxpool3s2_bwd_idx: null pointer");
    const int V = dtype == FB_F32 ? 4 : 8, cvec = C / V;
    const long long total = (long long)n_img * ((H + 1) / 2) * ((W + 1) / 2) * cvec;        // one thread per 2x2 input quad and channel vector
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    const bool small = total + 65536LL * 256 < (1LL << 32) && !(getenv("FB_MAXPOOL_IDX64") != nullptr);
    if (dtype == FB_F32) {
        if (small) hipLaunchKernelGGL((maxpool3s2_bwd_idx_kernel<float, unsigned>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)idx, (const uint4*)dy, (uint4*)dx, n_img, H, W, cvec);
        else hipLaunchKernelGGL((maxpool3s2_bwd_idx_kernel<float, long long>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)idx, (const uint4*)dy, (uint4*)dx, n_img, H, W, cvec);
    } else {
        if (small) hipLaunchKernelGGL((maxpool3s2_bwd_idx_kernel<bf16_tag, unsigned>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)idx, (const uint4*)dy, (uint4*)dx, n_img, H, W, cvec);
        else hipLaunchKernelGGL((maxpool3s2_bwd_idx_kernel<bf16_tag, long long>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)idx, (const uint4*)dy, (uint4*)dx, n_img, H, W, cvec);
    }
    FB_CHECK_LAUNCH("fb_maxpool3s2_bwd_idx");
    return FB_OK;
}

extern "C" int fb_maxpool3s2_fwd(const void* x, void* y, int32_t n_img, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
    if (!x || !y) FB_FAIL(FB_ERR_ARG, "fb_maxpool3s2_fwd: null pointer");
    const int V = dtype == FB_F32 ? 4 : 8, cvec = C / V;
    const long long total = (long long)n_img * ((H + 1) / 2) * ((W + 1) / 2) * cvec;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (dtype == FB_F32) hipLaunchKernelGGL((maxpool3s2_fwd_kernel<float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, n_img, H, W, cvec);
    else hipLaunchKernelGGL((maxpool3s2_fwd_kernel<bf16_tag>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, n_img, H, W, cvec);
    FB_CHECK_LAUNCH("fb_maxpool3s2_fwd");
    return FB_OK;
}
extern "C" int fb_maxpool3s2_bwd(const void* x, const void* dy, void* dx, int32_t n_img, int32_t H, int32_t W, int32_t C, int32_t dtype,
                                 void* stream) {
    if (!x || !dy || !dx) FB_FAIL(FB_ERR_ARG, "fb_maxpool3s2_bwd: null pointer");
    const int V = dtype == FB_F32 ? 4 : 8, cvec = C / V;
    const long long total = (long long)n_img * ((H + 1) / 2) * ((W + 1) / 2) * cvec;        // one thread per 2x2 input quad and channel vector
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    const bool small = total + 65536LL * 256 < (1LL << 32) && !(getenv("FB_MAXPOOL_IDX64") != nullptr);
    if (dtype == FB_F32) {
        if (small) hipLaunchKernelGGL((maxpool3s2_bwd_kernel<float, unsigned>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (const uint4*)dy, (uint4*)dx, n_img, H, W, cvec);
        else hipLaunchKernelGGL((maxpool3s2_bwd_kernel<float, long long>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (const uint4*)dy, (uint4*)dx, n_img, H, W, cvec);
    } else {
        if (small) hipLaunchKernelGGL((maxpool3s2_bwd_kernel<bf16_tag, unsigned>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (const uint4*)dy, (uint4*)dx, n_img, H, W, cvec);
        else hipLaunchKernelGGL((maxpool3s2_bwd_kernel<bf16_tag, long long>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (const uint4*)dy, (uint4*)dx, n_img, H, W, cvec);
    }
    FB_CHECK_LAUNCH("fb_maxpool3s2_bwd");
    return FB_OK;
}

// ---- global average pool -> fp32 features ------------------------------------------------------------------------------
template <typename T>
__global__ void head_pool_kernel(const uint4* __restrict__ a, float* __restrict__ feat, int n_img, int HW, int cvec) {
    constexpr int V = ET<T>::VEC;
    const long long total = (long long)n_img * cvec;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cv = (int)(i % cvec); const long long n = i / cvec;
    float acc[V], v[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    for (int p = 0; p < HW; ++p) {
        ET<T>::unpack(a[(n * HW + p) * cvec + cv], v);
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] += v[k];
    }
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int k = 0; k < V; ++k) feat[n * (long long)cvec * V + cv * V + k] = acc[k] * inv;
}

extern "C" int fb_head_pool(const void* a, float* feat, int32_t n_img, int32_t HW, int32_t C, int32_t dtype, void* stream) {
    if (!a || !feat) FB_FAIL(FB_ERR_ARG, "fb_head_pool: null pointer");
    const int V = dtype == FB_F32 ? 4 : 8, cvec = C / V;
    const long long total = (long long)n_img * cvec;
    if (dtype == FB_F32) hipLaunchKernelGGL((head_pool_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)a, feat, n_img, HW, cvec);
    else hipLaunchKernelGGL((head_pool_kernel<bf16_tag>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)a, feat, n_img, HW, cvec);
    FB_CHECK_LAUNCH("fb_head_pool");
    return FB_OK;
}

// ---- fc + log-softmax + NLL(mean) + argmax ------------------------------------------------------------------------------
// grid (imgs_per_group, n_groups) for the logits; one wave per image computes all classes.
__global__ void head_logits_kernel(const float* __restrict__ feat, const float* __restrict__ W, const float* __restrict__ b, long long pstride,
                                   float* __restrict__ logits, int ipg, int C, int classes) {
    const int n = blockIdx.x, g = blockIdx.y, lane = threadIdx.x;
    const float* f = feat + ((long long)g * ipg + n) * C;
    const float* Wg = W + (long long)g * pstride; const float* bg = b + (long long)g * pstride;
    for (int j = 0; j < classes; ++j) {
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc += f[c] * Wg[(long long)j * C + c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) logits[((long long)g * ipg + n) * classes + j] = acc + bg[j];
    }
}
// one block per group: per-image log-softmax (thread per image), then a fixed-order sum of the per-image losses.
// smoothing: label smoothing of LabelSmoothCrossEntropyLoss (reference modules.py:86-101: weight 1-s on the target, s/(C-1) elsewhere);
// only_incorrect: IncorrectCrossEntropyLoss (modules.py:104-119: already correct samples contribute no loss and no gradient).
__global__ void head_loss_kernel(const float* __restrict__ logits, const long long* __restrict__ labels, float* __restrict__ dlogits,
                                 float* __restrict__ loss, float* __restrict__ correct, int ipg, int classes, float smoothing,
                                 int only_incorrect) {
    extern __shared__ float sm[];   // [ipg] losses, [ipg] corrects
    const int g = blockIdx.x;
    int valid = 0;                  // rows with a negative label are padding of a ragged chunk
    for (int n = 0; n < ipg; ++n) valid += labels[(long long)g * ipg + n] >= 0 ? 1 : 0;
    for (int n = threadIdx.x; n < ipg; n += blockDim.x) {
        const float* z = logits + ((long long)g * ipg + n) * classes;
        float* dz = dlogits + ((long long)g * ipg + n) * classes;
        const int label = (int)labels[(long long)g * ipg + n];
        if (label < 0) {
            for (int j = 0; j < classes; ++j) dz[j] = 0.f;
            sm[n] = 0.f; sm[ipg + n] = 0.f;
            continue;
        }
        float m = z[0]; int am = 0;
        for (int j = 1; j < classes; ++j) if (z[j] > m) { m = z[j]; am = j; }   // first maximum, as torch.argmax
        float se = 0.f;
        for (int j = 0; j < classes; ++j) se += expf(z[j] - m);
        const float lse = logf(se);
        const float inv_n = 1.f / (float)valid;
        const float keep = (only_incorrect && am == label) ? 0.f : 1.f;
        const float w_t = 1.f - smoothing, w_o = smoothing / ((float)classes - 1.f);
        float li = 0.f;
        for (int j = 0; j < classes; ++j) {
            const float logp = z[j] - m - lse, w = j == label ? w_t : w_o;
            dz[j] = (expf(logp) - w) * inv_n * keep;
            if (w != 0.f) li -= w * logp;
        }
        sm[n] = li * keep;
        sm[ipg + n] = (am == label) ? 1.f : 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float l = 0.f, c = 0.f;
        for (int n = 0; n < ipg; ++n) { l += sm[n]; c += sm[ipg + n]; }
        loss[g] = l / (float)valid; correct[g] = c;
    }
}

extern "C" int fb_head_loss(const float* feat, const float* fc_w, const float* fc_b, int64_t param_group_stride, const int64_t* labels,
                            float* logits, float* dlogits, float* loss, float* correct, int32_t n_groups, int32_t imgs_per_group,
                            int32_t C, int32_t classes, float label_smoothing, int32_t only_incorrect, void* stream) {
    if (!feat || !fc_w || !fc_b || !labels || !logits || !dlogits || !loss || !correct) FB_FAIL(FB_ERR_ARG, "fb_head_loss: null pointer");
    if (label_smoothing < 0.f || label_smoothing >= 1.f || classes < 2) FB_FAIL(FB_ERR_ARG, "fb_head_loss: label_smoothing=%g, classes=%d", (double)label_smoothing, classes);
    hipLaunchKernelGGL(head_logits_kernel, dim3(imgs_per_group, n_groups), dim3(64), 0, (hipStream_t)stream, feat, fc_w, fc_b,
                       (long long)param_group_stride, logits, imgs_per_group, C, classes);
    hipLaunchKernelGGL(head_loss_kernel, dim3(n_groups), dim3(128), (size_t)2 * imgs_per_group * sizeof(float), (hipStream_t)stream, logits,
                       (const long long*)labels, dlogits, loss, correct, imgs_per_group, classes, label_smoothing, only_incorrect);
    FB_CHECK_LAUNCH("fb_head_loss");
    return FB_OK;
}

// ---- evaluation helpers ------------------------------------------------------------------------------------------------
__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                      const float* __restrict__ rv, float eps, float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * (1.f / sqrtf(rv[c] + eps));
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}
extern "C" int fb_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                                 float* scale, float* shift, int32_t C, void* stream) {
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift) FB_FAIL(FB_ERR_ARG, "fb_bn_eval_coeffs: null pointer");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, eps,
                       scale, shift, C);
    FB_CHECK_LAUNCH("fb_bn_eval_coeffs");
    return FB_OK;
}

// one workgroup: thread per image, then a fixed-order sum (thread 0)
__global__ void head_tta_kernel(const float* __restrict__ za, const float* __restrict__ zb, const long long* __restrict__ labels, int n_img,
                                int classes, float* __restrict__ ws) {
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < n_img; n += gridDim.x * blockDim.x) {
        const float* a = za + (long long)n * classes; const float* b = zb + (long long)n * classes;
        float ma = a[0], mb = b[0];
        for (int j = 1; j < classes; ++j) { ma = fmaxf(ma, a[j]); mb = fmaxf(mb, b[j]); }
        float sa = 0.f, sb = 0.f;
        for (int j = 0; j < classes; ++j) { sa += expf(a[j] - ma); sb += expf(b[j] - mb); }
        // outputs_j = softmax(a)_j + softmax(b)_j in (0, 2]; CE(outputs) = logsumexp(outputs) - outputs_label
        const int label = (int)labels[n];
        float best = -1.f, se = 0.f, ol = 0.f; int am = 0;
        for (int j = 0; j < classes; ++j) {
            const float o = expf(a[j] - ma) / sa + expf(b[j] - mb) / sb;
            if (o > best) { best = o; am = j; }
            se += expf(o);
            if (j == label) ol = o;
        }
        ws[n] = logf(se) - ol;
        ws[n_img + n] = am == label ? 1.f : 0.f;
    }
}
__global__ void head_tta_sum_kernel(const float* __restrict__ ws, int n_img, float* __restrict__ loss_sum, float* __restrict__ correct) {
    __shared__ double sl[256], sc[256];
    double l = 0.0, c = 0.0;
    for (int n = threadIdx.x; n < n_img; n += 256) { l += (double)ws[n]; c += (double)ws[n_img + n]; }
    sl[threadIdx.x] = l; sc[threadIdx.x] = c;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) { sl[threadIdx.x] += sl[threadIdx.x + h]; sc[threadIdx.x] += sc[threadIdx.x + h]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { loss_sum[0] = (float)sl[0]; correct[0] = (float)sc[0]; }
}
// ws: 2*n floats of caller-provided scratch (per-image loss terms and hits).
extern "C" int fb_head_tta(const float* logits_a, const float* logits_b, const int64_t* labels, int32_t n, int32_t classes, float* ws,
                           float* loss_sum, float* correct, void* stream) {
    if (!logits_a || !logits_b || !labels || !ws || !loss_sum || !correct) FB_FAIL(FB_ERR_ARG, "fb_head_tta: null pointer");
    hipLaunchKernelGGL(head_tta_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits_a, logits_b, (const long long*)labels, n,
                       classes, ws);
    hipLaunchKernelGGL(head_tta_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, n, loss_sum, correct);
    FB_CHECK_LAUNCH("fb_head_tta");
    return FB_OK;
}

// ---- head backward ---------------------------------------------------------------------------------------------------
// dW[g][j][c] = sum_n dlogits[n][j]*feat[n][c] (fixed order over n); db[g][j] = sum_n dlogits[n][j]
__global__ void head_bwd_w_kernel(const float* __restrict__ feat, const float* __restrict__ dlogits, float* __restrict__ dW, float* __restrict__ db,
                                  long long gstride, int ipg, int C, int classes) {
    const int g = blockIdx.y;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over classes*C
    if (idx < (long long)classes * C) {
        const int j = (int)(idx / C), c = (int)(idx % C);
        float acc = 0.f;
        for (int n = 0; n < ipg; ++n) acc += dlogits[((long long)g * ipg + n) * classes + j] * feat[((long long)g * ipg + n) * C + c];
        dW[(long long)g * gstride + idx] = acc;
    }
    if (blockIdx.x == 0 && threadIdx.x < classes) {
        for (int j = threadIdx.x; j < classes; j += blockDim.x) {
            float acc = 0.f;
            for (int n = 0; n < ipg; ++n) acc += dlogits[((long long)g * ipg + n) * classes + j];
            db[(long long)g * gstride + j] = acc;
        }
    }
}
template <typename T>
__global__ void head_bwd_a_kernel(const float* __restrict__ dlogits, const float* __restrict__ W, long long pstride, uint4* __restrict__ d_a,
                                  int ipg, int HW, int C, int classes) {
    constexpr int V = ET<T>::VEC;
    const int cvec = C / V;
    const long long n = blockIdx.x;   // image
    const int g = (int)(n / ipg);
    const float* Wg = W + (long long)g * pstride;
    const float* dz = dlogits + n * classes;
    for (int cv = threadIdx.x; cv < cvec; cv += blockDim.x) {
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        for (int j = 0; j < classes; ++j) {
            const float d = dz[j];
#pragma unroll
            for (int k = 0; k < V; ++k) acc[k] += d * Wg[(long long)j * C + cv * V + k];
        }
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] *= inv;
        const uint4 o = ET<T>::pack(acc);
        for (int p = 0; p < HW; ++p) d_a[(n * HW + p) * cvec + cv] = o;
    }
}

extern "C" int fb_head_bwd(const float* feat, const float* dlogits, const float* fc_w, int64_t param_group_stride, float* dfc_w, float* dfc_b,
                           int64_t grad_group_stride, void* d_a, int32_t n_groups, int32_t imgs_per_group, int32_t HW, int32_t C,
                           int32_t classes, int32_t dtype, void* stream) {
    if (!feat || !dlogits || !fc_w || !dfc_w || !dfc_b || !d_a) FB_FAIL(FB_ERR_ARG, "fb_head_bwd: null pointer");
    const long long total = (long long)classes * C;
    hipLaunchKernelGGL(head_bwd_w_kernel, dim3((unsigned)((total + 255) / 256), n_groups), dim3(256), 0, (hipStream_t)stream, feat, dlogits, dfc_w,
                       dfc_b, (long long)grad_group_stride, imgs_per_group, C, classes);
    const int n_img = n_groups * imgs_per_group;
    if (dtype == FB_F32) hipLaunchKernelGGL((head_bwd_a_kernel<float>), dim3(n_img), dim3(128), 0, (hipStream_t)stream, dlogits, fc_w, (long long)param_group_stride, (uint4*)d_a, imgs_per_group, HW, C, classes);
    else hipLaunchKernelGGL((head_bwd_a_kernel<bf16_tag>), dim3(n_img), dim3(128), 0, (hipStream_t)stream, dlogits, fc_w, (long long)param_group_stride, (uint4*)d_a, imgs_per_group, HW, C, classes);
    FB_CHECK_LAUNCH("fb_head_bwd");
    return FB_OK;
}
