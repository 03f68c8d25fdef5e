// Stem patch gather (+ on-device RandomCrop / RandomHorizontalFlip): images [N][C][H][W] fp32 -> patches [N][Ho][Wo][cin_pad]
// in the compute dtype, element e = tap*C + c (tap = r*k + s, tap-major like the KRSC weights), zero beyond k*k*C.
// The stem convolution then runs as a 1x1 convolution over these rows on the common kernels.
//
// Augmentation follows torchvision's order on the un-normalised image, restated on the normalised tensor the loader hands over
// (reference config/data/CIFAR10.yaml:11-13, data_preparation.py:173-200: transforms = [RandomCrop(32, 4), RandomHorizontalFlip,
// ToTensor, Normalize]):  aug[y][x] = padded[y + oy][xf + ox],  xf = flip ? W-1-x : x,  (oy, ox) in [0, 2*crop_pad]^2, where
// `padded` is the image with a border of crop_pad BLACK pixels -- after normalisation black is -mean[c]/std[c] = pad_value[c].
// HBM-bound: one 16-byte vector of output per thread.
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void stem_patches_kernel(const float* __restrict__ img, uint4* __restrict__ out, long long n_vec, int C, int H,
                                                          int W, int Ho, int Wo, int k, int stride, int pad, int cin_pad,
                                                          const signed char* __restrict__ oy, const signed char* __restrict__ ox,
                                                          const signed char* __restrict__ flip, int crop_pad, float pv0, float pv1, float pv2,
                                                          float pv3) {
    constexpr int V = ET<T>::VEC;
    const int vec_per_px = cin_pad / V;
    const int kkc = k * k * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i % vec_per_px);
        long long px = i / vec_per_px;
        const int x = (int)(px % Wo); px /= Wo;
        const int y = (int)(px % Ho);
        const long long n = px / Ho;
        const int dy = oy ? (int)oy[n] - crop_pad : 0, dx = ox ? (int)ox[n] - crop_pad : 0;
        const bool fl = flip ? flip[n] != 0 : false;
        float v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int el = j * V + e;
            float val = 0.f;
            if (el < kkc) {
                const int tap = el / C, c = el - tap * C;
                const int r = tap / k, s = tap - r * k;
                const int iy = y * stride + r - pad, ix = x * stride + s - pad;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {       // else: zero padding of the convolution
                    const int sy = iy + dy, sx = (fl ? W - 1 - ix : ix) + dx;          // position in the un-cropped image
                    if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) val = img[((n * C + c) * H + sy) * W + sx];
                    else val = c == 0 ? pv0 : (c == 1 ? pv1 : (c == 2 ? pv2 : pv3)); // black border of RandomCrop's padding
                }
            }
            v[e] = val;
        }
        out[i] = ET<T>::pack(v);
    }
}

extern "C" int fb_stem_patches(const float* images, void* patches, int64_t n_img, int32_t C, int32_t H, int32_t W, int32_t k, int32_t stride,
                               int32_t pad, int32_t cin_pad, const int8_t* crop_oy, const int8_t* crop_ox, const int8_t* flip,
                               int32_t crop_pad, const float* pad_value, int32_t dtype, void* stream) {
    if (!images || !patches) FB_FAIL(FB_ERR_ARG, "fb_stem_patches: null pointer");
    const int V = dtype == FB_F32 ? 4 : 8;
    if (C < 1 || C > 4 || k < 1 || cin_pad % V != 0 || k * k * C > cin_pad) FB_FAIL(FB_ERR_SHAPE, "fb_stem_patches: C=%d k=%d cin_pad=%d", C, k, cin_pad);
    if ((crop_oy == nullptr) != (crop_ox == nullptr)) FB_FAIL(FB_ERR_ARG, "fb_stem_patches: crop_oy and crop_ox come together");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const long long n_vec = (long long)n_img * Ho * Wo * (cin_pad / V);
    if (n_vec == 0) return FB_OK;
    float pv[4] = {0.f, 0.f, 0.f, 0.f};
    if (pad_value) for (int c = 0; c < C; ++c) pv[c] = pad_value[c];                  // host array (C floats)
    const long long want = (n_vec + 255) / 256;
    const unsigned nb = (unsigned)(want < 262144 ? want : 262144);
    if (dtype == FB_F32)
        hipLaunchKernelGGL((stem_patches_kernel<float>), dim3(nb), dim3(256), 0, (hipStream_t)stream, images, (uint4*)patches, n_vec, C, H, W, Ho, Wo,
                           k, stride, pad, cin_pad, (const signed char*)crop_oy, (const signed char*)crop_ox, (const signed char*)flip, crop_pad, pv[0],
                           pv[1], pv[2], pv[3]);
    else
        hipLaunchKernelGGL((stem_patches_kernel<bf16_tag>), dim3(nb), dim3(256), 0, (hipStream_t)stream, images, (uint4*)patches, n_vec, C, H, W, Ho,
                           Wo, k, stride, pad, cin_pad, (const signed char*)crop_oy, (const signed char*)crop_ox, (const signed char*)flip, crop_pad,
                           pv[0], pv[1], pv[2], pv[3]);
    FB_CHECK_LAUNCH("fb_stem_patches");
    return FB_OK;
}
