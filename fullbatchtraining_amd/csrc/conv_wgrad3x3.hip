// Weight gradient of 3x3 / stride 1 / pad 1 convolutions on wide feature maps (W = 32 or 16): all nine taps from one
// LDS-resident input halo.
//
//   dW[co][tap][ci] = sum_p dY[p][co] * X[p shifted by tap][ci]
//
// One workgroup owns a 64(co) x 64(ci) x 9(tap) output block and a slice of whole images of one chunk (split-K over images).
// Per K-step (64 output pixels = 2 or 4 full image rows) it stages dY[64 px][64 co] and the X halo [(rows+2)x(W+2)][64 ci]
// once and feeds all nine taps from shifted *transposed* fragment reads (ds_read_b64_tr_b16 / ds_read_b32), so dY and X are
// read from global memory once instead of nine times.  Wave w accumulates ci block [16w,16w+16) x 9 taps x 64 co
// (36 fragments = 144 accumulator VGPRs).  Output: the same fp32 slab layout as conv_wgrad.hip, reduced in fixed order.
#include "common.h"
#include "profile.h"

struct Wgrad3Params {
    const char* x; const char* dy; float* out;
    int n_img, H, Cs, Cd;
    int imgs_per_group, imgs_per_block, split_k;
    long long group_stride;
    const float* amax_x; const float* amax_dy;              // f32h (fp16x2 split): largest magnitudes of the two operand tensors
};

template <typename T> struct W3 { };
template <> struct W3<bf16_tag> { static constexpr int PAD = 16; };
template <> struct W3<float> { static constexpr int PAD = 64; };
template <> struct W3<f32s_tag> : W3<float> {};
template <> struct W3<f32h_tag> : W3<float> {};

// A operand: dY tile rows are the step's pixels in order
template <typename T> __device__ __forceinline__ void frag_plain(const char* tile, int row_bytes, int pb, int c0, int lane, uint4 (&out)[2]);
template <> __device__ __forceinline__ void frag_plain<bf16_tag>(const char* tile, int row_bytes, int pb, int c0, int lane, uint4 (&out)[2]) {
    const int t = lane & 15, g = lane >> 4;
    const char* a0 = tile + (pb + g * 8 + (t >> 2)) * row_bytes + (c0 + (t & 3) * 4) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0 + 4 * row_bytes));
    out[0] = make_uint4(((unsigned)(unsigned short)lo[0]) | ((unsigned)(unsigned short)lo[1] << 16),
                        ((unsigned)(unsigned short)lo[2]) | ((unsigned)(unsigned short)lo[3] << 16),
                        ((unsigned)(unsigned short)hi[0]) | ((unsigned)(unsigned short)hi[1] << 16),
                        ((unsigned)(unsigned short)hi[2]) | ((unsigned)(unsigned short)hi[3] << 16));
}
template <> __device__ __forceinline__ void frag_plain<float>(const char* tile, int row_bytes, int pb, int c0, int lane, uint4 (&out)[2]) {
    const int t = lane & 15, g = lane >> 4;
    const char* a0 = tile + (pb + g) * row_bytes + (c0 + t) * 4;
    unsigned v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = *(const unsigned*)(a0 + 4 * e * row_bytes);
    out[0] = make_uint4(v[0], v[1], v[2], v[3]);
    out[1] = make_uint4(v[4], v[5], v[6], v[7]);
}
// B operand: X halo; pixel p of the step sits at halo row (p/W + r)*(W+2) + p%W + s
template <typename T, int W> __device__ __forceinline__ void frag_halo(const char* tile, int row_bytes, int pb, int r, int s, int c0, int lane, uint4 (&out)[2]);
template <> __device__ __forceinline__ void frag_halo<bf16_tag, 32>(const char* tile, int row_bytes, int pb, int r, int s, int c0, int lane, uint4 (&out)[2]) {
    const int t = lane & 15, g = lane >> 4;
    const int p = pb + g * 8 + (t >> 2);                      // 8 consecutive pixels of one image row per lane group
    const char* a0 = tile + ((p / 32 + r) * 34 + (p % 32) + s) * row_bytes + (c0 + (t & 3) * 4) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0 + 4 * row_bytes));
    out[0] = make_uint4(((unsigned)(unsigned short)lo[0]) | ((unsigned)(unsigned short)lo[1] << 16),
                        ((unsigned)(unsigned short)lo[2]) | ((unsigned)(unsigned short)lo[3] << 16),
                        ((unsigned)(unsigned short)hi[0]) | ((unsigned)(unsigned short)hi[1] << 16),
                        ((unsigned)(unsigned short)hi[2]) | ((unsigned)(unsigned short)hi[3] << 16));
}
template <> __device__ __forceinline__ void frag_halo<bf16_tag, 16>(const char* tile, int row_bytes, int pb, int r, int s, int c0, int lane, uint4 (&out)[2]) {
    const int t = lane & 15, g = lane >> 4;
    const int p = pb + g * 8 + (t >> 2);
    const char* a0 = tile + ((p / 16 + r) * 18 + (p % 16) + s) * row_bytes + (c0 + (t & 3) * 4) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0 + 4 * row_bytes));
    out[0] = make_uint4(((unsigned)(unsigned short)lo[0]) | ((unsigned)(unsigned short)lo[1] << 16),
                        ((unsigned)(unsigned short)lo[2]) | ((unsigned)(unsigned short)lo[3] << 16),
                        ((unsigned)(unsigned short)hi[0]) | ((unsigned)(unsigned short)hi[1] << 16),
                        ((unsigned)(unsigned short)hi[2]) | ((unsigned)(unsigned short)hi[3] << 16));
}
template <> __device__ __forceinline__ void frag_halo<bf16_tag, 8>(const char* tile, int row_bytes, int pb, int r, int s, int c0, int lane, uint4 (&out)[2]) {
    const int t = lane & 15, g = lane >> 4;
    const int p = pb + g * 8 + (t >> 2);                      // a lane group's 8 pixels = one whole row of an 8x8 image
    const char* a0 = tile + ((p / 8 + r) * 10 + (p % 8) + s) * row_bytes + (c0 + (t & 3) * 4) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0 + 4 * row_bytes));
    out[0] = make_uint4(((unsigned)(unsigned short)lo[0]) | ((unsigned)(unsigned short)lo[1] << 16),
                        ((unsigned)(unsigned short)lo[2]) | ((unsigned)(unsigned short)lo[3] << 16),
                        ((unsigned)(unsigned short)hi[0]) | ((unsigned)(unsigned short)hi[1] << 16),
                        ((unsigned)(unsigned short)hi[2]) | ((unsigned)(unsigned short)hi[3] << 16));
}
// 4x4 maps: a K-step holds four whole images (6x6 halo each); a lane group's 8 pixels are two image rows of one image
template <> __device__ __forceinline__ void frag_halo<bf16_tag, 4>(const char* tile, int row_bytes, int pb, int r, int s, int c0, int lane, uint4 (&out)[2]) {
    const int t = lane & 15, g = lane >> 4;
    const int p = pb + g * 8 + (t >> 2), q = p & 15;
    const char* a0 = tile + ((p >> 4) * 36 + ((q >> 2) + r) * 6 + (q & 3) + s) * row_bytes + (c0 + (t & 3) * 4) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0 + 6 * row_bytes));
    out[0] = make_uint4(((unsigned)(unsigned short)lo[0]) | ((unsigned)(unsigned short)lo[1] << 16),
                        ((unsigned)(unsigned short)lo[2]) | ((unsigned)(unsigned short)lo[3] << 16),
                        ((unsigned)(unsigned short)hi[0]) | ((unsigned)(unsigned short)hi[1] << 16),
                        ((unsigned)(unsigned short)hi[2]) | ((unsigned)(unsigned short)hi[3] << 16));
}
// stride 2 (fp16x2 planes only): output pixel (py, px) of the step reads halo row (2 py + r) * (2 W + 1) + 2 px + s; the 8 pixels of a lane
// group are one output row segment (W >= 8: +4 pixels = +8 halo columns) or two output rows of a 4x4 map
template <int W> __device__ __forceinline__ void frag_halo_s2(const char* tile, int row_bytes, int pb, int r, int s, int c0, int lane, uint4 (&out)[2]) {
    constexpr int PW = 2 * W + 1, RS = W == 4 ? 4 : 64 / W, IMG_ROWS = (2 * RS + 1) * PW;
    const int t = lane & 15, g = lane >> 4;
    const int p = pb + g * 8 + (t >> 2);
    const int img_l = W == 4 ? p >> 4 : 0, q = W == 4 ? p & 15 : p;
    const char* a0 = tile + (img_l * IMG_ROWS + (2 * (q / W) + r) * PW + 2 * (q % W) + s) * row_bytes + (c0 + (t & 3) * 4) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0 + (W == 4 ? 2 * PW : 8) * row_bytes));
    out[0] = make_uint4(((unsigned)(unsigned short)lo[0]) | ((unsigned)(unsigned short)lo[1] << 16),
                        ((unsigned)(unsigned short)lo[2]) | ((unsigned)(unsigned short)lo[3] << 16),
                        ((unsigned)(unsigned short)hi[0]) | ((unsigned)(unsigned short)hi[1] << 16),
                        ((unsigned)(unsigned short)hi[2]) | ((unsigned)(unsigned short)hi[3] << 16));
}
template <int W> __device__ __forceinline__ void frag_halo_f32(const char* tile, int row_bytes, int pb, int r, int s, int c0, int lane, uint4 (&out)[2]) {
    const int t = lane & 15, g = lane >> 4;
    unsigned v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int p = pb + g + 4 * e;
        if constexpr (W == 4) v[e] = *(const unsigned*)(tile + ((p >> 4) * 36 + (((p & 15) >> 2) + r) * 6 + (p & 3) + s) * row_bytes + (c0 + t) * 4);
        else v[e] = *(const unsigned*)(tile + ((p / W + r) * (W + 2) + (p % W) + s) * row_bytes + (c0 + t) * 4);
    }
    out[0] = make_uint4(v[0], v[1], v[2], v[3]);
    out[1] = make_uint4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void frag_halo<float, 32>(const char* tile, int row_bytes, int pb, int r, int s, int c0, int lane, uint4 (&out)[2]) { frag_halo_f32<32>(tile, row_bytes, pb, r, s, c0, lane, out); }
template <> __device__ __forceinline__ void frag_halo<float, 16>(const char* tile, int row_bytes, int pb, int r, int s, int c0, int lane, uint4 (&out)[2]) { frag_halo_f32<16>(tile, row_bytes, pb, r, s, c0, lane, out); }
template <> __device__ __forceinline__ void frag_halo<float, 8>(const char* tile, int row_bytes, int pb, int r, int s, int c0, int lane, uint4 (&out)[2]) { frag_halo_f32<8>(tile, row_bytes, pb, r, s, c0, lane, out); }
template <> __device__ __forceinline__ void frag_halo<float, 4>(const char* tile, int row_bytes, int pb, int r, int s, int c0, int lane, uint4 (&out)[2]) { frag_halo_f32<4>(tile, row_bytes, pb, r, s, c0, lane, out); }

// W = OUTPUT map width (p.H = output height); SD = 2 (fp16x2 planes only): stride-2 layers, the halo holds the 2 RS + 1 input rows and
// 2 W + 1 input columns the step's 64 output pixels touch (the per-tap kernel re-read dY nine times there: 111-217 fp32-TFLOP/s)
template <typename T, int W, int SD = 1>
__global__ __launch_bounds__(256) void conv_wgrad3x3_kernel(const Wgrad3Params p) {
    constexpr int EB = ET<T>::EB;
    constexpr int IMGS = W == 4 ? 4 : 1;             // whole images per K-step (4x4 maps: four)
    constexpr int RS = 64 / (W * IMGS);              // output rows per K-step and image (64 pixels)
    constexpr int PW = SD == 1 ? W + 2 : 2 * W + 1, PH = SD == 1 ? RS + 2 : 2 * RS + 1;   // halo columns / rows of one image part
    constexpr int IMG_ROWS = PH * PW;                // halo pixels of one image part
    constexpr int HROWS = IMGS * IMG_ROWS;           // halo pixels per K-step
    static_assert(SD == 1 || is_hsplit<T>::value, "stride 2 is built for the fp16x2 planes only");
    constexpr int ROW = 64 * EB + W3<T>::PAD;        // LDS row: 64 channels + pad
    constexpr int CH = 64 * EB / 16;                 // 16-byte chunks per row
    constexpr int LD_A = 64 * CH / 256;              // dY chunks per thread
    constexpr int LD_B = (HROWS * CH + 255) / 256;   // halo chunks per thread
    // split path (fp32 storage, bf16x6 arithmetic): both tiles are stored as THREE bf16 planes (every loaded element split once, at
    // the LDS store) and read with the transposed bf16 fragment reads -- no per-fragment split in the tap loop
    // f32h: TWO scaled fp16 planes, three MFMAs per fragment pair; the accumulators hold the scaled sums, unscaled at the output
    constexpr bool HSPLIT = is_hsplit<T>::value;
    constexpr bool SPLIT = is_split<T>::value || HSPLIT;
    constexpr int NPL = HSPLIT ? 2 : 3;
    constexpr int PROW = 64 * 2 + W3<bf16_tag>::PAD; // plane row (bf16 / fp16)
    constexpr int PLANE_A = 64 * PROW, PLANE_B = HROWS * PROW;
    __shared__ __attribute__((aligned(16))) char lds[SPLIT ? NPL * (PLANE_A + PLANE_B) : (64 + HROWS) * ROW];
    char* tileA = lds;
    char* tileB = lds + (SPLIT ? NPL * PLANE_A : 64 * ROW);
    float hs_a = 1.f, hs_b = 1.f;
    if constexpr (HSPLIT) { hs_a = fb_pow2_scale(p.amax_dy[blockIdx.y / p.split_k]); hs_b = fb_pow2_scale(p.amax_x[blockIdx.y / p.split_k]); }   // per chunk

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = p.Cs / 64;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int group = blockIdx.y / p.split_k, split = blockIdx.y % p.split_k;
    const int img0 = group * p.imgs_per_group + split * p.imgs_per_block;
    const int img_end = min(img0 + p.imgs_per_block, (group + 1) * p.imgs_per_group);
    const int steps_per_img = p.H / RS;
    const int n_steps = IMGS > 1 ? (img_end - img0 + IMGS - 1) / IMGS : (img_end - img0) * steps_per_img;   // (a ragged last step reads zeros)

    f32x4_t acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    uint4 ra[LD_A], rb[LD_B];
    auto gload = [&](int step) {
        const int img = IMGS > 1 ? img0 + step * IMGS : img0 + step / steps_per_img, y0 = IMGS > 1 ? 0 : (step % steps_per_img) * RS;
        const char* dyb = p.dy + (((long long)img * p.H + y0) * W * p.Cd + tile_m * 64) * EB;
#pragma unroll
        for (int i = 0; i < LD_A; ++i) {
            const int id = tid + 256 * i, row = id / CH, ch = id % CH;
            ra[i] = make_uint4(0, 0, 0, 0);
            if (IMGS == 1 || img + row / (W * W) < img_end) ra[i] = *(const uint4*)(dyb + (long long)row * p.Cd * EB + ch * 16);
        }
        const int Hi = SD * p.H;                                  // input map: Hi x WI
        constexpr int WI = SD * W;
        const char* xb = p.x + ((long long)img * Hi * WI * p.Cs + tile_n * 64) * EB;
#pragma unroll
        for (int i = 0; i < LD_B; ++i) {
            const int id = tid + 256 * i, row = id / CH, ch = id % CH;
            rb[i] = make_uint4(0, 0, 0, 0);
            if (row < HROWS) {
                const int img_l = row / IMG_ROWS, rr = row - img_l * IMG_ROWS;
                const int hy = rr / PW, hx = rr - hy * PW;
                const int sy = SD * y0 + hy - 1, sx = hx - 1;
                if ((unsigned)sy < (unsigned)Hi && (unsigned)sx < (unsigned)WI && img + img_l < img_end)
                    rb[i] = *(const uint4*)(xb + ((long long)img_l * Hi * WI + sy * WI + sx) * p.Cs * EB + ch * 16);
            }
        }
    };
    auto lstore = [&]() {
        if constexpr (SPLIT) {
#pragma unroll
            for (int i = 0; i < LD_A; ++i) {
                const int id = tid + 256 * i, row = id / CH, ch = id % CH;
                uint2 h, m, l;
                char* d = tileA + row * PROW + ch * 8;
                if constexpr (HSPLIT) { split_h2x4(ra[i], hs_a, h, l); *(uint2*)d = h; *(uint2*)(d + PLANE_A) = l; }
                else { split_f32x4(ra[i], h, m, l); *(uint2*)d = h; *(uint2*)(d + PLANE_A) = m; *(uint2*)(d + 2 * PLANE_A) = l; }
            }
#pragma unroll
            for (int i = 0; i < LD_B; ++i) {
                const int id = tid + 256 * i, row = id / CH, ch = id % CH;
                if (row < HROWS) {
                    uint2 h, m, l;
                    char* d = tileB + row * PROW + ch * 8;
                    if constexpr (HSPLIT) { split_h2x4(rb[i], hs_b, h, l); *(uint2*)d = h; *(uint2*)(d + PLANE_B) = l; }
                    else { split_f32x4(rb[i], h, m, l); *(uint2*)d = h; *(uint2*)(d + PLANE_B) = m; *(uint2*)(d + 2 * PLANE_B) = l; }
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < LD_A; ++i) { const int id = tid + 256 * i, row = id / CH, ch = id % CH; *(uint4*)(tileA + row * ROW + ch * 16) = ra[i]; }
#pragma unroll
        for (int i = 0; i < LD_B; ++i) { const int id = tid + 256 * i, row = id / CH, ch = id % CH; if (row < HROWS) *(uint4*)(tileB + row * ROW + ch * 16) = rb[i]; }
        }
    };

    if (n_steps > 0) gload(0);
    for (int step = 0; step < n_steps; ++step) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (step + 1 < n_steps) gload(step + 1);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int pb = blk * 32;
            if constexpr (SPLIT) {
                uint4 ap[4][NPL], tmp[2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) { frag_plain<bf16_tag>(tileA + pl * PLANE_A, PROW, pb, i * 16, lane, tmp); ap[i][pl] = tmp[0]; }
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    uint4 bp[NPL];
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
                        if constexpr (SD == 2) frag_halo_s2<W>(tileB + pl * PLANE_B, PROW, pb, t / 3, t % 3, wave * 16, lane, tmp);
                        else frag_halo<bf16_tag, W>(tileB + pl * PLANE_B, PROW, pb, t / 3, t % 3, wave * 16, lane, tmp);
                        bp[pl] = tmp[0];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (HSPLIT) acc[t][i] = mma_planes3h_acc(ap[i], bp, acc[t][i]);
                        else acc[t][i] = mma_planes6(ap[i], bp, acc[t][i]);
                    }
                }
            } else {
            if constexpr (SD == 1) {
            uint4 af[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) frag_plain<T>(tileA, ROW, pb, i * 16, lane, af[i]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                uint4 bf[2];
                frag_halo<T, W>(tileB, ROW, pb, t / 3, t % 3, wave * 16, lane, bf);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[t][i] = mma_chunk<T>(af[i][0], bf[0], acc[t][i]);
                    if constexpr (EB == 4) acc[t][i] = mma_chunk<T>(af[i][1], bf[1], acc[t][i]);
                }
            }
            }
            }
        }
    }

    float* out = p.out + group * p.group_stride + ((long long)split * p.Cd) * 9 * p.Cs;
    const float inv = HSPLIT ? 1.f / (hs_a * hs_b) : 1.f;    // f32h: the accumulators hold sums of (2^e_dy dy) * (2^e_x x)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = tile_m * 64 + i * 16 + (lane >> 4) * 4 + q;
                const int ci = tile_n * 64 + wave * 16 + (lane & 15);
                out[((long long)co * 9 + t) * p.Cs + ci] = HSPLIT ? acc[t][i][q] * inv : acc[t][i][q];
            }
}

// returns 1 if handled.  The caller's slab must hold n_groups*split_k*Cd*9*Cs floats with the split_k passed in.
int fb_try_wgrad3x3(const fb_wgrad_args* a, hipStream_t st) {
    static const bool disabled = getenv("FB_DISABLE_WGRAD3") != nullptr;
    if (disabled) return 0;
    if (a->R != 3 || a->S != 3 || (a->stride != 1 && a->stride != 2) || a->pad != 1) return 0;
    if (a->Hs != a->stride * a->Hd || a->Ws != a->stride * a->Wd || a->Hd != a->Wd) return 0;
    const int W = a->Wd;                                      // OUTPUT map width
    if (a->stride == 2 && !(a->dtype == FB_F32 && a->amax_x && a->amax_dy && (W == 16 || W == 8 || W == 4))) return 0;   // stride 2: fp16x2 planes only
    if (W != 32 && W != 16 && W != 8 && W != 4) return 0;     // (8x8: one whole image per 64-pixel K-step, 4x4: four)
    if (a->Cs % 64 != 0 || a->Cd % 64 != 0) return 0;
    Wgrad3Params p;
    p.x = (const char*)a->x; p.dy = (const char*)a->dy; p.out = a->dw_partial;
    p.group_stride = a->group_stride ? a->group_stride : (long long)a->split_k * a->Cd * 9 * a->Cs;
    p.n_img = a->n_img; p.H = a->Hd; p.Cs = a->Cs; p.Cd = a->Cd;
    p.imgs_per_group = a->imgs_per_group; p.split_k = a->split_k; p.imgs_per_block = (a->imgs_per_group + a->split_k - 1) / a->split_k;   // ragged last K slice allowed
    p.amax_x = a->amax_x; p.amax_dy = a->amax_dy;
    const int n_groups = a->n_img / a->imgs_per_group;
    dim3 grid((a->Cd / 64) * (a->Cs / 64), n_groups * a->split_k);
    if (a->stride == 2) {
        if (W == 16) hipLaunchKernelGGL((conv_wgrad3x3_kernel<f32h_tag, 16, 2>), grid, dim3(256), 0, st, p);
        else if (W == 8) hipLaunchKernelGGL((conv_wgrad3x3_kernel<f32h_tag, 8, 2>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad3x3_kernel<f32h_tag, 4, 2>), grid, dim3(256), 0, st, p);
        return 1;
    }
    if (a->dtype == FB_F32 && a->amax_x && a->amax_dy) {
        if (W == 32) hipLaunchKernelGGL((conv_wgrad3x3_kernel<f32h_tag, 32>), grid, dim3(256), 0, st, p);
        else if (W == 16) hipLaunchKernelGGL((conv_wgrad3x3_kernel<f32h_tag, 16>), grid, dim3(256), 0, st, p);
        else if (W == 8) hipLaunchKernelGGL((conv_wgrad3x3_kernel<f32h_tag, 8>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad3x3_kernel<f32h_tag, 4>), grid, dim3(256), 0, st, p);
    } else if (a->dtype == FB_F32 && fb_f32_split_enabled()) {
        if (W == 32) hipLaunchKernelGGL((conv_wgrad3x3_kernel<f32s_tag, 32>), grid, dim3(256), 0, st, p);
        else if (W == 16) hipLaunchKernelGGL((conv_wgrad3x3_kernel<f32s_tag, 16>), grid, dim3(256), 0, st, p);
        else if (W == 8) hipLaunchKernelGGL((conv_wgrad3x3_kernel<f32s_tag, 8>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad3x3_kernel<f32s_tag, 4>), grid, dim3(256), 0, st, p);
    } else if (a->dtype == FB_F32) {
        if (W == 32) hipLaunchKernelGGL((conv_wgrad3x3_kernel<float, 32>), grid, dim3(256), 0, st, p);
        else if (W == 16) hipLaunchKernelGGL((conv_wgrad3x3_kernel<float, 16>), grid, dim3(256), 0, st, p);
        else if (W == 8) hipLaunchKernelGGL((conv_wgrad3x3_kernel<float, 8>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad3x3_kernel<float, 4>), grid, dim3(256), 0, st, p);
    } else {
        if (W == 32) hipLaunchKernelGGL((conv_wgrad3x3_kernel<bf16_tag, 32>), grid, dim3(256), 0, st, p);
        else if (W == 16) hipLaunchKernelGGL((conv_wgrad3x3_kernel<bf16_tag, 16>), grid, dim3(256), 0, st, p);
        else if (W == 8) hipLaunchKernelGGL((conv_wgrad3x3_kernel<bf16_tag, 8>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad3x3_kernel<bf16_tag, 4>), grid, dim3(256), 0, st, p);
    }
    return 1;
}
