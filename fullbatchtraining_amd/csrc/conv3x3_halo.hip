// 3x3 / stride 1 / pad 1 convolution (forward and input-gradient) with an LDS-resident input halo, for the wide early
// stages (32x32 and 16x16 feature maps, C = 64/128) where the layer sits at the HBM/MFMA ridge.
//
// One workgroup = 256 output pixels of ONE image (TH full rows of width W, TH*W = 256) x 64 output channels.
// For each 128-byte channel slice (64 bf16 / 32 f32) the (TH+2)x(W+2) input halo is staged in LDS ONCE and all nine
// taps are computed from it with shifted fragment reads (9x less global->LDS traffic than gathering per tap); only the
// 64x(128 B) weight tile of the current tap streams through a double-buffered LDS slot.  Zero padding = zero halo rows.
// MFMA orientation, fragment layout, swizzle and epilogue (addend, NHWC stores, per-128-pixel BN partial sums from the
// fp32 accumulators) are those of conv_igemm.hip.  Consecutive workgroups are remapped so that the co-tiles of one pixel
// tile and neighbouring pixel tiles land on the same XCD (shared L2).
#include "common.h"
#include "profile.h"

struct HaloParams {
    const char* src; const char* wgt; char* dst; const char* addend; float* stat;
    int n_img, H, Cs, Cd, mode;
    int imgs_per_wset; long long wset_stride_bytes;
    int addend_mode, n_mblocks, n_ct, n_blocks;
};

__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, xcd = b & 7, slot = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

template <typename T, int W>
__global__ __launch_bounds__(256) void conv3x3s1_halo_kernel(const HaloParams p) {
    constexpr int EB = ET<T>::EB;
    constexpr int CK = 128 / EB;                 // channels per 128-byte slice
    constexpr int TH = 256 / W, HW = W + 2, HR = (TH + 2) * HW;
    __shared__ uint4 lds[(HR + 128) * 8];
    uint4* halo = lds;
    uint4* wt = lds + HR * 8;                    // 2 x 64 rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = xcd_remap(blockIdx.x, p.n_blocks);
    const int ct = L % p.n_ct, pt = L / p.n_ct;
    const int tiles_per_img = p.H / TH;
    const int n = pt / tiles_per_img, y0 = (pt - n * tiles_per_img) * TH;
    const char* wbase = p.wgt + (long long)(n / p.imgs_per_wset) * p.wset_stride_bytes;
    const int chunk = tid & 7, lrow = tid >> 3;
    const long long wrow_bytes = 9LL * p.Cs * EB;
    const char* wptr0 = wbase + (long long)(ct * 64 + lrow) * wrow_bytes + chunk * 16;
    const char* wptr1 = wptr0 + 32 * wrow_bytes;
    const char* img = p.src + (long long)n * p.H * W * p.Cs * EB;

    int base[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = wave * 64 + j * 16 + (lane & 15);
        base[j] = (q / W + 1) * HW + (q % W) + 1;
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int n_cc = p.Cs / CK;
    const int n_iter = n_cc * 9;
    uint4 rw0, rw1;
    auto wload = [&](int it) {
        const int cc = it / 9, t = it - cc * 9;
        const long long off = ((long long)t * p.Cs + cc * CK) * EB;
        rw0 = *(const uint4*)(wptr0 + off);
        rw1 = *(const uint4*)(wptr1 + off);
    };
    auto wstore = [&](int buf) {
        uint4* b = wt + buf * 64 * 8;
        b[lrow * 8 + (chunk ^ (lrow & 7))] = rw0;
        b[(lrow + 32) * 8 + (chunk ^ (lrow & 7))] = rw1;
    };
    auto halo_load = [&](int cc) {
        for (int idx = tid; idx < HR * 8; idx += 256) {
            const int row = idx >> 3, ch = idx & 7;
            const int hy = row / HW, hx = row - hy * HW;
            const int sy = y0 + hy - 1, sx = hx - 1;
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)W)
                v = *(const uint4*)(img + ((long long)(sy * W + sx) * p.Cs + cc * CK) * EB + ch * 16);
            halo[row * 8 + (ch ^ (row & 7))] = v;
        }
    };

    wload(0);
    halo_load(0);
    wstore(0);
    __syncthreads();
    for (int it = 0; it < n_iter; ++it) {
        const int cc = it / 9, t = it - cc * 9;
        const int r = t / 3, s = t - r * 3;
        const int shift = (p.mode == 0) ? ((r - 1) * HW + (s - 1)) : ((1 - r) * HW + (1 - s));
        const bool more = it + 1 < n_iter;
        if (more) wload(it + 1);
        const uint4* wb = wt + (it & 1) * 64 * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = (lane >> 4) + 4 * h;
            uint4 wf[4], pf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = wb[(i * 16 + (lane & 15)) * 8 + (c ^ (lane & 7))];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int row = base[j] + shift; pf[j] = halo[row * 8 + (c ^ (row & 7))]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma_chunk<T>(wf[i], pf[j], acc[i][j]);
        }
        if (more) {
            wstore((it + 1) & 1);
            if (t == 8) {                 // next iteration starts a new channel slice: restage the halo
                __syncthreads();          // every wave is done reading the old halo
                halo_load(cc + 1);
            }
        }
        __syncthreads();
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------------
    float ssum[4][4], ssq[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = wave * 64 + j * 16 + (lane & 15);
        const int oy = y0 + q / W, ox = q % W;
        const long long pix = ((long long)n * p.H + oy) * W + ox;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = ct * 64 + i * 16 + (lane >> 4) * 4;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.addend_mode != 0) {
                const long long apix = p.addend_mode == 1 ? pix : ((long long)n * (p.H >> 1) + (oy >> 1)) * (W >> 1) + (ox >> 1);
                const float sc = p.addend_mode == 1 ? 1.f : 0.25f;
                const char* ap = p.addend + (apix * p.Cd + co) * EB;
                if constexpr (EB == 4) { const float4 a = *(const float4*)ap; v[0] += sc * a.x; v[1] += sc * a.y; v[2] += sc * a.z; v[3] += sc * a.w; }
                else { const uint2 a = *(const uint2*)ap; v[0] += sc * __uint_as_float(a.x << 16); v[1] += sc * __uint_as_float(a.x & 0xffff0000u);
                       v[2] += sc * __uint_as_float(a.y << 16); v[3] += sc * __uint_as_float(a.y & 0xffff0000u); }
            }
            char* dp = p.dst + (pix * p.Cd + co) * EB;
            if constexpr (EB == 4) *(float4*)dp = make_float4(v[0], v[1], v[2], v[3]);
            else *(uint2*)dp = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[i][r] += v[r]; ssq[i][r] += v[r] * v[r]; }
        }
    }
    if (p.stat != nullptr) {
        float* red = (float*)lds;   // [4 waves][64 co][2]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = ssum[i][r], b = ssq[i][r];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
                if ((lane & 15) == 0) {
                    const int col = i * 16 + (lane >> 4) * 4 + r;
                    red[(wave * 64 + col) * 2] = a; red[(wave * 64 + col) * 2 + 1] = b;
                }
            }
        __syncthreads();
        if (tid < 128) {                      // two 128-pixel statistic rows per tile: waves {0,1} and {2,3}
            const int half = tid >> 6, col = tid & 63;
            const float a = red[((2 * half) * 64 + col) * 2] + red[((2 * half + 1) * 64 + col) * 2];
            const float b = red[((2 * half) * 64 + col) * 2 + 1] + red[((2 * half + 1) * 64 + col) * 2 + 1];
            const long long blk = 2LL * pt + half;
            p.stat[blk * p.Cd + ct * 64 + col] = a;
            p.stat[((long long)p.n_mblocks + blk) * p.Cd + ct * 64 + col] = b;
        }
    }
}

// returns 1 if the halo kernel handled the call, 0 if the shape is not eligible
int fb_try_conv3x3_halo(const fb_conv_args* a, hipStream_t st) {
    static const bool disabled = getenv("FB_DISABLE_HALO") != nullptr;
    if (disabled) return 0;
    if (a->R != 3 || a->S != 3 || a->stride != 1 || a->pad != 1) return 0;
    if (a->Hs != a->Hd || a->Ws != a->Wd || a->Hs != a->Ws) return 0;
    const int W = a->Ws;
    if (W != 32 && W != 16) return 0;
    const int EB = a->dtype == FB_F32 ? 4 : 2, CK = 128 / EB;
    if (a->Cs % CK != 0 || a->Cd % 64 != 0) return 0;
    HaloParams p;
    p.src = (const char*)a->src; p.wgt = (const char*)a->wgt; p.dst = (char*)a->dst; p.addend = (const char*)a->addend;
    p.stat = a->stat_partial;
    p.n_img = a->n_img; p.H = a->Hs; p.Cs = a->Cs; p.Cd = a->Cd; p.mode = a->mode;
    p.imgs_per_wset = a->imgs_per_wset > 0 ? a->imgs_per_wset : a->n_img;
    p.wset_stride_bytes = a->wset_stride * EB;
    p.addend_mode = a->addend ? a->addend_mode : 0;
    const int n_pt = a->n_img * (a->Hs * W / 256);
    p.n_mblocks = n_pt * 2;
    p.n_ct = a->Cd / 64;
    p.n_blocks = n_pt * p.n_ct;
    dim3 grid(p.n_blocks);
    if (a->dtype == FB_F32) {
        if (W == 32) hipLaunchKernelGGL((conv3x3s1_halo_kernel<float, 32>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3s1_halo_kernel<float, 16>), grid, dim3(256), 0, st, p);
    } else {
        if (W == 32) hipLaunchKernelGGL((conv3x3s1_halo_kernel<bf16_tag, 32>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3s1_halo_kernel<bf16_tag, 16>), grid, dim3(256), 0, st, p);
    }
    return 1;
}
