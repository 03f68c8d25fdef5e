// 3x3 / stride 1 / pad 1 convolution (forward and input-gradient) from an LDS-resident input halo -- low-overhead version.
//
// Same decomposition as conv3x3_halo.hip (256 output pixels = TH full rows of one image x 64 output channels per workgroup,
// nine taps from one staged halo per 128-byte channel slice, weights of the current tap double-buffered) with the
// instruction overhead removed (PMC: the first version issued ~7 VALU per MFMA):
//   * halo and weight tiles go HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds`; zero padding = out-of-range voffset
//   * the halo row pitch is a multiple of 8 rows (40 for W=32, 24 for W=16), so the XOR swizzle (row & 7) depends only on the
//     horizontal position: the three horizontal variants x two K-halves of every fragment address are precomputed
//     (24 VGPRs) and the vertical tap offset is an instruction immediate -> no address arithmetic in the tap loop
//   * fragment reads are inline-asm ds_read_b128 with immediates, the nine taps are fully unrolled
//   * hardware bf16 conversion in the epilogue
#include "common.h"

#include <type_traits>

struct Halo2Params {
    const char* src; const char* wgt; char* dst; const char* addend; float* stat;
    int n_img, H, Cs, Cd, mode;
    int imgs_per_wset; long long wset_stride_bytes;
    int addend_mode, n_mblocks, n_ct, n_blocks;
};

namespace {
template <int N> __device__ __forceinline__ void h2_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void h2_wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
typedef __attribute__((ext_vector_type(4))) unsigned h2_u32x4;
template <int OFF> __device__ __forceinline__ uint4 h2_read16(unsigned byte_addr) {
    h2_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
template <int I, int N, typename F> __device__ __forceinline__ void h2_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); h2_static_for<I + 1, N>(f); }
}
__device__ __forceinline__ int h2_xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, xcd = b & 7, slot = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
constexpr unsigned H2_OOB = 0x80000000u;
}  // namespace

template <typename T, int W>
__global__ __launch_bounds__(256) void conv3x3s1_halo2_kernel(const Halo2Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int EB = ET<T>::EB;
    constexpr int TH = 256 / W;
    constexpr int HWP = W == 32 ? 40 : 24;               // halo row pitch (rows of 128 B), multiple of 8
    constexpr int HRP = (TH + 2) * HWP;                   // halo rows incl. pitch padding (multiple of 8)
    constexpr int NGRP = HRP / 8;                         // 1 KiB row groups of the halo
    constexpr int HALO_BYTES = HRP * 128, WT_BYTES = 64 * 128;
    __shared__ __attribute__((aligned(16))) char lds[HALO_BYTES + 2 * WT_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = h2_xcd_remap(blockIdx.x, p.n_blocks);
    // divisions run on the VALU: readfirstlane restores provable uniformity (else every buffer op gets a waterfall loop)
    const int ct = __builtin_amdgcn_readfirstlane(L % p.n_ct), pt = __builtin_amdgcn_readfirstlane(L / p.n_ct);
    const int tiles_per_img = p.H / TH;
    const int n = __builtin_amdgcn_readfirstlane(pt / tiles_per_img), y0 = (pt - n * tiles_per_img) * TH;
    const int wset = __builtin_amdgcn_readfirstlane(n / p.imgs_per_wset);
    const int row_b = p.Cs * EB;
    const int lrow8 = lane >> 3;                           // row within a 1 KiB group
    const int chunk = (lane & 7) ^ lrow8;                  // logical chunk fetched by this lane (source-side swizzle)

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + (long long)n * p.H * W * row_b), 0, p.H * W * row_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wgt + (long long)wset * p.wset_stride_bytes), 0, p.Cd * 9 * row_b, 0x00020000);
    const unsigned voffW0 = (unsigned)((ct * 64 + wave * 8 + lrow8) * 9 * row_b + chunk * 16);
    const unsigned voffW1 = voffW0 + (unsigned)(32 * 9 * row_b);

    auto halo_issue = [&](int cc) {
        const int soff = cc * 128;
#pragma unroll
        for (int k = 0; k < (NGRP + 3) / 4; ++k) {
            const int g = wave + 4 * k;
            if (g < NGRP) {
                const int row = g * 8 + lrow8;
                const int hy = row / HWP, hx = row - hy * HWP;
                const int sy = y0 + hy - 1, sx = hx - 1;
                const bool ok = (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)W;
                const unsigned voff = ok ? (unsigned)((sy * W + sx) * row_b + chunk * 16) : H2_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(lds + g * 1024), 16, voff, soff, 0, 0);
            }
        }
    };
    auto wt_issue = [&](int buf, int cc, int t) {
        const int soff = t * row_b + cc * 128;
        char* dst = lds + HALO_BYTES + buf * WT_BYTES + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)dst, 16, voffW0, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)(dst + 4096), 16, voffW1, soff, 0, 0);
    };

    // ---- precomputed fragment addresses --------------------------------------------------------------------------------
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    unsigned pa[4][3][2];                                  // [pixel fragment][horizontal variant dx+1][K half]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = wave * 64 + j * 16 + (lane & 15);
        const int ty = q / W, tx = q % W;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int hrow = ty * HWP + tx + b;            // halo row for vertical index a = 0 (add a*HWP rows as an immediate)
#pragma unroll
            for (int h = 0; h < 2; ++h) pa[j][b][h] = lds0 + hrow * 128 + ((((lane >> 4) + 4 * h) ^ (hrow & 7)) * 16);
        }
    }
    unsigned wa[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) wa[h] = lds0 + HALO_BYTES + (lane & 15) * 128 + ((((lane >> 4) + 4 * h) ^ (lane & 7)) * 16);

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int n_cc = row_b / 128;
    const int t_first = p.mode == 0 ? 0 : 8;
    halo_issue(0);
    wt_issue(0, 0, t_first);
    h2_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    for (int cc = 0; cc < n_cc; ++cc) {
        // nine taps toggle the weight double buffer an odd number of times per channel slice: the buffer of tap U is
        // (U & 1) ^ (cc & 1).  Two base registers per K-half (even / odd taps) keep the read offsets immediates.
        const int par = cc & 1;
        const unsigned wE[2] = {wa[0] + par * WT_BYTES, wa[1] + par * WT_BYTES};
        const unsigned wO[2] = {wa[0] + (par ^ 1) * WT_BYTES, wa[1] + (par ^ 1) * WT_BYTES};
        h2_static_for<0, 9>([&](auto uc) {
            constexpr int U = decltype(uc)::value, A = U / 3, B = U % 3;
            const int nbuf = ((U & 1) ^ par) ^ 1;
            // weights of the next tap (or of tap 0 of the next channel slice) into the other buffer
            if (U < 8) wt_issue(nbuf, cc, p.mode == 0 ? U + 1 : 7 - U);
            else if (cc + 1 < n_cc) wt_issue(nbuf, cc + 1, t_first);
            const unsigned wb0 = (U & 1) ? wO[0] : wE[0], wb1 = (U & 1) ? wO[1] : wE[1];
            uint4 wf0[4], pf0[4], wf1[4], pf1[4];
            h2_static_for<0, 4>([&](auto i) { wf0[decltype(i)::value] = h2_read16<decltype(i)::value * 2048>(wb0); });
            h2_static_for<0, 4>([&](auto j) { pf0[decltype(j)::value] = h2_read16<A * HWP * 128>(pa[decltype(j)::value][B][0]); });
            h2_static_for<0, 4>([&](auto i) { wf1[decltype(i)::value] = h2_read16<decltype(i)::value * 2048>(wb1); });
            h2_static_for<0, 4>([&](auto j) { pf1[decltype(j)::value] = h2_read16<A * HWP * 128>(pa[decltype(j)::value][B][1]); });
            h2_wait_lgkmcnt<8>();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma_chunk<T>(wf0[i], pf0[j], acc[i][j]);
            h2_wait_lgkmcnt<0>();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma_chunk<T>(wf1[i], pf1[j], acc[i][j]);
            if (U == 8 && cc + 1 < n_cc) {
                __builtin_amdgcn_s_barrier();             // every wave is done with the old halo
                halo_issue(cc + 1);
            }
            h2_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        });
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------------------
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
        if (tid < 128) {
            const int half = tid >> 6, col = tid & 63;
            const float a = red[((2 * half) * 64 + col) * 2] + red[((2 * half + 1) * 64 + col) * 2];
            const float b = red[((2 * half) * 64 + col) * 2 + 1] + red[((2 * half + 1) * 64 + col) * 2 + 1];
            const long long blk = 2LL * pt + half;
            p.stat[blk * p.Cd + ct * 64 + col] = a;
            p.stat[((long long)p.n_mblocks + blk) * p.Cd + ct * 64 + col] = b;
        }
    }
#endif
}

// returns 1 if the kernel handled the call
int fb_try_conv3x3_halo2(const fb_conv_args* a, hipStream_t st) {
    static const bool disabled = getenv("FB_DISABLE_HALO2") != nullptr;
    if (disabled) return 0;
    if (a->R != 3 || a->S != 3 || a->stride != 1 || a->pad != 1) return 0;
    if (a->Hs != a->Hd || a->Ws != a->Wd || a->Hs != a->Ws) return 0;
    const int W = a->Ws;
    if (W != 32 && W != 16) return 0;
    const int EB = a->dtype == FB_F32 ? 4 : 2;
    if (a->Cs * EB % 128 != 0 || a->Cd % 64 != 0) return 0;
    if ((long long)a->Hs * W * a->Cs * EB >= (1LL << 31)) return 0;
    Halo2Params p;
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
        if (W == 32) hipLaunchKernelGGL((conv3x3s1_halo2_kernel<float, 32>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3s1_halo2_kernel<float, 16>), grid, dim3(256), 0, st, p);
    } else {
        if (W == 32) hipLaunchKernelGGL((conv3x3s1_halo2_kernel<bf16_tag, 32>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3s1_halo2_kernel<bf16_tag, 16>), grid, dim3(256), 0, st, p);
    }
    return 1;
}
