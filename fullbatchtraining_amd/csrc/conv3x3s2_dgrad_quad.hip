// Input gradient of the 3x3 / stride-2 / pad-1 convolutions (first convolution of a downsampling block): all four parity classes of an
// output 2x2 quad from ONE staged dY neighbourhood.
//
//   d_in[2qy+cy][2qx+cx][ci] = sum over the taps (r, s) with (cy+1-r), (cx+1-s) even of  dY[qy+(cy+1-r)/2][qx+(cx+1-s)/2][:] . W[:, r, s, ci]
//
// i.e. class (0,0) sees 1 tap, (0,1) and (1,0) two, (1,1) four -- 9 taps over the 4 pixels of a quad, every one reading the 2x2 dY
// neighbourhood dY[qy..qy+1][qx..qx+1].  The implicit-GEMM kernel runs the classes as separate tile passes with 1-4 taps each: a tile
// lives in set-up, first-load latency and epilogue (128->64 at 32x32: 344 TF/s, 256->128 at 16x16: 540 TF/s).
// Here a workgroup owns 128 quads x 64 output channels x all four classes (512 output pixels):
//   * per 32-channel half-slice of dY (64-byte rows) the (rows+1) x (W+1) halo of the quads (<= 200 rows, 12.5 KiB) and the 64 x 64-byte
//     weight rows of ALL nine taps (36 KiB) are staged by LDS-DMA into ONE 49 KiB stage; two workgroups share a CU, so one's staging
//     wait and epilogue run under the other's 72 MFMAs per wave and half-slice (template STAGES = 2: the double-buffered one-workgroup form)
//   * a wave owns 32 quads: 4 classes x 2 quad fragments x 4 channel fragments = 128 accumulator registers; a tap contributes to
//     exactly one class, so it costs 4 weight-fragment + 2 pixel-fragment reads for 8 MFMAs; every fragment address is a lane register
//     + an immediate (64-byte rows: a wave's 16-row fragment read is one contiguous KiB, no swizzle needed)
//   * the epilogue adds the shortcut branch's gradient (AvgPool2d(2,2) backward: 0.25 x the quad's value for all four pixels, read
//     once per quad; or a full-resolution addend) and leaves with 16-byte stores (v_permlane16_swap row exchange)
#include "common.h"

#include <type_traits>

struct S2DParams {
    const char* src; const char* wgt; char* dst; const char* addend;
    int n_img, Hq, Cs, Cd, addend_mode, n_ct, n_tiles;
};

namespace {
typedef __attribute__((ext_vector_type(4))) unsigned s2d_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned s2d_u32x2;
template <int N> __device__ __forceinline__ void s2d_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void s2d_wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int OFF> __device__ __forceinline__ uint4 s2d_read16(unsigned byte_addr) {
    s2d_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
template <int I, int N, typename F> __device__ __forceinline__ void s2d_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); s2d_static_for<I + 1, N>(f); }
}
__device__ __forceinline__ int s2d_xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, xcd = b & 7, slot = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
constexpr unsigned S2D_OOB = 0x80000000u;

template <int WQ> struct S2DGeo {                                  // WQ = width of dY (quads per row): 16, 8, 4
    static constexpr int THQ = WQ == 16 ? 8 : WQ;                   // quad rows per image part of a 128-quad tile
    static constexpr int IMGS = 128 / (THQ * WQ);                   // image parts per tile: 1, 2, 8
    static constexpr int PITCH = WQ + 1;
    static constexpr int IMG_ROWS = (THQ + 1) * PITCH;              // halo rows per image part: 153, 81, 25
    static constexpr int ROWS = IMGS * IMG_ROWS;                    // 153, 162, 200
    static constexpr int NGRP = (ROWS + 15) / 16;                   // 1 KiB DMA groups of 16 rows x 64 bytes
};
}  // namespace

template <int WQ, int STAGES>
__global__ __launch_bounds__(256, STAGES == 1 ? 2 : 1) void conv3x3s2_dgrad_quad_kernel(const S2DParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = S2DGeo<WQ>;
    constexpr int PITCH = G::PITCH, NGRP = G::NGRP;
    constexpr int HALO_BYTES = NGRP * 1024, WT_BYTES = 9 * 4096, STAGE_BYTES = HALO_BYTES + WT_BYTES;
    __shared__ __attribute__((aligned(16))) char lds[STAGES * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int L = s2d_xcd_remap(blockIdx.x, gridDim.x);
    const int pt = __builtin_amdgcn_readfirstlane(L / p.n_ct), ct = __builtin_amdgcn_readfirstlane(L % p.n_ct);
    // tile origin in dY: WQ = 16 -> image pt / 2, quad rows (pt & 1) * 8 .. +7; WQ = 8 -> images 2 pt, 2 pt + 1; WQ = 4 -> images 8 pt .. 8 pt + 7
    const int tiles_per_img = WQ == 16 ? 2 : 1;
    const int n0 = WQ == 16 ? pt / tiles_per_img : pt * G::IMGS;
    const int y0 = WQ == 16 ? (pt % tiles_per_img) * G::THQ : 0;
    const int row_b = p.Cs * 2;                                     // bytes of a dY pixel / of a (co, tap) weight row
    const int n_cc = row_b / 64;                                    // 32-channel half-slices

    // ---- per-lane DMA offsets (tile invariant but for the validity of the rows below / right of the image) -------------------
    const int drow = lane >> 2;                                     // row of a 16-row DMA group this lane fetches; chunk = lane & 3
    constexpr int KH = (NGRP + 3) / 4;
    unsigned voffH[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k) {
        const int row = (wave + 4 * k) * 16 + drow;
        const int img_l = row / G::IMG_ROWS, rr = row - img_l * G::IMG_ROWS;
        const int hy = rr / PITCH, hx = rr - hy * PITCH;
        const bool ok = row < G::ROWS && hx < WQ && y0 + hy < p.Hq && n0 + img_l < p.n_img;
        voffH[k] = ok ? (unsigned)(((img_l * p.Hq + hy) * WQ + hx) * row_b + (lane & 3) * 16) : S2D_OOB;
    }
    const long long originA = ((long long)(n0 * p.Hq + y0) * WQ) * row_b;
    const __amdgpu_buffer_rsrc_t rsrcA =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + originA), 0, (G::IMGS * p.Hq * WQ) * row_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, p.Cd * 9 * row_b, 0x00020000);
    const unsigned voffW = (unsigned)(((ct * 64 + wave * 16 + drow) * 9) * row_b + (lane & 3) * 16);
    auto issue = [&](int stage, int cc) {
        char* base = lds + stage * STAGE_BYTES;
        const int soff = cc * 64;
        s2d_static_for<0, KH>([&](auto kc) {
            constexpr int K = decltype(kc)::value;
            const int grp = wave + 4 * K;
            if (grp < NGRP)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(base + grp * 1024), 16, voffH[K], soff, 0, 0);
        });
        s2d_static_for<0, 9>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)(base + HALO_BYTES + T * 4096 + wave * 1024), 16,
                                                     voffW, T * row_b + soff, 0, 0);
        });
    };

    // ---- fragment read addresses -------------------------------------------------------------------------------------------------
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned wa = lds0 + HALO_BYTES + col * 64 + g * 16;      // + tap * 4096 + i * 1024 (+ stage)
    unsigned pa[2];                                                  // quad fragment jq of this wave: quads 32 wave + 16 jq + col
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
        const int q = wave * 32 + jq * 16 + col;
        const int img_l = q / (G::THQ * WQ), qi = q - img_l * (G::THQ * WQ);
        pa[jq] = lds0 + (img_l * G::IMG_ROWS + (qi / WQ) * PITCH + qi % WQ) * 64 + g * 16;
    }

    f32x4_t acc[4][4][2];                                            // [channel fragment][class][quad fragment]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int jq = 0; jq < 2; ++jq) acc[i][c][jq] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    if constexpr (STAGES == 2) {
        issue(0, 0);
        s2d_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }
    for (int cc = 0; cc < n_cc; ++cc) {
        const int cur = STAGES == 2 ? (cc & 1) : 0;
        if constexpr (STAGES == 2) {
            if (cc + 1 < n_cc) issue(cur ^ 1, cc + 1);
        } else {
            // one stage (49 KiB): two workgroups share a CU and cover each other's staging and epilogue
            issue(0, cc);
            s2d_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        const unsigned so = cur * STAGE_BYTES;
        const unsigned w0 = wa + so, p0 = pa[0] + so, p1 = pa[1] + so;
        // one wave per SIMD (98 KiB of LDS per workgroup): the six fragment reads of tap T+1 are issued before the MFMAs of tap T
        uint4 wf[2][4], pf[2][2];
        auto read_tap = [&](auto tc, int set) {
            constexpr int T = decltype(tc)::value, R = T / 3, S = T % 3;
            constexpr int DY = R == 0 ? 1 : 0, DX = S == 0 ? 1 : 0;
            s2d_static_for<0, 4>([&](auto i) { wf[set][decltype(i)::value] = s2d_read16<T * 4096 + decltype(i)::value * 1024>(w0); });
            pf[set][0] = s2d_read16<(DY * PITCH + DX) * 64>(p0);
            pf[set][1] = s2d_read16<(DY * PITCH + DX) * 64>(p1);
        };
        read_tap(std::integral_constant<int, 0>{}, 0);
        s2d_static_for<0, 9>([&](auto tc) {
            constexpr int T = decltype(tc)::value, R = T / 3, S = T % 3;
            // tap (R, S) feeds class (cy, cx) from the dY pixel (qy + dy, qx + dx): r = 1 -> cy = 0, dy = 0; r = 0 -> cy = 1, dy = 1; r = 2 -> cy = 1, dy = 0
            constexpr int CY = R == 1 ? 0 : 1, CX = S == 1 ? 0 : 1, CLS = CY * 2 + CX;
            if constexpr (T < 8) { read_tap(std::integral_constant<int, T + 1>{}, (T + 1) & 1); s2d_wait_lgkmcnt<6>(); }
            else s2d_wait_lgkmcnt<0>();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jq = 0; jq < 2; ++jq) acc[i][CLS][jq] = mma_chunk<bf16_tag>(wf[T & 1][i], pf[T & 1][jq], acc[i][CLS][jq]);
        });
        if constexpr (STAGES == 2) s2d_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------------------------
    const int Hd = 2 * p.Hq, Wd = 2 * WQ;
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
        const int q = wave * 32 + jq * 16 + col;
        const int img_l = q / (G::THQ * WQ), qi = q - img_l * (G::THQ * WQ);
        const int n = n0 + img_l, qy = y0 + qi / WQ, qx = qi % WQ;
        const bool valid = n < p.n_img;
        float add2[4][4];                                            // addend_mode 2: 0.25 x the quad's value, shared by the four classes
        if (p.addend_mode == 2 && valid) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint2 a = *(const uint2*)(p.addend + ((((long long)n * p.Hq + qy) * WQ + qx) * p.Cd + ct * 64 + i * 16 + g * 4) * 2);
                add2[i][0] = 0.25f * __uint_as_float(a.x << 16); add2[i][1] = 0.25f * __uint_as_float(a.x & 0xffff0000u);
                add2[i][2] = 0.25f * __uint_as_float(a.y << 16); add2[i][3] = 0.25f * __uint_as_float(a.y & 0xffff0000u);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int oy = 2 * qy + (c >> 1), ox = 2 * qx + (c & 1);
            const long long pix = ((long long)n * Hd + oy) * Wd + ox;
            unsigned pk[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[4] = {acc[i][c][jq][0], acc[i][c][jq][1], acc[i][c][jq][2], acc[i][c][jq][3]};
                if (p.addend_mode == 2) {
                    if (valid) { v[0] += add2[i][0]; v[1] += add2[i][1]; v[2] += add2[i][2]; v[3] += add2[i][3]; }
                } else if (p.addend_mode == 1 && valid) {
                    const uint2 a = *(const uint2*)(p.addend + (pix * p.Cd + ct * 64 + i * 16 + g * 4) * 2);
                    v[0] += __uint_as_float(a.x << 16); v[1] += __uint_as_float(a.x & 0xffff0000u);
                    v[2] += __uint_as_float(a.y << 16); v[3] += __uint_as_float(a.y & 0xffff0000u);
                }
                pk[i][0] = pack_bf16x2(v[0], v[1]); pk[i][1] = pack_bf16x2(v[2], v[3]);
            }
            // a lane holds channels 4g..4g+3 of each 16-channel fragment; after the row swap of a fragment pair it owns 8 consecutive
            // channels ({0, 16, 8, 24}[g] of the pair's 32): one 16-byte store per pair (all lanes swap, only the store is predicated)
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const s2d_u32x2 lo = __builtin_amdgcn_permlane16_swap(pk[i][0], pk[i + 1][0], false, false);
                const s2d_u32x2 hi = __builtin_amdgcn_permlane16_swap(pk[i][1], pk[i + 1][1], false, false);
                const int co = ct * 64 + i * 16 + (g & 1) * 16 + (g >> 1) * 8;
                if (valid) *(uint4*)(p.dst + (pix * p.Cd + co) * 2) = make_uint4(lo[0], hi[0], lo[1], hi[1]);
            }
        }
    }
#endif
}

// returns 1 if the kernel handled the call: bf16 input gradient of a 3x3 / stride-2 / pad-1 convolution onto a 32x32, 16x16 or 8x8 map
int fb_try_conv3x3s2_dgrad_quad(const fb_conv_args* a, hipStream_t st) {
    static const bool disabled = getenv("FB_DISABLE_S2_QUAD") != nullptr;
    if (disabled) return 0;
    if (a->mode != 1 || a->R != 3 || a->S != 3 || a->stride != 2 || a->pad != 1 || a->dtype != FB_BF16) return 0;
    if (a->Hs != a->Ws || a->Hd != 2 * a->Hs || a->Wd != 2 * a->Ws) return 0;
    const int WQ = a->Ws;
    if (WQ != 16 && WQ != 8 && WQ != 4) return 0;
    if (a->Cs % 32 != 0 || a->Cd % 64 != 0) return 0;
    if (a->wset_stride != 0 && a->imgs_per_wset > 0 && a->imgs_per_wset < a->n_img) return 0;          // one shared weight set
    if (a->addend_mask) return 0;
    const int imgs_per_tile = WQ == 16 ? 1 : (WQ == 8 ? 2 : 8);
    if ((long long)(imgs_per_tile * a->Hs * WQ) * a->Cs * 2 >= (1LL << 31) || (long long)a->Cd * 9 * a->Cs * 2 >= (1LL << 31)) return 0;
    S2DParams p;
    p.src = (const char*)a->src; p.wgt = (const char*)a->wgt; p.dst = (char*)a->dst; p.addend = (const char*)a->addend;
    p.n_img = a->n_img; p.Hq = a->Hs; p.Cs = a->Cs; p.Cd = a->Cd; p.addend_mode = a->addend ? a->addend_mode : 0;
    p.n_ct = a->Cd / 64;
    const int n_pt = WQ == 16 ? a->n_img * 2 : (a->n_img + imgs_per_tile - 1) / imgs_per_tile;
    p.n_tiles = n_pt * p.n_ct;
    dim3 grid(p.n_tiles);
    // One 49 KiB stage and TWO workgroups per CU (round 4) against two stages and one workgroup (round 3), 12 544 images, us, without / with
    // a full-resolution addend: dY 16x16 (128->64 ch) 1010 / 1707 -> 793 / 1227, 8x8 (256->128) 776 / 1117 -> 589 / 790, 4x4 (512->256)
    // 678 / 839 -> 506 / 612 (implicit GEMM: 649 / 778).  The tile's life is staging latency + a 128 KiB epilogue around 4-9 us of MFMA work:
    // a second resident workgroup covers both, a second stage covers neither.  FB_S2Q_STAGES=2 selects the round-3 form.
    const char* st_env = fb_getenv_experimental("FB_S2Q_STAGES");              // read per call: the tests compare the two forms in one process
    const int stages = st_env ? atoi(st_env) : 1;
    if (stages == 1) {
        if (WQ == 16) hipLaunchKernelGGL((conv3x3s2_dgrad_quad_kernel<16, 1>), grid, dim3(256), 0, st, p);
        else if (WQ == 8) hipLaunchKernelGGL((conv3x3s2_dgrad_quad_kernel<8, 1>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3s2_dgrad_quad_kernel<4, 1>), grid, dim3(256), 0, st, p);
    } else {
        if (WQ == 16) hipLaunchKernelGGL((conv3x3s2_dgrad_quad_kernel<16, 2>), grid, dim3(256), 0, st, p);
        else if (WQ == 8) hipLaunchKernelGGL((conv3x3s2_dgrad_quad_kernel<8, 2>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3s2_dgrad_quad_kernel<4, 2>), grid, dim3(256), 0, st, p);
    }
    return 1;
}
