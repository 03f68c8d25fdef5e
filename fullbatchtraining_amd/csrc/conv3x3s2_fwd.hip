// Forward 3x3 / stride-2 / pad-1 convolution (first convolution of a downsampling block) from an LDS-resident input halo.
//
//   out[oy][ox][co] = sum_{r,s} X[2 oy + r - 1][2 ox + s - 1][:] . W[co][r][s][:]
//
// The implicit-GEMM kernel gathers the rows of every tap separately and lives from one 128-byte K-step to the next (one barrier and one
// L2 round trip per 32 MFMAs): 474-837 TF/s on the three shapes of ResNet-18.  Here a workgroup owns 128 output pixels x 64 channels:
//   * per 32-channel half-slice the (2 rows + 1) x (2 W + 1) input halo of the tile (<= 648 rows of 64 bytes) and the 64-byte weight
//     rows of all nine taps (36 KiB) are staged once by LDS-DMA (double buffered) and feed 72 MFMAs per wave between two barriers
//   * the halo is stored DE-INTERLEAVED: the even columns of a halo line first, then the odd ones (the DMA places every 64-byte row
//     wherever its lane says).  A 16-pixel fragment of a tap reads columns 2 ox + s, all of one parity: consecutive LDS rows, one
//     contiguous KiB per fragment read instead of a 128-byte stride (which is an 8-way bank conflict for ds_read_b128)
//   * a wave owns 32 pixels x 64 channels; fragment reads of tap T+1 are issued before the MFMAs of tap T (one wave per SIMD)
//   * epilogue: 16-byte stores after the v_permlane16_swap row exchange, BatchNorm partial sums of the tile (= one 128-pixel
//     statistics block) by DPP row sums + one LDS hand-over between the waves
#include "common.h"

#include <type_traits>

struct S2FParams {
    const char* src; const char* wgt; char* dst; float* stat;
    int n_img, Ho, Cs, Cd, n_ct, n_tiles, n_mblocks;
};

namespace {
typedef __attribute__((ext_vector_type(4))) unsigned s2f_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned s2f_u32x2;
template <int N> __device__ __forceinline__ void s2f_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void s2f_wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int OFF> __device__ __forceinline__ uint4 s2f_read16(unsigned byte_addr) {
    s2f_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
template <int I, int N, typename F> __device__ __forceinline__ void s2f_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); s2f_static_for<I + 1, N>(f); }
}
__device__ __forceinline__ int s2f_xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, xcd = b & 7, slot = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
constexpr unsigned S2F_OOB = 0x80000000u;

template <int WO> struct S2FGeo {                                  // WO = output width: 16, 8, 4 (input 32, 16, 8)
    static constexpr int THO = WO == 16 ? 8 : WO;                   // output rows per image part of a 128-pixel tile
    static constexpr int IMGS = 128 / (THO * WO);                   // 1, 2, 8
    static constexpr int PITCH = 2 * WO + 1;                        // halo columns -1 .. 2 WO - 1
    static constexpr int IMG_ROWS = (2 * THO + 1) * PITCH;          // 561, 289, 81
    static constexpr int ROWS = IMGS * IMG_ROWS;                    // 561, 578, 648
    static constexpr int NGRP = (ROWS + 15) / 16;
};
}  // namespace

template <int WO>
__global__ __launch_bounds__(256) void conv3x3s2_fwd_kernel(const S2FParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = S2FGeo<WO>;
    constexpr int PITCH = G::PITCH, NGRP = G::NGRP;
    constexpr int HALO_BYTES = NGRP * 1024, WT_BYTES = 9 * 4096, STAGE_BYTES = HALO_BYTES + WT_BYTES;
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int L = s2f_xcd_remap(blockIdx.x, gridDim.x);
    const int pt = __builtin_amdgcn_readfirstlane(L / p.n_ct), ct = __builtin_amdgcn_readfirstlane(L % p.n_ct);
    const int n0 = WO == 16 ? pt / 2 : pt * G::IMGS;
    const int y0 = WO == 16 ? (pt & 1) * G::THO : 0;               // first output row of the tile
    const int Hi = 2 * p.Ho, Wi = 2 * WO;
    const int row_b = p.Cs * 2;
    const int n_cc = row_b / 64;

    // ---- halo DMA: LDS row `row` of an image part holds halo line hy = pos / PITCH; within a line the even columns come first ------
    const int drow = lane >> 2;
    constexpr int KH = (NGRP + 3) / 4;
    unsigned voffH[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k) {
        const int row = (wave + 4 * k) * 16 + drow;
        const int img_l = row / G::IMG_ROWS, rr = row - img_l * G::IMG_ROWS;
        const int hy = rr / PITCH, pos = rr - hy * PITCH;
        const int hx = pos <= WO ? 2 * pos : 2 * (pos - WO - 1) + 1;              // halo column (0 = input column -1)
        const int sy = 2 * y0 + hy - 1, sx = hx - 1;
        const bool ok = row < G::ROWS && (unsigned)sy < (unsigned)Hi && (unsigned)sx < (unsigned)Wi && n0 + img_l < p.n_img;
        voffH[k] = ok ? (unsigned)(((img_l * Hi + sy) * Wi + sx) * row_b + (lane & 3) * 16) : S2F_OOB;
    }
    // origin = pixel (n0, 0, 0) of the input; a tile spans at most IMGS images
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + (long long)n0 * Hi * Wi * row_b), 0,
                                                                           (G::IMGS * Hi * Wi) * row_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, p.Cd * 9 * row_b, 0x00020000);
    const unsigned voffW = (unsigned)(((ct * 64 + wave * 16 + drow) * 9) * row_b + (lane & 3) * 16);
    auto issue = [&](int stage, int cc) {
        char* base = lds + stage * STAGE_BYTES;
        const int soff = cc * 64;
        s2f_static_for<0, KH>([&](auto kc) {
            constexpr int K = decltype(kc)::value;
            const int grp = wave + 4 * K;
            if (grp < NGRP)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(base + grp * 1024), 16, voffH[K], soff, 0, 0);
        });
        s2f_static_for<0, 9>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)(base + HALO_BYTES + T * 4096 + wave * 1024), 16,
                                                     voffW, T * row_b + soff, 0, 0);
        });
    };

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned wa = lds0 + HALO_BYTES + col * 64 + g * 16;
    unsigned pa[2];                                                  // pixel fragment j of this wave: pixels 32 wave + 16 j + col
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = wave * 32 + j * 16 + col;
        const int img_l = q / (G::THO * WO), qi = q - img_l * (G::THO * WO);
        pa[j] = lds0 + (img_l * G::IMG_ROWS + 2 * (qi / WO) * PITCH + qi % WO) * 64 + g * 16;     // halo line 2 oy_l, even-column slot ox
    }

    f32x4_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    issue(0, 0);
    s2f_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    for (int cc = 0; cc < n_cc; ++cc) {
        const int cur = cc & 1;
        if (cc + 1 < n_cc) issue(cur ^ 1, cc + 1);
        const unsigned so = cur * STAGE_BYTES;
        const unsigned w0 = wa + so, p0 = pa[0] + so, p1 = pa[1] + so;
        uint4 wf[2][4], pf[2][2];
        auto read_tap = [&](auto tc, int set) {
            constexpr int T = decltype(tc)::value, R = T / 3, S = T % 3;
            // column 2 ox + S: S = 0 -> even slot ox, S = 1 -> odd slot ox (after the WO + 1 even ones), S = 2 -> even slot ox + 1
            constexpr int SLOT = S == 0 ? 0 : (S == 1 ? WO + 1 : 1);
            s2f_static_for<0, 4>([&](auto i) { wf[set][decltype(i)::value] = s2f_read16<T * 4096 + decltype(i)::value * 1024>(w0); });
            pf[set][0] = s2f_read16<(R * PITCH + SLOT) * 64>(p0);
            pf[set][1] = s2f_read16<(R * PITCH + SLOT) * 64>(p1);
        };
        read_tap(std::integral_constant<int, 0>{}, 0);
        s2f_static_for<0, 9>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            if constexpr (T < 8) { read_tap(std::integral_constant<int, T + 1>{}, (T + 1) & 1); s2f_wait_lgkmcnt<6>(); }
            else s2f_wait_lgkmcnt<0>();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma_chunk<bf16_tag>(wf[T & 1][i], pf[T & 1][j], acc[i][j]);
        });
        s2f_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: outputs + per-channel partial statistics of the tile's 128 pixels ----------------------------------------------
    float ssum[4][4], ssq[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = wave * 32 + j * 16 + col;
        const int img_l = q / (G::THO * WO), qi = q - img_l * (G::THO * WO);
        const int n = n0 + img_l;
        const bool valid = n < p.n_img;                              // (pixels of images past the end accumulated zeros only)
        const long long pix = ((long long)n * p.Ho + y0 + qi / WO) * WO + qi % WO;
        unsigned pk[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pk[i][0] = pack_bf16x2(acc[i][j][0], acc[i][j][1]); pk[i][1] = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[i][r] += acc[i][j][r]; ssq[i][r] += acc[i][j][r] * acc[i][j][r]; }
        }
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const s2f_u32x2 lo = __builtin_amdgcn_permlane16_swap(pk[i][0], pk[i + 1][0], false, false);
            const s2f_u32x2 hi = __builtin_amdgcn_permlane16_swap(pk[i][1], pk[i + 1][1], false, false);
            const int co = ct * 64 + i * 16 + (g & 1) * 16 + (g >> 1) * 8;
            if (valid) *(uint4*)(p.dst + (pix * p.Cd + co) * 2) = make_uint4(lo[0], hi[0], lo[1], hi[1]);
        }
    }
    if (p.stat != nullptr) {
        float* red = (float*)lds;                                    // [4 waves][64 co][2]; every wave is past its last fragment read
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = row16_sum(ssum[i][r]), b = row16_sum(ssq[i][r]);
                if (col == 0) { red[(wave * 64 + i * 16 + g * 4 + r) * 2] = a; red[(wave * 64 + i * 16 + g * 4 + r) * 2 + 1] = b; }
            }
        __syncthreads();
        if (tid < 64) {
            const float a = ((red[tid * 2] + red[(64 + tid) * 2]) + red[(128 + tid) * 2]) + red[(192 + tid) * 2];
            const float b = ((red[tid * 2 + 1] + red[(64 + tid) * 2 + 1]) + red[(128 + tid) * 2 + 1]) + red[(192 + tid) * 2 + 1];
            p.stat[(long long)pt * p.Cd + ct * 64 + tid] = a;
            p.stat[((long long)p.n_mblocks + pt) * p.Cd + ct * 64 + tid] = b;
        }
    }
#endif
}

// returns 1 if the kernel handled the call: bf16 forward 3x3 / stride-2 / pad-1 from a 32x32, 16x16 or 8x8 map, one shared weight set
int fb_try_conv3x3s2_fwd(const fb_conv_args* a, hipStream_t st) {
    static const bool disabled = getenv("FB_DISABLE_S2_FWD") != nullptr;
    if (disabled) return 0;
    if (a->mode != 0 || a->R != 3 || a->S != 3 || a->stride != 2 || a->pad != 1 || a->dtype != FB_BF16 || a->addend) return 0;
    if (a->Hs != a->Ws || a->Hs != 2 * a->Hd || a->Ws != 2 * a->Wd) return 0;
    const int WO = a->Wd;
    if (WO != 16 && WO != 8 && WO != 4) return 0;
    if (a->Cs % 32 != 0 || a->Cd % 64 != 0) return 0;
    if (a->wset_stride != 0 && a->imgs_per_wset > 0 && a->imgs_per_wset < a->n_img) return 0;
    const int imgs_per_tile = WO == 16 ? 1 : (WO == 8 ? 2 : 8);
    if (a->n_img % imgs_per_tile != 0) return 0;                   // tiles = whole 128-pixel statistics blocks
    if ((long long)(imgs_per_tile * a->Hs * a->Ws) * a->Cs * 2 >= (1LL << 31) || (long long)a->Cd * 9 * a->Cs * 2 >= (1LL << 31)) return 0;
    S2FParams p;
    p.src = (const char*)a->src; p.wgt = (const char*)a->wgt; p.dst = (char*)a->dst; p.stat = a->stat_partial;
    p.n_img = a->n_img; p.Ho = a->Hd; p.Cs = a->Cs; p.Cd = a->Cd;
    p.n_ct = a->Cd / 64;
    const int n_pt = WO == 16 ? a->n_img * 2 : a->n_img / imgs_per_tile;
    p.n_tiles = n_pt * p.n_ct;
    p.n_mblocks = n_pt;                                            // = ceil(M / 128): a tile is one statistics block
    dim3 grid(p.n_tiles);
    if (WO == 16) hipLaunchKernelGGL((conv3x3s2_fwd_kernel<16>), grid, dim3(256), 0, st, p);
    else if (WO == 8) hipLaunchKernelGGL((conv3x3s2_fwd_kernel<8>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv3x3s2_fwd_kernel<4>), grid, dim3(256), 0, st, p);
    return 1;
}
