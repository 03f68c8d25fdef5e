// Weight gradient of 1x1 convolutions, bf16:  dW[co][ci] = sum over the pixels of a chunk of dY[p][co] * X[p][ci]
// (the Bottleneck blocks' channel-changing convolutions and the downsample shortcuts, reference resnets.py:150,289-291).
//
// A GEMM whose contraction runs over PIXELS with both operands stored pixel-major, and -- unlike the 3x3 case, where one staged halo
// feeds nine taps -- nothing to reuse but the output tile: the per-tap kernel (conv_wgrad.hip, 128 x 128 tiles, register staging) moves
// 16 KiB per 32 MFMAs and sits at 220-490 TFLOP/s.  Here:
//   * output tiles of up to 256 x 256 (co x ci) per workgroup, 128 x 128 per wave (256 accumulator registers, one wave per SIMD): 64 KiB
//     of operands per 512 MFMAs, half of the L2 -> LDS bytes per FLOP of the 128 x 128 tile
//   * operands go HBM/L2 -> LDS by `buffer_load ... lds` in 64-pixel steps, double buffered, as 64-channel sub-tiles of 128-byte rows with
//     the 32-byte-slot swizzle of the all-taps 3x3 kernel; fragments by the transposed read `ds_read_b64_tr_b16`
//   * split-K over pixel ranges of a chunk (fp32 slabs, fixed-order reduction by fb_wgrad_reduce: deterministic), XCD-aware work order
#include "common.h"

#include <type_traits>

struct Wgrad1Params {
    const char* x; const char* dy; float* out;
    long long n_px;                      // pixels of the whole launch
    int Cs, Cd, px_per_group, px_per_split, split_k, n_groups;
    long long group_stride;
};

namespace {
typedef __attribute__((ext_vector_type(2))) unsigned w1_u32x2;
template <int OFF> __device__ __forceinline__ w1_u32x2 w1_read_tr(unsigned byte_addr) {
    w1_u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
template <int N> __device__ __forceinline__ void w1_wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> __device__ __forceinline__ void w1_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int I, int N, typename F> __device__ __forceinline__ void w1_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); w1_static_for<I + 1, N>(f); }
}
__device__ __forceinline__ uint4 w1_join(w1_u32x2 lo, w1_u32x2 hi) { return make_uint4(lo[0], lo[1], hi[0], hi[1]); }
__device__ __forceinline__ int w1_f(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }
constexpr unsigned W1_OOB = 0x80000000u;
}  // namespace

// wave tile: 16 MI output channels x 16 NJ input channels; 2 x 2 waves per workgroup
template <int MI, int NJ>
__global__ __launch_bounds__(256) void conv_wgrad1x1_kernel(const Wgrad1Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int CO_T = 32 * MI, CI_T = 32 * NJ;
    constexpr int SA = (CO_T + 63) / 64, SB = (CI_T + 63) / 64;          // 64-channel sub-tiles (8 KiB: 64 pixel rows of 128 bytes)
    constexpr int STAGE = (SA + SB) * 8192;
    constexpr int NDMA = (SA + SB) * 8;                                     // 1 KiB LDS-DMA instructions per step
    constexpr int KD = (NDMA + 3) / 4;
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = p.Cs / CI_T, tiles = tiles_n * (p.Cd / CO_T);
    int item;
    {
        const int n_items = gridDim.x, b = blockIdx.x, q = n_items >> 3, r = n_items & 7, xcd = b & 7, slot = b >> 3;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;      // consecutive items (tiles of one pixel range) on one XCD
    }
    const int tile = item % tiles, gs = item / tiles;
    const int tile_m = __builtin_amdgcn_readfirstlane(tile / tiles_n), tile_n = __builtin_amdgcn_readfirstlane(tile % tiles_n);
    const int group = __builtin_amdgcn_readfirstlane(gs / p.split_k), split = __builtin_amdgcn_readfirstlane(gs % p.split_k);
    const long long px0 = (long long)group * p.px_per_group + (long long)split * p.px_per_split;
    int n_px = p.px_per_group - split * p.px_per_split;
    n_px = n_px < 0 ? 0 : (n_px > p.px_per_split ? p.px_per_split : n_px);
    const int n_steps = (n_px + 63) / 64;
    const int rowA_b = p.Cd * 2, rowB_b = p.Cs * 2;

    // the pixel range of this workgroup as buffers of its own: rows past the range read zeros
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dy + px0 * rowA_b), 0, n_px * rowA_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + px0 * rowB_b), 0, n_px * rowB_b, 0x00020000);
    // LDS-DMA instruction d = wave + 4k: sub-tile d / 8 (A sub-tiles first), rows 8 (d % 8) .. + 7; lane -> row lane >> 3, 16-byte half
    // lane & 1 of the 32-byte slot ((lane & 7) >> 1) ^ f(row)
    unsigned voff[KD]; int rowk[KD];
#pragma unroll
    for (int k = 0; k < KD; ++k) {
        const int d = wave + 4 * k, st = d >> 3, row = (d & 7) * 8 + (lane >> 3);
        const int lslot = ((lane & 7) >> 1) ^ w1_f(row);
        const bool a = st < SA;
        const int ch0 = a ? tile_m * CO_T + st * 64 : tile_n * CI_T + (st - SA) * 64;
        rowk[k] = row;
        voff[k] = (unsigned)(row * (a ? rowA_b : rowB_b) + ch0 * 2 + lslot * 32 + (lane & 1) * 16);
    }
    auto issue = [&](int stage, int step) {
        char* base = lds + stage * STAGE;
        const int left = n_px - step * 64;                   // rows of this step inside the pixel range (the scalar offset is not range checked)
#pragma unroll
        for (int k = 0; k < KD; ++k) {
            const int d = wave + 4 * k;                       // wave-uniform
            if (d < NDMA) {
                const unsigned v = rowk[k] < left ? voff[k] : W1_OOB;
                if ((d >> 3) < SA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(base + d * 1024), 16, v, step * 64 * rowA_b, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (__attribute__((address_space(3))) void*)(base + d * 1024), 16, v, step * 64 * rowB_b, 0, 0);
            }
        }
    };

    // fragment read addresses: pixel pl of a 32-pixel half (second read: pixel + 4 = +512 bytes), 16-channel slot i of a 64-channel sub-tile
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int t = lane & 15, g = lane >> 4;
    const int pl = g * 8 + (t >> 2);
    // this wave's co fragment i = fragment wm * MI + i of the tile (sub-tile fr >> 2, slot fr & 3), ci fragment j likewise behind the A sub-tiles
    unsigned la[MI], lb[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int fr = wm * MI + i;
        la[i] = lds0 + (fr >> 2) * 8192 + pl * 128 + (((fr & 3) ^ w1_f(pl)) * 32) + (t & 3) * 8;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int fr = wn * NJ + j;
        lb[j] = lds0 + (SA + (fr >> 2)) * 8192 + pl * 128 + (((fr & 3) ^ w1_f(pl)) * 32) + (t & 3) * 8;
    }

    f32x4_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    if (n_steps > 0) {
        issue(0, 0);
        w1_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }
    for (int step = 0; step < n_steps; ++step) {
        const int cur = step & 1;
        if (step + 1 < n_steps) issue(cur ^ 1, step + 1);
        const unsigned so = cur * STAGE;
        w1_static_for<0, 2>([&](auto hc) {                                 // two 32-pixel halves of the step
            constexpr int HB = decltype(hc)::value * 4096;
            uint4 af[MI];
            w1_static_for<0, MI>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                af[I] = w1_join(w1_read_tr<HB>(la[I] + so), w1_read_tr<HB + 512>(la[I] + so));
            });
            w1_static_for<0, (NJ + 3) / 4>([&](auto jbc) {
                constexpr int JB = decltype(jbc)::value * 4;
                constexpr int JN = NJ - JB < 4 ? NJ - JB : 4;
                uint4 bf[JN];
                w1_static_for<0, JN>([&](auto jc) {
                    constexpr int J = decltype(jc)::value;
                    bf[J] = w1_join(w1_read_tr<HB>(lb[JB + J] + so), w1_read_tr<HB + 512>(lb[JB + J] + so));
                });
                w1_wait_lgkmcnt<0>();
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < JN; ++j) acc[i][JB + j] = mma_chunk<bf16_tag>(af[i], bf[j], acc[i][JB + j]);
            });
        });
        w1_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }

    // fp32 slab [co][ci] of this (group, split)
    float* out = p.out + group * p.group_stride + (long long)split * p.Cd * p.Cs;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = tile_m * CO_T + (wm * MI + i) * 16 + (lane >> 4) * 4 + q;
                const int ci = tile_n * CI_T + (wn * NJ + j) * 16 + (lane & 15);
                out[(long long)co * p.Cs + ci] = acc[i][j][q];
            }
#endif
}

template <int MI, int NJ> static void w1_launch(const Wgrad1Params& p, hipStream_t st) {
    const int grid = (p.Cd / (32 * MI)) * (p.Cs / (32 * NJ)) * p.n_groups * p.split_k;
    hipLaunchKernelGGL((conv_wgrad1x1_kernel<MI, NJ>), dim3(grid), dim3(256), 0, st, p);
}

// returns 1 if handled: bf16, 1x1, stride 1, unpadded channel counts that are multiples of 64 (one side at least 128)
int fb_try_wgrad1x1(const fb_wgrad_args* a, hipStream_t st) {
    static const bool disabled = getenv("FB_DISABLE_WGRAD1X1") != nullptr;
    if (disabled || a->dtype != FB_BF16) return 0;
    if (a->R != 1 || a->S != 1 || a->stride != 1 || a->pad != 0 || a->Hs != a->Hd || a->Ws != a->Wd) return 0;
    if (a->Cs % 64 != 0 || a->Cd % 64 != 0 || (a->Cs < 128 && a->Cd < 128)) return 0;
    const long long n_px = (long long)a->n_img * a->Hd * a->Wd;
    const long long px_per_group = (long long)a->imgs_per_group * a->Hd * a->Wd;
    if (n_px * a->Cs * 2 >= (1LL << 40) || px_per_group * (a->Cs > a->Cd ? a->Cs : a->Cd) * 2 >= (1LL << 31)) return 0;
    Wgrad1Params p;
    p.x = (const char*)a->x; p.dy = (const char*)a->dy; p.out = a->dw_partial;
    p.n_px = n_px; p.Cs = a->Cs; p.Cd = a->Cd;
    p.px_per_group = (int)px_per_group; p.split_k = a->split_k; p.n_groups = a->n_img / a->imgs_per_group;
    p.px_per_split = (int)(ceil_div64(ceil_div64(px_per_group, a->split_k), 64) * 64);
    p.group_stride = a->group_stride ? a->group_stride : (long long)a->split_k * a->Cd * a->Cs;
    // wave tile 16 MI x 16 NJ with MI, NJ in {2, 4, 8}: the largest that divides the layer (tile = 64 / 128 / 256 channels per side)
    int mi = a->Cd % 256 == 0 ? 8 : (a->Cd % 128 == 0 ? 4 : 2), nj = a->Cs % 256 == 0 ? 8 : (a->Cs % 128 == 0 ? 4 : 2);
    // ... capped at 128 x 128 per workgroup (two workgroups per CU, 64 KiB of LDS each) since the end of round 5: the 256 x 256 tile (one wave per SIMD, 128 KiB) halves the
    // L2 -> LDS bytes per FLOP and is no slower alone, but inside the ResNet-152 @224 step the small tile wins (6514 against 6453-6473 images/s; ResNet-50 @224
    // +0.25 %, the ResNet-18 headline unchanged, ResNet-50 @32 beside the weight-gradient stream -0.3 %).  FB_W1_CAP_M / FB_W1_CAP_N = 8 bring the big tile back.
    const int cap_m = fb_getenv_experimental("FB_W1_CAP_M") ? atoi(fb_getenv_experimental("FB_W1_CAP_M")) : 4, cap_n = fb_getenv_experimental("FB_W1_CAP_N") ? atoi(fb_getenv_experimental("FB_W1_CAP_N")) : 4;      // (read per call: the tests compare)
    if (mi > cap_m) mi = cap_m;
    if (nj > cap_n) nj = cap_n;
    if (mi == 8 && nj == 8) w1_launch<8, 8>(p, st);
    else if (mi == 8 && nj == 4) w1_launch<8, 4>(p, st);
    else if (mi == 4 && nj == 8) w1_launch<4, 8>(p, st);
    else if (mi == 8 && nj == 2) w1_launch<8, 2>(p, st);
    else if (mi == 2 && nj == 8) w1_launch<2, 8>(p, st);
    else if (mi == 4 && nj == 4) w1_launch<4, 4>(p, st);
    else if (mi == 4 && nj == 2) w1_launch<4, 2>(p, st);
    else if (mi == 2 && nj == 4) w1_launch<2, 4>(p, st);
    else return 0;
    return 1;
}
