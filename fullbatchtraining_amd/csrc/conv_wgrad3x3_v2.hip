// Weight gradient of 3x3 / stride 1 / pad 1 convolutions on 32x32 / 16x16 maps, bf16, all nine taps from one LDS halo --
// low-overhead version of conv_wgrad3x3.hip (same decomposition: 64 co x 64 ci x 9 taps per workgroup, split-K over whole
// images, wave w owns ci block [16w,16w+16) for all taps and 64 co; K-step = 64 output pixels).
//   * dY rows and the X halo go HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds` (zero padding = out-of-range voffset),
//     double-buffered: the loads of step s+1 are in flight while step s is multiplied
//   * 128-byte pixel rows, 32-byte-slot XOR swizzle  slot' = slot ^ ((row>>1)&1 | ((row>>3)&1)<<1)  applied on the source
//     side: the four rows of a transposed 4x16 read and the two lane groups of a half wave hit eight distinct bank octets
//   * halo pitch 48 (W=32) / 32 (W=16) rows keeps bits 1 and 3 of the row index independent of the vertical tap, so every
//     ds_read_b64_tr_b16 address is (one of 4+3 precomputed lane registers) + immediate: no address arithmetic in the loop
#include "common.h"

#include <type_traits>

struct Wgrad3V2Params {
    const char* x; const char* dy; float* out;
    int n_img, H, Cs, Cd;
    int imgs_per_group, imgs_per_block, split_k;
};

namespace {
typedef __attribute__((ext_vector_type(2))) unsigned w3_u32x2;
template <int OFF> __device__ __forceinline__ w3_u32x2 w3_read_tr(unsigned byte_addr) {
    w3_u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
template <int N> __device__ __forceinline__ void w3_wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> __device__ __forceinline__ void w3_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int I, int N, typename F> __device__ __forceinline__ void w3_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); w3_static_for<I + 1, N>(f); }
}
__device__ __forceinline__ uint4 w3_join(w3_u32x2 lo, w3_u32x2 hi) { return make_uint4(lo[0], lo[1], hi[0], hi[1]); }
__device__ __forceinline__ int w3_f(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }
constexpr unsigned W3_OOB = 0x80000000u;
}  // namespace

template <int W>
__global__ __launch_bounds__(256) void conv_wgrad3x3_v2_kernel(const Wgrad3V2Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int RS = 64 / W;                          // image rows per K-step
    constexpr int PITCH = W == 32 ? 48 : 32;            // halo row pitch (multiple of 16)
    constexpr int HROWS = (RS + 2) * PITCH;             // 192 for both widths
    constexpr int A_BYTES = 64 * 128, B_BYTES = HROWS * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int NGB = HROWS / 8;                      // halo row groups (1 KiB each)
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = p.Cs / 64;
    const int tile_m = __builtin_amdgcn_readfirstlane(blockIdx.x / tiles_n), tile_n = __builtin_amdgcn_readfirstlane(blockIdx.x % tiles_n);
    const int group = __builtin_amdgcn_readfirstlane(blockIdx.y / p.split_k), split = __builtin_amdgcn_readfirstlane(blockIdx.y % p.split_k);
    const int img0 = group * p.imgs_per_group + split * p.imgs_per_block;
    const int img_end = min(img0 + p.imgs_per_block, (group + 1) * p.imgs_per_group);
    const int steps_per_img = p.H / RS;
    const int n_steps = (img_end - img0) * steps_per_img;
    const int rowA_b = p.Cd * 2, rowB_b = p.Cs * 2;    // bytes per pixel of dY / X
    const int lrow8 = lane >> 3;

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.n_img * p.H * W * rowA_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.n_img * p.H * W * rowB_b, 0x00020000);
    // per-lane source offsets (relative to the first pixel row of the step); source-side slot swizzle
    unsigned voffA[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int row = (wave + 4 * k) * 8 + lrow8;                          // pixel of the step
        const int pslot = (lane & 7) >> 1, lslot = pslot ^ w3_f(row);
        voffA[k] = (unsigned)(row * rowA_b + tile_m * 128 + lslot * 32 + (lane & 1) * 16);
    }
    constexpr int KB = (NGB + 3) / 4;
    int hyB[KB]; unsigned voffB[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const int row = (wave + 4 * k) * 8 + lrow8;                          // halo row index (pitch PITCH)
        const int hy = row / PITCH, hx = row - hy * PITCH;
        const int pslot = (lane & 7) >> 1, lslot = pslot ^ w3_f(row);
        const bool xok = hx >= 1 && hx <= W && (wave + 4 * k) < NGB;
        hyB[k] = xok ? hy - 1 : -(1 << 20);
        voffB[k] = (unsigned)((hx - 1) * rowB_b + tile_n * 128 + lslot * 32 + (lane & 1) * 16);   // + (hy-1)*W*rowB_b when valid
    }
    auto issue = [&](int stage, int step) {
        const int img = img0 + step / steps_per_img, y0 = (step % steps_per_img) * RS;
        char* base = lds + stage * STAGE;
        const int soffA = (img * p.H + y0) * W * rowA_b;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(base + (wave + 4 * k) * 1024), 16,
                                                     voffA[k], soffA, 0, 0);
        const int soffB = img * p.H * W * rowB_b;
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            if (wave + 4 * k < NGB) {
                const int sy = y0 + hyB[k];
                const unsigned v = (unsigned)sy < (unsigned)p.H ? voffB[k] + (unsigned)(sy * W * rowB_b) : W3_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (__attribute__((address_space(3))) void*)(base + A_BYTES + (wave + 4 * k) * 1024),
                                                         16, v, soffB, 0, 0);
            }
        }
    };

    // ---- precomputed fragment read addresses -------------------------------------------------------------------------------
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int t = lane & 15, g = lane >> 4;
    // A: row = pb + g*8 + (t>>2) (+4); slot = i ^ f(row)   (f is independent of pb (multiple of 32) and of the +4)
    unsigned la[4];
    {
        const int row = g * 8 + (t >> 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) la[i] = lds0 + row * 128 + ((i ^ w3_f(row)) * 32) + (t & 3) * 8;
    }
    // B: halo row = (ky + r)*PITCH + kx + s ; lane part = ky_l*PITCH + kx_l ; slot = wave ^ f(kx_l + s)
    unsigned lb[3], lbh[3];                              // rows kx..kx+3 and kx+4..kx+7 (the +4 can carry into bit 3)
    {
        const int pl = g * 8 + (t >> 2);                 // pixel within the 32-pixel block
        const int ky_l = pl / W, kx_l = pl % W;          // W=32: ky_l = 0
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int row = ky_l * PITCH + kx_l + s;
            lb[s] = lds0 + A_BYTES + row * 128 + ((wave ^ w3_f(row)) * 32) + (t & 3) * 8;
            lbh[s] = lds0 + A_BYTES + (row + 4) * 128 + ((wave ^ w3_f(row + 4)) * 32) + (t & 3) * 8;
        }
    }

    f32x4_t acc[9][4];
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[u][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    if (n_steps > 0) {
        issue(0, 0);
        w3_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }
    for (int step = 0; step < n_steps; ++step) {
        const int cur = step & 1;
        if (step + 1 < n_steps) issue(cur ^ 1, step + 1);
        const unsigned so = cur * STAGE;
        unsigned a0[4], b0[3], b1[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) a0[i] = la[i] + so;
#pragma unroll
        for (int s = 0; s < 3; ++s) { b0[s] = lb[s] + so; b1[s] = lbh[s] + so; }
        w3_static_for<0, 2>([&](auto blkc) {
            constexpr int BLK = decltype(blkc)::value;
            constexpr int PB = BLK * 32;                                  // first pixel of the 32-pixel block
            constexpr int KY = PB / W;                                    // image row of the block within the step (W=16: first of two)
            uint4 af[4];
            w3_static_for<0, 4>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                af[I] = w3_join(w3_read_tr<PB * 128>(a0[I]), w3_read_tr<PB * 128 + 512>(a0[I]));
            });
            // the LGKM counter holds 15 outstanding operations: read in three batches of <= 14
            uint4 bf[9];
            w3_static_for<0, 3>([&](auto bc) {
                constexpr int B0 = decltype(bc)::value * 3;
                w3_static_for<B0, B0 + 3>([&](auto uc) {
                    constexpr int U = decltype(uc)::value, R = U / 3, S = U % 3;
                    constexpr int OFF = (KY + R) * PITCH * 128;
                    bf[U] = w3_join(w3_read_tr<OFF>(b0[S]), w3_read_tr<OFF>(b1[S]));
                });
                w3_wait_lgkmcnt<0>();
#pragma unroll
                for (int u = B0; u < B0 + 3; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[u][i] = mma_chunk<bf16_tag>(af[i], bf[u], acc[u][i]);
            });
        });
        w3_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }

    float* out = p.out + ((long long)(group * p.split_k + split) * p.Cd) * 9 * p.Cs;
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = tile_m * 64 + i * 16 + (lane >> 4) * 4 + q;
                const int ci = tile_n * 64 + wave * 16 + (lane & 15);
                out[((long long)co * 9 + u) * p.Cs + ci] = acc[u][i][q];
            }
#endif
}

// returns 1 if handled (bf16 only; f32 stays on conv_wgrad3x3.hip)
int fb_try_wgrad3x3_v2(const fb_wgrad_args* a, hipStream_t st) {
    static const bool disabled = getenv("FB_DISABLE_WGRAD3_V2") != nullptr;
    if (disabled || a->dtype != FB_BF16) return 0;
    if (a->R != 3 || a->S != 3 || a->stride != 1 || a->pad != 1) return 0;
    if (a->Hs != a->Hd || a->Ws != a->Wd || a->Hs != a->Ws) return 0;
    const int W = a->Ws;
    if (W != 32 && W != 16) return 0;
    if (a->Cs % 64 != 0 || a->Cd % 64 != 0) return 0;
    if (a->imgs_per_group % a->split_k != 0) return 0;
    const long long bytes = (long long)a->n_img * a->Hs * W * (a->Cs > a->Cd ? a->Cs : a->Cd) * 2;
    if (bytes >= (1LL << 31)) return 0;
    Wgrad3V2Params p;
    p.x = (const char*)a->x; p.dy = (const char*)a->dy; p.out = a->dw_partial;
    p.n_img = a->n_img; p.H = a->Hs; p.Cs = a->Cs; p.Cd = a->Cd;
    p.imgs_per_group = a->imgs_per_group; p.split_k = a->split_k; p.imgs_per_block = a->imgs_per_group / a->split_k;
    const int n_groups = a->n_img / a->imgs_per_group;
    dim3 grid((a->Cd / 64) * (a->Cs / 64), n_groups * a->split_k);
    if (W == 32) hipLaunchKernelGGL((conv_wgrad3x3_v2_kernel<32>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_wgrad3x3_v2_kernel<16>), grid, dim3(256), 0, st, p);
    return 1;
}
