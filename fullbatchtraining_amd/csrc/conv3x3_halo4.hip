// 3x3 / stride 1 / pad 1 convolution (forward and input-gradient) from an LDS-resident input halo: persistent workgroups.
//
// Tile decomposition: 256 output pixels x 64 output channels per tile; the 256 pixels are contiguous in memory -- 8 full
// rows of one 32x32 image, one 16x16 image, or four 8x8 images.  Per 128-byte input-channel slice the input halo is staged
// once in LDS and feeds all nine taps; wave w owns pixels [64w, 64w+64) x 64 channels (16 accumulator fragments).
//
// A timeline of the one-tile-per-workgroup version (tools/h4_trace.hip) showed 9 of the 14 us a 64-channel tile lives being
// spent outside the MFMA loop: ~1000 instructions of address set-up per workgroup, the halo round trip, 128 LDS shuffles for
// the BN statistics.  Here 2 x #CU workgroups stay resident and walk the tile list:
//   * all per-lane addresses (halo sources, fragment reads, weight rows) are computed once per workgroup; per tile only
//     scalar work remains (two buffer descriptors, a scalar offset, the top/bottom-row validity of 32x32 tiles)
//   * the next tile's first halo slice and its first two weight taps are requested before the epilogue of the current tile,
//     so their round trip overlaps the output stores and the statistics
//   * weights stream through a ring of three 8 KiB buffers, two taps ahead (9 taps = 3 ring turns: the buffer of tap U is the
//     compile-time constant U % 3), `s_waitcnt vmcnt(2)` instead of draining the queue
//   * zero padding = out-of-range buffer offsets (the DMA writes zeros); XOR swizzle keyed on the halo column (hx & 7), which
//     makes it independent of the vertical tap: every fragment address is a precomputed lane register + an immediate
//   * BN partial sums over the 16 pixel lanes with DPP row rotations (no LDS)
#include "common.h"

#include <type_traits>

struct Halo4Params {
    const char* src; const char* wgt; char* dst; const char* addend; float* stat;
    const char* bst_x; const unsigned char* bst_mask;       // BST: input and ReLU bitmask of the BatchNorm whose backward consumes dst
    const unsigned char* addend_mask;                        // addend_mode 1, bf16: the addend counts only where its ReLU bit is set (fb_conv_args.addend_mask)
    const float* amax_src; const float* amax_wgt;           // f32h (fp16x2 split): largest magnitudes per chunk of src / per weight set
    int amax_imgs;
    int n_img, H, Cs, Cd, mode;
    int imgs_per_wset; long long wset_stride_bytes;
    int addend_mode, n_mblocks, n_ct, n_tiles;
    unsigned magic_ct, magic_wset;                          // ceil(2^32 / d), 0 for d == 1
    int phase_mode, phase_sleeps;                           // FB_H4_PHASE (A/B switch): one of the two workgroups of a CU starts late by ~3.9 us x sleeps
#ifdef FB_H4_TRACE
    long long* trace;                                       // tools/h4_trace.hip: 8 timestamps per tile
#endif
};
#ifdef FB_H4_TRACE
#define H4_STAMP(k) do { if (tid == 0) p.trace[(long long)L * 8 + (k)] = wall_clock64(); } while (0)
#else
#define H4_STAMP(k) do { } while (0)
#endif

namespace {
template <int N> __device__ __forceinline__ void h4_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void h4_wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
typedef __attribute__((ext_vector_type(4))) unsigned h4_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned h4_u32x2;
template <int OFF> __device__ __forceinline__ uint4 h4_read16(unsigned byte_addr) {
    h4_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void h4_write16(unsigned byte_addr, const uint4& v) {
    const h4_u32x4 d = {v.x, v.y, v.z, v.w};
    asm volatile("ds_write_b128 %0, %1" ::"v"(byte_addr), "v"(d) : "memory");
}
template <int I, int N, typename F> __device__ __forceinline__ void h4_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); h4_static_for<I + 1, N>(f); }
}
__device__ __forceinline__ int h4_xcd_remap(int b, int n) {          // consecutive results live on the same XCD
    const int q = n >> 3, r = n & 7, xcd = b & 7, slot = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
// halo rows between fragment 0 and fragment j of a wave
template <int W> constexpr int h4_frag_rows(int j) { return W == 32 ? (j >> 1) * 40 + (j & 1) * 16 : (W == 16 ? j * 18 : (W == 8 ? j * 2 * 10 : j * 36)); }
__device__ __forceinline__ int h4_div(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }
constexpr unsigned H4_OOB = 0x80000000u;

// CP (4x4 maps only): the COMPACT layout -- no halo cells.  A padded 4x4 image is 36 cells for 16 pixels (72 KiB per 128-byte slice of a
// 16-image tile: one workgroup per CU, 2.25 x the bytes through the LDS-DMA path); compact, an image is its 16 pixels + ONE zero row
// (17 rows, 34 KiB per slice: two workgroups per CU with 64-channel tiles).  The price is nine per-lane fragment addresses instead of
// three (a lane whose tap falls outside the image points at the zero row), i.e. 12 more registers.
template <int W, bool CP = false> struct H4Geo {
    static constexpr int TH = W >= 16 ? 256 / W : W;              // image rows per image part of the tile
    static constexpr int IMGS = W >= 16 ? 1 : 256 / (W * W);      // whole images per tile (W = 8: 4, W = 4: 16)
    static constexpr int TILES_PER_IMG = W == 32 ? 4 : 1;
    // 32x32: pitch 40 = five 1 KiB load groups per halo row, so the top/bottom halo rows are whole groups (their validity
    // depends on the tile's position in the image and is decided per load on the scalar unit)
    static constexpr int PITCH = W == 32 ? 40 : W + 2;
    static constexpr int IMG_ROWS = CP ? W * W + 1 : (TH + 2) * PITCH;   // halo rows per image part
    static constexpr int ROWS = IMGS * IMG_ROWS;                  // 400, 324, 400, 576 (compact 4x4: 272)
    static constexpr int NGRP = (ROWS + 7) / 8;                   // 1 KiB groups of 8 rows
};

struct H4Tile { int pt, ct, n0, y0; };                             // wave-uniform description of one tile
}  // namespace

// FI = 16-channel output fragments per wave: 4 (64-channel tiles, two workgroups per CU) or 8 (128-channel tiles, ONE workgroup per CU with
// 128 accumulator registers: 24 fragment reads per 64 MFMAs instead of 16 per 32, and half the weight / halo bytes from L2 per MFMA --
// the 512-channel 4x4 layers are limited by exactly that stream in the implicit GEMM)
// BST (bf16 input gradient): the reduction pass of the BatchNorm backward that consumes dst, fused into the epilogue -- g = the bf16 output
// where the bit of bst_mask is set, x = bst_x at the same position; `stat` receives the 128-pixel-block sums of g and of g*x.
template <typename T, int W, int FI = 4, bool BST = false, bool CP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FI == 8 ? 1 : 2, FI == 8 ? 1 : 2))) void conv3x3s1_halo4_kernel(const Halo4Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(!CP || (W == 4 && std::is_same<T, bf16_tag>::value), "compact layout: bf16 4x4 maps");
    using G = H4Geo<W, CP>;
    constexpr int EB = ET<T>::EB;
    constexpr int PITCH = G::PITCH, NGRP = G::NGRP;
    constexpr int CO_T = 16 * FI, NWL = FI / 2;                   // output channels per tile; weight LDS-DMA instructions per wave and tap
    constexpr int HALO_BYTES = NGRP * 1024, WT_BYTES = CO_T * 128, NW = 3, RED_BYTES = 4 * CO_T * 2 * 4;
#ifndef FB_H4_COUNTED_TOP
#define FB_H4_COUNTED_TOP 0                                 // A/B build switch (tools/build_variant.py top1 --only conv3x3_halo4.hip -DFB_H4_COUNTED_TOP=1): measured without effect
#endif
#ifndef FB_H4_LDS_PAD
#define FB_H4_LDS_PAD 0                                     // tools/h4_trace.hip: pad to force one workgroup per CU
#endif
    __shared__ __attribute__((aligned(16))) char lds[HALO_BYTES + NW * WT_BYTES + RED_BYTES + FB_H4_LDS_PAD];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NB = gridDim.x;
    const int row_b = p.Cs * EB;
    const int n_cc = row_b / 128;
    const int lrow8 = lane >> 3;                           // row within a 1 KiB group

    // ---- per-lane constants (once per workgroup) ---------------------------------------------------------------------------
    const unsigned voffW0 = (unsigned)((wave * 8 + lrow8) * 9 * row_b + (((lane & 7) ^ lrow8) * 16));       // + k * 32 rows, k < NWL
    // halo source offsets relative to the tile's origin pixel (row y0-1, column -1 of the first image); rows outside the
    // image in x (and in y for whole-image tiles) are out of range for good
    constexpr int KH = (NGRP + 3) / 4;
    unsigned voffH[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k) {
        const int row = (wave + 4 * k) * 8 + lrow8;
        const int img_l = row / G::IMG_ROWS, rr = row - img_l * G::IMG_ROWS;
        const int hy = rr / PITCH, hx = rr - hy * PITCH;
        bool ok = row < G::ROWS && hx >= 1 && hx <= W;
        if constexpr (W != 32) ok = ok && hy >= 1 && hy <= G::TH;
        voffH[k] = ok ? (unsigned)(((img_l * p.H + hy) * W + hx) * row_b + (((lane & 7) ^ (hx & 7)) * 16)) : H4_OOB;
        if constexpr (CP)                                      // row rr < 16 of an image part = its pixel rr (swizzle keyed on the pixel), row 16 = zeros
            voffH[k] = row < G::ROWS && rr < W * W ? (unsigned)((img_l * W * W + rr) * row_b + (((lane & 7) ^ (rr & 7)) * 16)) : H4_OOB;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    // fragment j of a wave = pixels 16j..16j+15 of its 64: the same columns (W = 32: +16, same residue mod 8) a fixed number of
    // halo rows further down, so only fragment 0 needs registers -- [horizontal variant dx+1][K half]; j and the vertical tap
    // are instruction immediates
    // (compact 4x4: [tap][K half] -- a fragment is one image, the tap's neighbour pixel or the image's zero row; fragment j = + j * 17 rows)
    unsigned pa[CP ? 9 : 3][2];
    {
        const int q = wave * 64 + (lane & 15);
        const int img_l = q / (G::TH * W), qi = q - img_l * (G::TH * W);
        const int ty = qi / W, tx = qi % W;
        if constexpr (CP) {
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                const int ny = ty + u / 3 - 1, nx = tx + u % 3 - 1;
                const bool in = ny >= 0 && ny < W && nx >= 0 && nx < W;
                const int pr = in ? ny * W + nx : W * W;
#pragma unroll
                for (int h = 0; h < 2; ++h) pa[u][h] = lds0 + (img_l * G::IMG_ROWS + pr) * 128 + ((((lane >> 4) + 4 * h) ^ (pr & 7)) * 16);
            }
        } else {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int hrow = img_l * G::IMG_ROWS + ty * PITCH + tx + b;
#pragma unroll
                for (int h = 0; h < 2; ++h) pa[b][h] = lds0 + hrow * 128 + ((((lane >> 4) + 4 * h) ^ ((tx + b) & 7)) * 16);
            }
        }
    }
    // output offset of this lane inside a tile: pixel 64*wave + (lane & 15), channels 4*(lane >> 4) .. +3 of a 16-channel fragment
    const int g4 = lane >> 4;
    const int voffD = ((wave * 64 + (lane & 15)) * p.Cd + (lane >> 4) * 4) * EB;
    // bf16 outputs after the row swap of a fragment pair: lane group g owns channels {0, 16, 8, 24}[g] .. +7 of the pair's 32
    const int voffT = ((wave * 64 + (lane & 15)) * p.Cd + (g4 & 1) * 16 + (g4 >> 1) * 8) * EB;
    // BN partial sums: writer lanes (lane & 15 == 0) and the 128 reader threads
    const unsigned red0 = lds0 + HALO_BYTES + NW * WT_BYTES;
    const unsigned redw = red0 + (wave * CO_T + (lane >> 4) * 4) * 8;
    const unsigned redr = red0 + (((tid / CO_T) * 2) * CO_T + (tid % CO_T)) * 8;
    unsigned wa[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) wa[h] = lds0 + HALO_BYTES + (lane & 15) * 128 + ((((lane >> 4) + 4 * h) ^ (lane & 7)) * 16);

    // f32h: the halo slice is converted IN PLACE, once, from fp32 to fp16x2 planes (common.h) before its nine taps -- not per fragment and
    // tap (32 VALU x 4 fragments x 9 taps per wave and slice before).  A 128-byte row holds the 16-byte chunks c = 0..7 of 32 channels at
    // position c ^ (hx & 7); the lane group g of an MFMA step reads chunk g (channels 4g..4g+3) and chunk g + 4 (16+4g..), so the pair
    // of positions (p, p + 4) is converted together: the high pieces of its eight values go where chunk g was, the low pieces where
    // chunk g + 4 was, and the two fragment reads of a lane become its high and low operands.
    constexpr int KI = is_hsplit<T>::value ? (NGRP * 8 * 4 + 255) / 256 : 1;
    unsigned cvA[KI];
    if constexpr (is_hsplit<T>::value) {
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int item = tid + 256 * k, row = item >> 2, pp = item & 3;
            const int rr = row % G::IMG_ROWS, hx = rr % PITCH;
            cvA[k] = row < G::ROWS ? (lds0 + row * 128 + pp * 16) | (((hx & 7) >> 2) & 1u) : 0xffffffffu;
        }
    }
    float hs_src = 1.f, hs_inv = 1.f;                       // f32h: power-of-two scales of the tile's chunk and weight set
    auto convert_halo = [&]() {
        if constexpr (is_hsplit<T>::value) {
#pragma unroll
            for (int k = 0; k < KI; ++k) {
                if (cvA[k] != 0xffffffffu) {
                    const unsigned a = cvA[k] & ~1u, swap = cvA[k] & 1u;
                    const uint4 A = h4_read16<0>(a), B = h4_read16<64>(a);
                    h4_wait_lgkmcnt<0>();
                    const split2h_t sp = swap ? split_h2x8(B, A, hs_src) : split_h2x8(A, B, hs_src);
                    const uint4 hv = __builtin_bit_cast(uint4, sp.h), lv = __builtin_bit_cast(uint4, sp.l);
                    h4_write16(swap ? a + 64 : a, hv);
                    h4_write16(swap ? a : a + 64, lv);
                }
            }
            h4_wait_lgkmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
    };

#ifdef FB_H4_BNFOLD_TIMING
    // TIMING-ONLY experiment (round-4 verdict item 4; tools/build_variant.py bnfold -DFB_H4_BNFOLD_TIMING; the OUTPUT IS NOT the convolution of
    // the normalised input: scale / shift are whatever p.stat holds): what BatchNorm + ReLU of the producing layer costs when it is applied to the
    // staged halo slice in LDS, once per nine taps -- thread (row = tid / 8 + 32 k, octet = tid % 8) rewrites the 16-byte piece of its channel
    // octet in every row that lies inside the image (zero padding must stay zero), 8 scale + 8 shift values per slice from global memory
    constexpr int KF = (G::ROWS + 31) / 32;
    unsigned bfA[KF];
    if constexpr (std::is_same<T, bf16_tag>::value) {
#pragma unroll
        for (int k = 0; k < KF; ++k) {
            const int row = (tid >> 3) + 32 * k, o = tid & 7;
            const int rr = row % G::IMG_ROWS, hy = rr / PITCH, hx = rr % PITCH;
            const bool ok = row < G::ROWS && hx >= 1 && hx <= W && hy >= 1 && hy <= G::TH;      // (W = 32: top / bottom rows of edge tiles ignored here)
            bfA[k] = ok ? lds0 + row * 128 + ((o ^ (hx & 7)) * 16) : 0xffffffffu;
        }
    }
    auto bn_fold = [&](int cc, int n0) {
        if constexpr (std::is_same<T, bf16_tag>::value) {
            const float* sp = (const float*)p.stat + ((n0 >> 7) & 1) * 2 * p.Cs + cc * 64 + (tid & 7) * 8;   // (per-chunk table in the real thing)
            float sc[8], sh[8];
            const float4 a0 = *(const float4*)sp, a1 = *(const float4*)(sp + 4), b0 = *(const float4*)(sp + p.Cs), b1 = *(const float4*)(sp + p.Cs + 4);
            sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
            sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
#pragma unroll
            for (int i = 0; i < 8; ++i) { sc[i] = 1.f + 1e-30f * fminf(fabsf(sc[i]), 1.f); sh[i] = 1e-30f * fminf(fabsf(sh[i]), 1.f); }   // (loaded values kept live; ~ReLU of the input)
            h4_static_for<0, KF>([&](auto kc) {
                constexpr int K = decltype(kc)::value;
                if (bfA[K] != 0xffffffffu) {
                    const uint4 v = h4_read16<0>(bfA[K]);
                    h4_wait_lgkmcnt<0>();
                    const unsigned w[4] = {v.x, v.y, v.z, v.w};
                    unsigned o[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = fmaxf(fmaf(__uint_as_float(w[q] << 16), sc[2 * q], sh[2 * q]), 0.f);
                        const float hi = fmaxf(fmaf(__uint_as_float(w[q] & 0xffff0000u), sc[2 * q + 1], sh[2 * q + 1]), 0.f);
                        o[q] = pack_bf16x2(lo, hi);
                    }
                    h4_write16(bfA[K], make_uint4(o[0], o[1], o[2], o[3]));
                }
            });
            h4_wait_lgkmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
    };
#define H4_BN_FOLD(cc, n0) do { if (p.mode == 0 && p.stat != nullptr) bn_fold(cc, n0); } while (0)
#else
#define H4_BN_FOLD(cc, n0) do { } while (0)
#endif

    // ---- tile bookkeeping (scalar; descriptors are rebuilt where they are needed to keep SGPR pressure low) ----------------
    auto decode = [&](int L) {
        H4Tile t;
        t.pt = h4_div(L, p.magic_ct);
        t.ct = L - t.pt * p.n_ct;
        if constexpr (W == 32) { t.n0 = t.pt >> 2; t.y0 = (t.pt & 3) * G::TH; }
        else { t.n0 = t.pt * G::IMGS; t.y0 = 0; }
        return t;
    };
    auto rsrcA_of = [&](const H4Tile& t) {
        // origin pixel (y0-1, -1): may lie before the tensor for the first tile -- only in-image offsets are ever dereferenced
        if constexpr (CP) return __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + (long long)t.n0 * W * W * row_b), 0, G::IMGS * W * W * row_b, 0x00020000);
        const long long origin = ((long long)(t.n0 * p.H + t.y0 - 1) * W - 1) * row_b;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + origin), 0, (G::IMGS * p.H * W + 2 * W + 2) * row_b, 0x00020000);
    };
    auto rsrcW_of = [&](const H4Tile& t) {
        const int wset = h4_div(t.n0, p.magic_wset);
        return __builtin_amdgcn_make_buffer_rsrc((void*)(p.wgt + (long long)wset * p.wset_stride_bytes), 0, p.Cd * 9 * row_b, 0x00020000);
    };
    auto halo_issue = [&](const H4Tile& t, int cc) {
        const __amdgpu_buffer_rsrc_t rA = rsrcA_of(t);
        const int soff = cc * 128;
        const bool top = t.y0 == 0, bot = t.y0 + G::TH == p.H;
        h4_static_for<0, KH>([&](auto kc) {
            constexpr int K = decltype(kc)::value;
            const int g = wave + 4 * K;
            if (g < NGRP) {
                __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(lds + g * 1024);
                // 32x32: groups 0..4 = halo row 0 (above the image for the top tile), groups 45..49 = halo row TH+1
                constexpr bool edge = W == 32 && (4 * K < 5 || 4 * K + 3 >= 5 * (G::TH + 1));
                bool dead = false;
                if constexpr (edge) dead = (g < 5 && top) || (g >= 5 * (G::TH + 1) && bot);
                if (dead) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, dst, 16, H4_OOB, soff, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, dst, 16, voffH[K], soff, 0, 0);
            }
        });
    };
    auto wt_issue = [&](int buf, const H4Tile& t, int cc, int tap) {
        const __amdgpu_buffer_rsrc_t rW = rsrcW_of(t);
        const int soff = t.ct * CO_T * 9 * row_b + tap * row_b + cc * 128;
        char* dst = lds + HALO_BYTES + buf * WT_BYTES + wave * 1024;
        h4_static_for<0, NWL>([&](auto kc) {
            constexpr int K = decltype(kc)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)(dst + K * 4096), 16,
                                                     voffW0 + (unsigned)(K * 32 * 9 * row_b), soff, 0, 0);
        });
    };
    // forward walks the taps 0..8, the input gradient walks the flipped filter 8..0
    auto tap_of = [&](int u) { return p.mode == 0 ? u : 8 - u; };

    // Static round-robin tile assignment: workgroup v (XCD-grouped id) computes tiles v, v + NB, v + 2 NB, ...  Co-tiles of the
    // same pixels and neighbouring pixel tiles are adjacent in the tile list, so at any time an XCD's L2 serves one contiguous
    // window of the input.  (A dynamic scheduler -- per-XCD atomic counters, index drawn two tiles ahead and handed over through
    // LDS -- was measured: it removes the ~12 % tail imbalance but costs the same in the tap loop; the kernel's throughput is
    // set by the per-CU load/store pipeline, not by idle workgroups.)
    int L = h4_xcd_remap(blockIdx.x, NB);
    if (L >= p.n_tiles) return;
    if (p.phase_mode != 0) {
        // (verdict r4 item 8) The two workgroups of a CU start together and walk tiles of equal length: are they in their epilogues together, with the
        // matrix pipe idle?  Delay one of them by about half a tile.  Which two share a CU is not documented: mode 1 = the second half of an XCD's
        // workgroups (round-robin placement), mode 2 = every other one (packed placement).
        const int k = blockIdx.x >> 3;
        const bool second = p.phase_mode == 1 ? k >= (NB >> 4) : (k & 1) != 0;
        if (second)
            for (int i = 0; i < p.phase_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    auto scales_of = [&](const H4Tile& t) {
        if constexpr (is_hsplit<T>::value) {
            hs_src = fb_pow2_scale(p.amax_src[t.n0 / p.amax_imgs]);
            hs_inv = 1.f / (hs_src * fb_pow2_scale(p.amax_wgt[h4_div(t.n0, p.magic_wset)]));
        }
    };
    H4Tile cur = decode(L);
    bool first_tile = true;
    halo_issue(cur, 0);
    wt_issue(0, cur, 0, tap_of(0));
    wt_issue(1, cur, 0, tap_of(1));

    while (true) {
        H4_STAMP(0); H4_STAMP(1);
#ifdef FB_H4_TRACE
        if (tid == 0) p.trace[(long long)L * 8 + 6] = clock64();
#endif
        const int Ln = L + NB;
        const bool has_next = Ln < p.n_tiles;
        H4Tile nxt = cur;
        if (has_next) nxt = decode(Ln);
        scales_of(cur);
        // halo slice 0 + weight taps 0, 1 have landed (and, with `vmcnt(0)`, the previous tile's stores have been acknowledged).  The loads were requested
        // BEFORE the previous tile's epilogue, so a counted wait would let its stores fly on (FB_H4_COUNTED_TOP=1: bf16 one 16-byte store per fragment pair
        // and pixel fragment, fp32 one per fragment, + 2 statistics stores in the waves of the 2 CO_T reader threads; all issued unconditionally).  Built and
        // measured in round 5, same box: 871 / 864 us (128 -> 128 @16x16 forward), 229.9 / 229.5 against 230.0 / 228.9 ms per step: the other workgroup of
        // the CU already covers that wait.  Off.
        if (first_tile || FB_H4_COUNTED_TOP == 0) h4_wait_vmcnt<0>();
        else if (p.stat != nullptr && wave < 2 * CO_T / 64) h4_wait_vmcnt<(EB == 2 ? 2 * FI : 4 * FI) + 2>();
        else h4_wait_vmcnt<(EB == 2 ? 2 * FI : 4 * FI)>();
        first_tile = false;
        __builtin_amdgcn_s_barrier();
        convert_halo();
        H4_BN_FOLD(0, cur.n0);
        H4_STAMP(2);

        f32x4_t acc[FI][4];
#ifdef FB_H4_MFMA32
        typedef __attribute__((ext_vector_type(16))) float h4_f32x16;
        h4_f32x16 acc32[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc32[i][e] = 0.f;
#endif
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

#ifndef FB_H4_SPLIT_PIPE
#define FB_H4_SPLIT_PIPE 2                                  // 0: round 2-4 order; 1: next weight fragment's split under the MFMAs; 2: + next tap's pixel fragments
#endif
        constexpr bool XPIPE = is_split<T>::value && FB_H4_SPLIT_PIPE == 2;
        split3_t spA[XPIPE ? 4 : 1], spB[XPIPE ? 4 : 1];    // XPIPE: the pixel pieces of the current / the next tap (roles alternate with the tap's parity)
#ifndef FB_H4_PF_PIPE
#define FB_H4_PF_PIPE 0                                     // bf16: the next tap's pixel fragments are read before the barrier that ends this tap (A/B build switch; measured without effect)
#endif
        // PFP (bf16, 64-channel tiles): the halo slice is resident for all nine taps, so the pixel fragments of tap U + 1 are requested at the END of tap U -- their LDS
        // latency runs under the end-of-tap wait and barrier instead of in front of the next tap's first MFMA (the weight fragments cannot: the next
        // tap's weights are only guaranteed behind that barrier).  32 more registers (two sets, alternating with the tap's parity).  Built and measured in
        // round 5 (same bits): 12 544 images, us, with / without: 1018 / 1009, 1012 / 1015 (16 x 16 forward), 768 / 759, 761 / 768 (8 x 8), 705 / 709, 707 / 701 (4 x 4);
        // headline step 228.9 / 229.4, 228.7 / 228.1 ms: nothing -- the LDS latency at a tap's start is already covered by the CU's other workgroup.  Off.
        constexpr bool PFP = std::is_same<T, bf16_tag>::value && FI == 4 && FB_H4_PF_PIPE != 0;
        uint4 pfc0[PFP ? 2 : 1][4], pfc1[PFP ? 2 : 1][4];
        for (int cc = 0; cc < n_cc; ++cc) {
            const bool more = cc + 1 < n_cc;
            h4_static_for<0, 9>([&](auto uc) {
                constexpr int U = decltype(uc)::value, A = U / 3, B = U % 3;
                constexpr int BUF = U % 3, NBUF = (U + 2) % 3;
                // weights two taps ahead go into the buffer tap U-1 used (every wave has passed the barrier that ended U-1)
                bool issued = true;
                if constexpr (U < 7) wt_issue(NBUF, cur, cc, tap_of(U + 2));
                else if (more) wt_issue(NBUF, cur, cc + 1, tap_of(U - 7));
                else if (has_next) wt_issue(NBUF, nxt, 0, tap_of(U - 7));
                else issued = false;
                if constexpr (XPIPE) {
                    // fp32 operands as three bf16 pieces (bf16x6), everything that can be split ahead IS split under MFMAs (round 5): while the 24 MFMAs of weight
                    // fragment I run, the lane splits weight fragment I + 1 of this tap and pixel fragment I of the NEXT tap (the halo slice is resident for all
                    // nine taps; the next tap's weights are not visible before the barrier that ends this one).  What stays exposed per tap: the first weight
                    // fragment's read + split (and, at tap 0 of a slice, the four pixel fragments).  Reads are issued one block before their split, so the
                    // `lgkmcnt(0)` in front of a block's VALU finds them landed.
                    split3_t (&spc)[4] = (U & 1) ? spB : spA;
                    split3_t (&spn)[4] = (U & 1) ? spA : spB;
                    constexpr int UN = U + 1, AN = UN / 3, BN = UN % 3, PAN = CP ? UN : BN, PAC = CP ? U : B;
                    uint4 wl[FI], wh[FI], pl[4], ph[4];
                    if constexpr (U == 0) {
                        h4_static_for<0, 4>([&](auto j) {
                            constexpr int J = decltype(j)::value;
                            pl[J] = h4_read16<(CP ? J * G::IMG_ROWS : A * PITCH + h4_frag_rows<W>(J)) * 128>(pa[PAC][0]);
                            ph[J] = h4_read16<(CP ? J * G::IMG_ROWS : A * PITCH + h4_frag_rows<W>(J)) * 128>(pa[PAC][1]);
                        });
                        h4_wait_lgkmcnt<0>();
#pragma unroll
                        for (int j = 0; j < 4; ++j) spc[j] = split_f32x8(pl[j], ph[j]);
                    }
                    auto rd_w = [&](auto ic) {
                        constexpr int I = decltype(ic)::value;
                        wl[I] = h4_read16<I * 2048 + BUF * WT_BYTES>(wa[0]); wh[I] = h4_read16<I * 2048 + BUF * WT_BYTES>(wa[1]);
                    };
                    auto rd_pn = [&](auto jc) {
                        constexpr int J = decltype(jc)::value;
                        pl[J] = h4_read16<(CP ? J * G::IMG_ROWS : AN * PITCH + h4_frag_rows<W>(J)) * 128>(pa[PAN % (CP ? 9 : 3)][0]);
                        ph[J] = h4_read16<(CP ? J * G::IMG_ROWS : AN * PITCH + h4_frag_rows<W>(J)) * 128>(pa[PAN % (CP ? 9 : 3)][1]);
                    };
                    rd_w(std::integral_constant<int, 0>{});
                    rd_w(std::integral_constant<int, 1>{});
                    if constexpr (U < 8) rd_pn(std::integral_constant<int, 0>{});
                    h4_wait_lgkmcnt<0>();
                    split3_t sw = split_f32x8(wl[0], wh[0]);
                    h4_static_for<0, FI>([&](auto ic) {
                        constexpr int I = decltype(ic)::value;
                        if constexpr (I + 2 < FI) rd_w(std::integral_constant<int, I + 2>{});            // split one block later
                        if constexpr (U < 8 && I + 1 < 4) rd_pn(std::integral_constant<int, I + 1>{});
                        split3_t swn = sw;
                        if constexpr (I + 1 < FI) swn = split_f32x8(wl[I + 1], wh[I + 1]);
                        if constexpr (U < 8) spn[I] = split_f32x8(pl[I], ph[I]);
                        if constexpr (I + 1 < FI || U < 8) mma_split6_row_mix<4, (I + 1 < FI && U < 8) ? 4 : 2>(sw, spc, acc[I]);
                        else mma_split6_row<4>(sw, spc, acc[I]);
                        __builtin_amdgcn_sched_barrier(0);
                        h4_wait_lgkmcnt<0>();
                        sw = swn;
                    });
                } else {
                uint4 wf0[FI], pf0[4], wf1[FI], pf1[4];
                h4_static_for<0, FI>([&](auto i) { wf0[decltype(i)::value] = h4_read16<decltype(i)::value * 2048 + BUF * WT_BYTES>(wa[0]); });
                constexpr int PA = CP ? U : B;                 // which fragment address; the fragment / vertical-tap offset is an immediate
                constexpr bool HAVE_PF = PFP && U > 0;          // this tap's pixel fragments were requested at the end of the previous tap
                if constexpr (!HAVE_PF)
                    h4_static_for<0, 4>([&](auto j) { pf0[decltype(j)::value] = h4_read16<(CP ? decltype(j)::value * G::IMG_ROWS : A * PITCH + h4_frag_rows<W>(decltype(j)::value)) * 128>(pa[PA][0]); });
                h4_static_for<0, FI>([&](auto i) { wf1[decltype(i)::value] = h4_read16<decltype(i)::value * 2048 + BUF * WT_BYTES>(wa[1]); });
                if constexpr (!HAVE_PF)
                    h4_static_for<0, 4>([&](auto j) { pf1[decltype(j)::value] = h4_read16<(CP ? decltype(j)::value * G::IMG_ROWS : A * PITCH + h4_frag_rows<W>(decltype(j)::value)) * 128>(pa[PA][1]); });
                if constexpr (HAVE_PF) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { pf0[j] = pfc0[PFP ? (U & 1) : 0][j]; pf1[j] = pfc1[PFP ? (U & 1) : 0][j]; }
                }
                if constexpr (is_hsplit<T>::value) {     // fp32 operands as two scaled fp16 pieces each, three MFMAs per fragment pair (common.h)
                    h4_wait_lgkmcnt<0>();
                    split2h_t sp[4];                       // both reads of a fragment are operands already: the halo was converted in place
#pragma unroll
                    for (int j = 0; j < 4; ++j) { sp[j].h = __builtin_bit_cast(f16x8_t, pf0[j]); sp[j].l = __builtin_bit_cast(f16x8_t, pf1[j]); }
#pragma unroll
                    for (int i = 0; i < FI; ++i) {
                        split2h_t sw;                          // the weights arrive as fp16x2 planes (fb_weight_prep)
                        sw.h = __builtin_bit_cast(f16x8_t, wf0[i]); sw.l = __builtin_bit_cast(f16x8_t, wf1[i]);
                        mma_split3h_row<4>(sw, sp, acc[i], hs_inv);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if constexpr (is_split<T>::value) {      // fp32 operands as three bf16 pieces each, six MFMAs per fragment pair (common.h)
                    h4_wait_lgkmcnt<0>();
                    split3_t sp[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) sp[j] = split_f32x8(pf0[j], pf1[j]);
#if FB_H4_SPLIT_PIPE
                    // One weight fragment at a time, four zero-started chains in flight -- and the split of fragment i + 1 (44 VALU) threaded through the 24
                    // MFMAs of fragment i: with the split in front of its MFMAs (round 2-4) the two waves of a SIMD, which leave the per-tap barrier together,
                    // ran their split phases against each other and then their MFMA phases (MFMA time + VALU time, not the larger of the two)
                    split3_t sw = split_f32x8(wf0[0], wf1[0]);
                    h4_static_for<0, FI>([&](auto ic) {
                        constexpr int I = decltype(ic)::value;
                        split3_t swn = sw;
                        if constexpr (I + 1 < FI) {
                            swn = split_f32x8(wf0[I + 1], wf1[I + 1]);
                            mma_split6_row_mix<4, 2>(sw, sp, acc[I]);
                        } else {
                            mma_split6_row<4>(sw, sp, acc[I]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        sw = swn;
                    });
#else
#pragma unroll
                    for (int i = 0; i < FI; ++i) {                 // one weight fragment split at a time: four zero-started chains in flight
                        const split3_t sw = split_f32x8(wf0[i], wf1[i]);
                        mma_split6_row<4>(sw, sp, acc[i]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#endif
                } else {
#ifdef FB_H4_MFMA32
                // TIMING-ONLY experiment (tools/h4_trace.hip -DFB_H4_MFMA32; results are wrong): the same fragment reads and the same FLOP per
                // tap issued as v_mfma_f32_32x32x16_bf16 (half as many instructions of twice the work) instead of 16x16x32
                if constexpr (std::is_same<T, bf16_tag>::value && FI == 4) {
                    h4_wait_lgkmcnt<FI + 4>();
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; j += 2)
                            acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf0[i]), __builtin_bit_cast(bf16x8_t, pf0[j] ), acc32[i], 0, 0, 0);
                    h4_wait_lgkmcnt<0>();
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 1; j < 4; j += 2)
                            acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf1[i]), __builtin_bit_cast(bf16x8_t, pf1[j]), acc32[i], 0, 0, 0);
                    // (the unused fragment reads stay: they are volatile asm)
                } else
#endif
                {
                // (LDS returns in order) first half landed: with the pixel fragments requested a tap ago only the second half's FI weight reads may be outstanding
                if constexpr (HAVE_PF) h4_wait_lgkmcnt<FI>(); else h4_wait_lgkmcnt<FI + 4>();
#pragma unroll
                for (int i = 0; i < FI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mma_chunk<T>(wf0[i], pf0[j], acc[i][j]);
                h4_wait_lgkmcnt<0>();
#pragma unroll
                for (int i = 0; i < FI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mma_chunk<T>(wf1[i], pf1[j], acc[i][j]);
                if constexpr (PFP && U < 8) {                  // the next tap's pixel fragments, into the other register set
                    constexpr int UN = U + 1, AN = UN / 3, PAN = CP ? UN : UN % 3;
                    h4_static_for<0, 4>([&](auto j) {
                        constexpr int J = decltype(j)::value;
                        pfc0[PFP ? (UN & 1) : 0][J] = h4_read16<(CP ? J * G::IMG_ROWS : AN * PITCH + h4_frag_rows<W>(J)) * 128>(pa[PAN][0]);
                        pfc1[PFP ? (UN & 1) : 0][J] = h4_read16<(CP ? J * G::IMG_ROWS : AN * PITCH + h4_frag_rows<W>(J)) * 128>(pa[PAN][1]);
                    });
                }
                }
                }
                }   // (!XPIPE)
                if constexpr (U == 8) {
                    __builtin_amdgcn_s_barrier();         // every wave is done with this halo slice
                    if (more) {
                        halo_issue(cur, cc + 1);
                        h4_wait_vmcnt<0>();
                        __builtin_amdgcn_s_barrier();
                        convert_halo();
                        H4_BN_FOLD(cc + 1, cur.n0);
                    } else if (has_next) {
                        halo_issue(nxt, 0);               // lands during the epilogue; the loop top waits for it
                    }
                } else {
                    if (issued) h4_wait_vmcnt<NWL>();     // the weights of the next tap have landed; the NWL loads of the tap after it stay in flight
                    else h4_wait_vmcnt<0>();
                    __builtin_amdgcn_s_barrier();
                }
            });
        }
        H4_STAMP(3);

        // ---- epilogue ------------------------------------------------------------------------------------------------------
        // outputs (and the same-shape addend) are addressed as  tile base (descriptor) + lane offset (one register, tile
        // invariant) + scalar/immediate offsets of the fragment: no 64-bit address arithmetic on the vector unit
        const __amdgpu_buffer_rsrc_t rsrcD = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dst + (long long)cur.pt * 256 * p.Cd * EB), 0,
                                                                                256 * p.Cd * EB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrcE = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.addend_mode == 1 ? p.addend + (long long)cur.pt * 256 * p.Cd * EB : p.dst), 0, p.addend_mode == 1 ? 256 * p.Cd * EB : 0, 0x00020000);
        float ssum[FI][4], ssq[FI][4];
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
        // BST: the mask bytes of the whole tile are requested up front, x one pixel fragment ahead (the other workgroup of the CU covers
        // the round trips; FI = 8 has no registers for more)
        h4_u32x2 bx[BST ? 2 : 1][BST ? FI : 1];
        unsigned bm[BST ? 4 : 1][BST ? FI / 2 : 1];              // CO_T / 8 mask bytes per pixel
        const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc((void*)(BST ? p.bst_x + (long long)cur.pt * 256 * p.Cd * EB : p.dst), 0,
                                                                                BST ? 256 * p.Cd * EB : 0, 0x00020000);
        auto bx_issue = [&](auto jc) {
            constexpr int J = decltype(jc)::value;
            if constexpr (BST && J < 4) {
                const int soff = (J * 16 * p.Cd + cur.ct * CO_T) * EB;
                h4_static_for<0, FI>([&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    bx[J & 1][I] = __builtin_amdgcn_raw_buffer_load_b64(rsrcX, voffD + I * 16 * EB, soff, 0);
                });
            }
        };
        // masked addend (the residual-branch gradient d * (out > 0) of an identity block without a materialised copy): CO_T / 8 mask bytes per pixel, the
        // whole tile's requested up front like the BST mask
        // (bf16: a mask byte covers the 8 channels of a 16-byte vector -- CO_T / 8 bytes per pixel; fp32: the 4 channels of one -- CO_T / 4 bytes, bits 0-3)
        constexpr int AMW = EB == 4 ? FI : FI / 2;
        unsigned am[4][AMW];
        const bool masked = p.addend_mode == 1 && p.addend_mask != nullptr;
        if (masked) {
            constexpr int SH = EB == 4 ? 2 : 3;                  // log2(channels per mask byte)
            const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.addend_mask + (((long long)cur.pt * 256 * p.Cd) >> SH)), 0, (256 * p.Cd) >> SH, 0x00020000);
            const int voffA = (wave * 64 + (lane & 15)) * (p.Cd >> SH);
            h4_static_for<0, 4>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                const int soffA = (J * 16 * p.Cd + cur.ct * CO_T) >> SH;
                if constexpr (AMW == 2) {
                    const h4_u32x2 m = __builtin_amdgcn_raw_buffer_load_b64(rsrcA, voffA, soffA, 0);
                    am[J][0] = m[0]; am[J][1] = m[1];
                } else {
                    static_assert(AMW == 4, "masked addend: 64-channel tiles (bf16: also 128)");
                    const h4_u32x4 m = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, voffA, soffA, 0);
                    am[J][0] = m[0]; am[J][1] = m[1]; am[J][2] = m[2]; am[J][3] = m[3];
                }
            });
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < AMW; ++i) am[j][i] = 0xffffffffu;
        }
        if constexpr (BST) {
            const __amdgpu_buffer_rsrc_t rsrcN = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bst_mask + (long long)cur.pt * 32 * p.Cd), 0, 32 * p.Cd, 0x00020000);
            const int voffN = (wave * 64 + (lane & 15)) * (p.Cd >> 3);
            h4_static_for<0, 4>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                const int soffN = (J * 16 * p.Cd + cur.ct * CO_T) >> 3;
                if constexpr (FI == 4) {
                    const h4_u32x2 m = __builtin_amdgcn_raw_buffer_load_b64(rsrcN, voffN, soffN, 0);
                    bm[J][0] = m[0]; bm[J][1] = m[1];
                } else {
                    const h4_u32x4 m = __builtin_amdgcn_raw_buffer_load_b128(rsrcN, voffN, soffN, 0);
                    bm[J][0] = m[0]; bm[J][1] = m[1]; bm[J][2] = m[2]; bm[J][3] = m[3];
                }
            });
            bx_issue(std::integral_constant<int, 0>{});
        }
        h4_static_for<0, 4>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            const int soff = (J * 16 * p.Cd + cur.ct * CO_T) * EB;
            unsigned pk[FI][2];
            bx_issue(std::integral_constant<int, J + 1>{});
            h4_static_for<0, FI>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
#ifdef FB_H4_MFMA32
                if constexpr (std::is_same<T, bf16_tag>::value && FI == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[I][J][e] = acc32[I][4 * J + e];
                }
#endif
                float v[4] = {acc[I][J][0], acc[I][J][1], acc[I][J][2], acc[I][J][3]};
                if (p.addend_mode == 1) {
                    if constexpr (EB == 4) {
                        h4_u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrcE, voffD + I * 16 * EB, soff, 0);
                        if (masked) {        // (fp32 storage: mask byte 4 I + g4 of the pixel's CO_T / 4 holds this lane's four channels in bits 0-3)
                            const unsigned bits = am[J][I] >> (g4 * 8);
                            a[0] &= (unsigned)__builtin_amdgcn_sbfe((int)bits, 0, 1); a[1] &= (unsigned)__builtin_amdgcn_sbfe((int)bits, 1, 1);
                            a[2] &= (unsigned)__builtin_amdgcn_sbfe((int)bits, 2, 1); a[3] &= (unsigned)__builtin_amdgcn_sbfe((int)bits, 3, 1);
                        }
                        v[0] += __uint_as_float(a[0]); v[1] += __uint_as_float(a[1]); v[2] += __uint_as_float(a[2]); v[3] += __uint_as_float(a[3]);
                    } else {
                        h4_u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rsrcE, voffD + I * 16 * EB, soff, 0);
                        if (masked) {        // mask byte 2 (I & 1) + (g4 >> 1) of the word covers this lane's channels in its low (g4 even) or high nibble
                            const unsigned bits = am[J][I >> 1] >> (((2 * (I & 1) + (g4 >> 1)) * 8) + (g4 & 1) * 4);
                            const unsigned m0 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 0, 1), m1 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 1, 1);
                            const unsigned m2 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 2, 1), m3 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 3, 1);
                            a[0] &= (m0 & 0x0000ffffu) | (m1 & 0xffff0000u);
                            a[1] &= (m2 & 0x0000ffffu) | (m3 & 0xffff0000u);
                        }
                        v[0] += __uint_as_float(a[0] << 16); v[1] += __uint_as_float(a[0] & 0xffff0000u);
                        v[2] += __uint_as_float(a[1] << 16); v[3] += __uint_as_float(a[1] & 0xffff0000u);
                    }
                } else if (p.addend_mode == 2) {               // 2x2-average-pooled addend, broadcast back (0.25 each)
                    const int q = wave * 64 + J * 16 + (lane & 15);
                    const int img_l = q / (G::TH * W), qi = q - img_l * (G::TH * W);
                    const int oy = cur.y0 + qi / W, ox = qi % W;
                    const long long apix = ((long long)(cur.n0 + img_l) * (p.H >> 1) + (oy >> 1)) * (W >> 1) + (ox >> 1);
                    const char* ap = p.addend + (apix * p.Cd + cur.ct * CO_T + I * 16 + (lane >> 4) * 4) * EB;
                    if constexpr (EB == 4) { const float4 a = *(const float4*)ap; v[0] += 0.25f * a.x; v[1] += 0.25f * a.y; v[2] += 0.25f * a.z; v[3] += 0.25f * a.w; }
                    else { const uint2 a = *(const uint2*)ap; v[0] += 0.25f * __uint_as_float(a.x << 16); v[1] += 0.25f * __uint_as_float(a.x & 0xffff0000u);
                           v[2] += 0.25f * __uint_as_float(a.y << 16); v[3] += 0.25f * __uint_as_float(a.y & 0xffff0000u); }
                }
                if constexpr (EB == 4) {
                    const h4_u32x4 outv = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(outv, rsrcD, voffD + I * 16 * EB, soff, 0);
                    store_b128_guard(outv);
                } else {
                    // bf16: a lane holds channels 4g..4g+3 of this 16-channel fragment (8 bytes).  v_permlane16_swap exchanges the
                    // odd 16-lane rows of fragment I with the even rows of fragment I+1, after which every lane owns 8 CONSECUTIVE
                    // channels: one 16-byte store per fragment pair -- half the store instructions (the epilogue is store-issue
                    // bound), without a trip through LDS
                    pk[I][0] = pack_bf16x2(v[0], v[1]); pk[I][1] = pack_bf16x2(v[2], v[3]);
                    if constexpr ((I & 1) == 1) {
                        const h4_u32x2 lo = __builtin_amdgcn_permlane16_swap(pk[I - 1][0], pk[I][0], false, false);   // rows (X0,Y0,X2,Y2) / (X1,Y1,X3,Y3)
                        const h4_u32x2 hi = __builtin_amdgcn_permlane16_swap(pk[I - 1][1], pk[I][1], false, false);
                        const h4_u32x4 outv = {lo[0], hi[0], lo[1], hi[1]};
                        __builtin_amdgcn_raw_buffer_store_b128(outv, rsrcD, voffT + (I >> 1) * 32 * EB, soff, 0);
                        store_b128_guard(outv);
                    }
                }
                if constexpr (BST) {
                    // mask byte 2I + (g4 >> 1) of the pixel's CO_T / 8 covers this lane's channels in its low (g4 even) or high nibble
                    const unsigned bits = bm[J][I >> 1] >> (((2 * (I & 1) + (g4 >> 1)) * 8) + (g4 & 1) * 4);
                    const unsigned m0 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 0, 1), m1 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 1, 1);
                    const unsigned m2 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 2, 1), m3 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 3, 1);
                    const h4_u32x2 xv = bx[J & 1][I];
                    const float g0 = __uint_as_float((pk[I][0] << 16) & m0), g1 = __uint_as_float(pk[I][0] & 0xffff0000u & m1);
                    const float g2 = __uint_as_float((pk[I][1] << 16) & m2), g3 = __uint_as_float(pk[I][1] & 0xffff0000u & m3);
                    ssum[I][0] += g0; ssum[I][1] += g1; ssum[I][2] += g2; ssum[I][3] += g3;
                    ssq[I][0] = fmaf(g0, __uint_as_float(xv[0] << 16), ssq[I][0]); ssq[I][1] = fmaf(g1, __uint_as_float(xv[0] & 0xffff0000u), ssq[I][1]);
                    ssq[I][2] = fmaf(g2, __uint_as_float(xv[1] << 16), ssq[I][2]); ssq[I][3] = fmaf(g3, __uint_as_float(xv[1] & 0xffff0000u), ssq[I][3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ssum[I][r] += v[r]; ssq[I][r] += v[r] * v[r]; }
                }
            });
        });
        H4_STAMP(4);
        if (p.stat != nullptr) {
            // [4 waves][64 co]{sum, sumsq} in its own LDS region (the halo is being refilled); inline-asm LDS ops: the compiler
            // would drain the in-flight LDS-DMA of the next tile before any LDS access it can see
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { ssum[i][r] = row16_sum(ssum[i][r]); ssq[i][r] = row16_sum(ssq[i][r]); }
            if ((lane & 15) == 0) {
                h4_static_for<0, 4 * FI>([&](auto c) {
                    constexpr int I = decltype(c)::value / 4, R = decltype(c)::value % 4;
                    const f32x2_t d = {ssum[I][R], ssq[I][R]};
                    const unsigned wr = redw;             // (a plain use: asm operands alone do not capture in a nested lambda)
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(wr), "v"(d), "n"((I * 16 + R) * 8) : "memory");
                });
            }
            h4_wait_lgkmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (tid < 2 * CO_T) {
                const int half = tid / CO_T, col = tid % CO_T;
                f32x2_t x, y;
                asm volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(redr));
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(y) : "v"(redr), "n"(CO_T * 8));
                h4_wait_lgkmcnt<0>();
                const long long blk = 2LL * cur.pt + half;
                p.stat[blk * p.Cd + cur.ct * CO_T + col] = x[0] + y[0];
                p.stat[((long long)p.n_mblocks + blk) * p.Cd + cur.ct * CO_T + col] = x[1] + y[1];
            }
        }
        H4_STAMP(5);
#ifdef FB_H4_TRACE
        if (tid == 0) p.trace[(long long)L * 8 + 7] = clock64();
#endif
        if (!has_next) break;
        cur = nxt;
        L = Ln;
    }
#endif
}

static unsigned h4_magic(int d) { return d <= 1 ? 0u : (unsigned)(((1ULL << 32) + (unsigned)d - 1) / (unsigned)d); }

// 0: not for this kernel; 1: 64-channel tiles; 2: 128-channel tiles; 3: 64-channel tiles, compact 4x4 layout
static int h4_variant(const fb_conv_args* a) {
    static const bool disabled = getenv("FB_DISABLE_HALO4") != nullptr;
    if (disabled) return 0;
    if (a->R != 3 || a->S != 3 || a->stride != 1 || a->pad != 1) return 0;
    if (a->Hs != a->Hd || a->Ws != a->Wd || a->Hs != a->Ws) return 0;
    const int W = a->Ws;
    if (W != 32 && W != 16 && W != 8 && W != 4) return 0;
    const int EB = a->dtype == FB_F32 ? 4 : 2;
    if (a->Cs * EB % 128 != 0 || a->Cd % 64 != 0) return 0;
    const int imgs_per_wset = a->imgs_per_wset > 0 ? a->imgs_per_wset : a->n_img;
    const int imgs_per_tile = W >= 16 ? 1 : 256 / (W * W);
    if (a->n_img % imgs_per_tile != 0 || imgs_per_wset % imgs_per_tile != 0) return 0;
    // 128-channel tiles (one workgroup per CU, FI = 8): bf16; always for the 4x4 maps (which have no 64-channel variant), elsewhere only
    // with FB_H4_WIDE (A/B switch: a list of map widths, e.g. "8,16")
    static const char* wide_env = fb_getenv_experimental("FB_H4_WIDE");
    char wtag[8];
    snprintf(wtag, sizeof(wtag), "%d", W);
    // 4x4 maps (bf16): the compact layout with 64-channel tiles, two workgroups per CU (FB_H4_COMPACT=0: the padded layout, which only fits
    // 128-channel tiles with one workgroup per CU)
    const char* cp_env = getenv("FB_H4_COMPACT");                  // read per call: the tests compare the two layouts in one process
    const bool compact_off = cp_env != nullptr && atoi(cp_env) == 0;
    const bool compact = W == 4 && a->dtype == FB_BF16 && !compact_off;
    const bool wide = !compact && a->dtype == FB_BF16 && a->Cd % 128 == 0 && (W == 4 || (wide_env && W != 32 && strstr(wide_env, wtag)));
    if (W == 4 && !wide && !compact) return 0;
    if ((long long)(imgs_per_tile * a->Hs * W + 2 * W + 2) * a->Cs * EB >= (1LL << 31)) return 0;
    const int n_pt = a->n_img * a->Hs * W / 256, n_ct = a->Cd / (wide ? 128 : 64);
    if ((long long)n_pt * n_ct * n_ct >= (1LL << 32) || (long long)a->n_img * imgs_per_wset >= (1LL << 32)) return 0;
    // masked addend: input gradients with a same-shape addend, bf16 and fp32 storage (FB_H4_NO_MASK: A/B switch -- the engine then materialises d * (out > 0))
    if (a->addend_mask && (a->mode != 1 || !a->addend || a->addend_mode != 1 || getenv("FB_H4_NO_MASK") != nullptr)) return 0;
    // fused BatchNorm-backward reduction: bf16 input gradients on 16x16 / 8x8 / 4x4 maps
    if (a->bst_x && (a->mode != 1 || !a->bst_mask || !a->stat_partial || a->dtype != FB_BF16 || W == 32)) return 0;
    return compact ? 3 : (wide ? 2 : 1);
}

int fb_conv3x3_halo4_takes(const fb_conv_args* a) { return h4_variant(a) != 0; }

// returns 1 if the kernel handled the call
int fb_try_conv3x3_halo4(const fb_conv_args* a, hipStream_t st) {
    const int variant = h4_variant(a);
    if (variant == 0) return 0;
    const bool wide = variant == 2;
    const int W = a->Ws, EB = a->dtype == FB_F32 ? 4 : 2;
    const int imgs_per_wset = a->imgs_per_wset > 0 ? a->imgs_per_wset : a->n_img;
    const int n_cu = fb_persistent_cus();
    Halo4Params p;
    p.src = (const char*)a->src; p.wgt = (const char*)a->wgt; p.dst = (char*)a->dst; p.addend = (const char*)a->addend;
    p.stat = a->stat_partial;
    p.bst_x = (const char*)a->bst_x; p.bst_mask = (const unsigned char*)a->bst_mask;
    p.addend_mask = (const unsigned char*)a->addend_mask;
    p.amax_src = a->amax_src; p.amax_wgt = a->amax_wgt; p.amax_imgs = a->amax_imgs > 0 ? a->amax_imgs : a->n_img;
    p.n_img = a->n_img; p.H = a->Hs; p.Cs = a->Cs; p.Cd = a->Cd; p.mode = a->mode;
    p.imgs_per_wset = imgs_per_wset;
    p.wset_stride_bytes = a->wset_stride * EB;
    p.addend_mode = a->addend ? a->addend_mode : 0;
    const int n_pt = a->n_img * a->Hs * W / 256;
    p.n_mblocks = n_pt * 2;
    p.n_ct = a->Cd / (wide ? 128 : 64);
    p.n_tiles = n_pt * p.n_ct;
    p.magic_ct = h4_magic(p.n_ct);
    p.magic_wset = h4_magic(imgs_per_wset);
    {
        static const char* ph = fb_getenv_experimental("FB_H4_PHASE");      // "mode,sleeps"
        p.phase_mode = ph ? atoi(ph) : 0;
        p.phase_sleeps = ph && strchr(ph, ',') ? atoi(strchr(ph, ',') + 1) : 2;
    }
#ifdef FB_H4_TRACE
    extern long long* g_h4_trace;
    p.trace = g_h4_trace;
#endif
    // (FB_H4_WG_PER_CU=1: one resident workgroup per CU instead of two -- leaves half of every CU's registers and LDS to a kernel of another
    // stream; A/B switch of the co-scheduling experiments, profiles/r4_notes.md)
    static const int wg_per_cu = fb_getenv_experimental("FB_H4_WG_PER_CU") ? atoi(fb_getenv_experimental("FB_H4_WG_PER_CU")) : 2;
    const int slots = (wg_per_cu == 1 ? 1 : 2) * n_cu;
    dim3 grid(p.n_tiles < slots ? p.n_tiles : slots);
    if (variant == 3) {
        if (a->bst_x) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<bf16_tag, 4, 4, true, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3s1_halo4_kernel<bf16_tag, 4, 4, false, true>), grid, dim3(256), 0, st, p);
        return 1;
    }
    if (a->bst_x) {
        dim3 grid1(p.n_tiles < n_cu ? p.n_tiles : n_cu);
        if (wide && W == 16) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<bf16_tag, 16, 8, true>), grid1, dim3(256), 0, st, p);
        else if (wide && W == 8) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<bf16_tag, 8, 8, true>), grid1, dim3(256), 0, st, p);
        else if (wide) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<bf16_tag, 4, 8, true>), grid1, dim3(256), 0, st, p);
        else if (W == 16) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<bf16_tag, 16, 4, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3s1_halo4_kernel<bf16_tag, 8, 4, true>), grid, dim3(256), 0, st, p);
        return 1;
    }
    if (wide) {
        dim3 grid1(p.n_tiles < n_cu ? p.n_tiles : n_cu);
        if (W == 16) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<bf16_tag, 16, 8>), grid1, dim3(256), 0, st, p);
        else if (W == 8) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<bf16_tag, 8, 8>), grid1, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3s1_halo4_kernel<bf16_tag, 4, 8>), grid1, dim3(256), 0, st, p);
        return 1;
    }
    if (a->dtype == FB_F32 && a->amax_src && a->amax_wgt) {
        if (W == 32) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<f32h_tag, 32>), grid, dim3(256), 0, st, p);
        else if (W == 16) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<f32h_tag, 16>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3s1_halo4_kernel<f32h_tag, 8>), grid, dim3(256), 0, st, p);
    } else if (a->dtype == FB_F32 && fb_f32_split_enabled()) {
        if (W == 32) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<f32s_tag, 32>), grid, dim3(256), 0, st, p);
        else if (W == 16) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<f32s_tag, 16>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3s1_halo4_kernel<f32s_tag, 8>), grid, dim3(256), 0, st, p);
    } else if (a->dtype == FB_F32) {
        if (W == 32) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<float, 32>), grid, dim3(256), 0, st, p);
        else if (W == 16) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<float, 16>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3s1_halo4_kernel<float, 8>), grid, dim3(256), 0, st, p);
    } else {
        if (W == 32) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<bf16_tag, 32>), grid, dim3(256), 0, st, p);
        else if (W == 16) hipLaunchKernelGGL((conv3x3s1_halo4_kernel<bf16_tag, 16>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3s1_halo4_kernel<bf16_tag, 8>), grid, dim3(256), 0, st, p);
    }
    return 1;
}
