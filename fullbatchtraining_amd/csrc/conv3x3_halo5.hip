// 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels on 32x32 maps, bf16 (ResNet layer 1: 8 of the 40 convolution launches of a
// chunk group but the largest single share of the step): one workgroup per CU, one wave per SIMD, ALL NINE weight taps resident.
//
// The persistent halo kernel (conv3x3_halo4.hip) is bound on chip, not by memory (profiles/r1_pmc_notes.md): fragment reads in
// front of every MFMA batch, nine workgroup barriers per tile for the weight ring, an epilogue that only another workgroup can
// overlap.  With 64 input and 64 output channels the whole filter is 73,728 bytes, so here
//   * the filter is loaded once per workgroup and stays in LDS: no weight stream, no barrier inside a tile
//   * the halo (10 rows x 34 pixels x 128 B, pitch 34) is double buffered: the next tile's halo is requested before the taps of
//     the current tile and lands under them
//   * a wave computes 128 pixels x 32 output channels (16 accumulator fragments); the LDS reads of MFMA batch s+1 are issued
//     before batch s runs (two register sets), so the MFMA pipe does not wait for LDS
//   * the BN partial sums of a wave cover a whole 128-pixel statistics block of its 32 channels: no LDS, no barrier
// LDS: 9 x 8 KiB weights + 2 x 43 KiB halo = 161,792 of 163,840 bytes.
#include "common.h"

#include <type_traits>

struct Halo5Params {
    const char* src; const char* wgt; char* dst; const char* addend; float* stat;
    const unsigned char* addend_mask;                        // ADD == 2: ReLU bitmask of the addend (1 byte per 8 channels)
    const char* bst_x; const unsigned char* bst_mask;       // BST: input and ReLU bitmask of the BatchNorm whose backward consumes dst
    int n_img, mode, addend_mode, n_mblocks, n_tiles;
#ifdef FB_H5_TRACE
    long long* trace;                                       // tools/h5_trace.hip: 8 timestamps per tile
#endif
};
#ifdef FB_H5_TRACE
#define H5_STAMP(k) do { if (tid == 0) p.trace[(long long)L * 8 + (k)] = wall_clock64(); } while (0)
#else
#define H5_STAMP(k) do { } while (0)
#endif

namespace {
template <int N> __device__ __forceinline__ void h5_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void h5_wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
typedef __attribute__((ext_vector_type(4))) unsigned h5_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned h5_u32x2;
template <int OFF> __device__ __forceinline__ uint4 h5_read16(unsigned byte_addr) {
    h5_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
template <int I, int N, typename F> __device__ __forceinline__ void h5_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); h5_static_for<I + 1, N>(f); }
}
__device__ __forceinline__ int h5_xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, xcd = b & 7, slot = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
constexpr unsigned H5_OOB = 0x80000000u;
constexpr int H5_W = 32, H5_TH = 8, H5_PITCH = 34, H5_ROWS = (H5_TH + 2) * H5_PITCH;   // 340 halo rows of 128 B
constexpr int H5_NGRP = (H5_ROWS + 7) / 8;                                             // 43 groups of 1 KiB
constexpr int H5_HALO_BYTES = H5_NGRP * 1024, H5_WT_BYTES = 64 * 128, H5_WGT_BYTES = 9 * H5_WT_BYTES;
constexpr int H5_KH = (H5_NGRP + 3) / 4;                                               // halo pieces per wave (11; wave 3: 10)
#ifndef FB_H5_ADD_DEPTH
#define FB_H5_ADD_DEPTH 3                                   // epilogue operands (addend / BST x) are requested this many pixel fragments ahead
#endif
constexpr int H5_AD = FB_H5_ADD_DEPTH, H5_ADN = H5_AD + 1; // ring of H5_ADN register sets
}  // namespace

// MODE 0: forward (BN partial sums), MODE 1: input gradient (flipped taps; ADD 1: + addend of the same shape, ADD 2: + addend where
// its ReLU bitmask is set -- the masked gradient of the residual branch without a materialised copy).  Compile-time so that
// the epilogue slices are straight-line code the scheduler can thread through the MFMA batches.
// BST (input gradient only): the reduction pass of the BatchNorm backward that consumes the output rides along -- with g = the bf16 output
// where the bit of bst_mask is set and x = bst_x at the same position, `stat` receives the 128-pixel-block sums of g and of g*x (the forward
// variant's sums / sums of squares slots).  8 more bytes per lane and fragment pair in the prefetch queue, ~20 VALU per fragment.
template <int MODE, int ADD, int BST = 0>
__global__ __launch_bounds__(256) void conv3x3s1_c64_halo5_kernel(const Halo5Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int W = H5_W, PITCH = H5_PITCH;
    __shared__ __attribute__((aligned(16))) char lds[H5_WGT_BYTES + 2 * H5_HALO_BYTES + 1024];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = wave & 1, ch = wave >> 1;               // pixel half (128 pixels = 4 image rows) and output-channel half (32)
    const int NB = gridDim.x;
    const int lrow8 = lane >> 3, t = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    // ---- the filter, once: LDS slot U holds the tap the loop uses at position U (forward 0..8, input gradient 8..0) ----------
    {
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, 64 * 9 * 128, 0x00020000);
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int tap = MODE == 0 ? u : 8 - u;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int piece = wave + 4 * k;                                         // 8 weight rows (output channels) of 128 B
                const unsigned voff = (unsigned)((piece * 8 + lrow8) * 9 * 128 + (((lane & 7) ^ lrow8) * 16));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)(lds + u * H5_WT_BYTES + piece * 1024), 16, voff,
                                                         tap * 128, 0, 0);
            }
        }
    }
    // ---- per-lane constants ---------------------------------------------------------------------------------------------------
    // halo source offsets relative to the tile's origin pixel (row y0-1, column -1); bit k of topbits / botbits: the lane's row of
    // piece k is halo row 0 / TH+1 (outside the image for the first / last tile of an image)
    unsigned voffH[H5_KH];
    unsigned topbits = 0, botbits = 0;
#pragma unroll
    for (int k = 0; k < H5_KH; ++k) {
        const int row = (wave + 4 * k) * 8 + lrow8;
        const int hy = row / PITCH, hx = row - hy * PITCH;
        const bool ok = row < H5_ROWS && hx >= 1 && hx <= W;
        voffH[k] = ok ? (unsigned)((hy * W + hx) * 128 + (((lane & 7) ^ (hx & 7)) * 16)) : H5_OOB;
        topbits |= (hy == 0 ? 1u : 0u) << k;
        botbits |= (hy == H5_TH + 1 ? 1u : 0u) << k;
    }
    // fragment read addresses: A (weights) rows ch*32 + 16 i + t, B (pixels) halo rows; j, the vertical tap and the K half of the
    // weights are immediates, the horizontal tap and the halo buffer select a register
    unsigned wa[2], wb[2];                                   // taps 0..3 / 4..8 (16-bit immediate range)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        wa[h] = lds0 + (ch * 32 + t) * 128 + (((g + 4 * h) ^ (t & 7)) * 16);
        wb[h] = wa[h] + 4 * H5_WT_BYTES;
    }
    unsigned pa[2][3][2];                                    // [halo buffer][horizontal tap][K half]
#pragma unroll
    for (int bf = 0; bf < 2; ++bf)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                pa[bf][b][h] = lds0 + H5_WGT_BYTES + bf * H5_HALO_BYTES + (ph * 4 * PITCH + t + b) * 128 + (((g + 4 * h) ^ ((t + b) & 7)) * 16);
    // output: pixel ph*128 + 16 j + t, channels ch*32 + 16 i + 4 g .. +3
    const int voffD = ((ph * 128 + t) * 64 + ch * 32 + g * 4) * 2;
    // after the row swap lane group g owns channels {0, 16, 8, 24}[g] .. +7 of the wave's 32
    const int voffS = ((ph * 128 + t) * 64 + ch * 32 + (g & 1) * 16 + (g >> 1) * 8) * 2;

    // one 1 KiB piece (index K of this wave) of the halo of tile L into halo buffer `buf`
    // (branch-free: `live` = false turns the piece into an out-of-range load that writes zeros; the 44th group of wave 3 lands in
    // the spare KiB behind the halo buffers)
    auto halo_piece = [&](auto kc, int buf, int L, bool live) {
        constexpr int K = decltype(kc)::value;
        const int grp = wave + 4 * K;
        const int n0 = L >> 2, y0 = (L & 3) * H5_TH;
        const long long origin = ((long long)(n0 * W + y0 - 1) * W - 1) * 128;       // may precede the tensor: only in-image offsets are read
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + origin), 0, ((H5_TH + 2) * W + 2) * 128, 0x00020000);
        const unsigned dead = (y0 == 0 ? topbits : 0u) | (y0 + H5_TH == W ? botbits : 0u);
        const unsigned v = (((dead >> K) & 1u) || !live) ? H5_OOB : voffH[K];
        const int dst = grp < H5_NGRP ? H5_WGT_BYTES + buf * H5_HALO_BYTES + grp * 1024 : H5_WGT_BYTES + 2 * H5_HALO_BYTES;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(lds + dst), 16, v, 0, 0, 0);
    };

    // ---- epilogue of one tile, cut into 18 slices so that it can ride on the MFMA batches of the NEXT tile -----------------------
    //   slices 0..15: fragment (i = S & 1, j = S >> 1): (+addend) -> bf16 -> dst, BN partial sums;  16: the DPP row sums;  17: stat stores
    float ssum[2][4], ssq[2][4];
    // addend (input gradient of the residual branch): loaded three pixel fragments (six batches) ahead of its use -- a load issued
    // where it is consumed would stall the whole in-order stream for an HBM round trip
    h5_u32x2 ad[H5_ADN][2];
    unsigned adm[H5_ADN];                                   // ADD == 2: the four mask bytes of this lane's pixel and channel half
    h5_u32x2 bx[H5_ADN][2];                                 // BST: x of the consuming BatchNorm, same positions as the outputs
    unsigned bxm[H5_ADN];
    auto ad_issue = [&](auto jc, const int Lp) {
        constexpr int J = decltype(jc)::value;
        if constexpr (BST != 0 && J < 8) {
            const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bst_x + (long long)Lp * 256 * 128), 0, 256 * 128, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i) bx[J % H5_ADN][i] = __builtin_amdgcn_raw_buffer_load_b64(rsrcX, voffD + i * 32, J * 2048, 0);
            const __amdgpu_buffer_rsrc_t rsrcN = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bst_mask + (long long)Lp * 256 * 8), 0, 256 * 8, 0x00020000);
            bxm[J % H5_ADN] = __builtin_amdgcn_raw_buffer_load_b32(rsrcN, ((ph * 128 + t) * 8 + ch * 4), J * 128, 0);
        }
        if constexpr (ADD != 0 && J < 8) {
            const __amdgpu_buffer_rsrc_t rsrcE = __builtin_amdgcn_make_buffer_rsrc((void*)(p.addend + (long long)Lp * 256 * 128), 0, 256 * 128, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i) ad[J % H5_ADN][i] = __builtin_amdgcn_raw_buffer_load_b64(rsrcE, voffD + i * 32, J * 2048, 0);
            if constexpr (ADD == 2) {
                const __amdgpu_buffer_rsrc_t rsrcM = __builtin_amdgcn_make_buffer_rsrc((void*)(p.addend_mask + (long long)Lp * 256 * 8), 0, 256 * 8, 0x00020000);
                adm[J % H5_ADN] = __builtin_amdgcn_raw_buffer_load_b32(rsrcM, ((ph * 128 + t) * 8 + ch * 4), J * 128, 0);
            }
        }
    };
    auto epi_prefetch = [&](const int Lp) { h5_static_for<0, H5_AD>([&](auto jc) { ad_issue(jc, Lp); }); };
    auto epi_slice = [&](auto sc, f32x4_t (&accp)[2][8], const int Lp) {
        constexpr int S = decltype(sc)::value;
        if constexpr (S < 16) {
            if constexpr (S == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
            }
            if constexpr ((S & 1) == 0) {
                // pixel fragment J, both channel fragments: a lane holds channels 4g..4g+3 of each 16-channel fragment (8 bytes);
                // v_permlane16_swap exchanges the odd 16-lane rows of one with the even rows of the other, after which every lane
                // owns 8 CONSECUTIVE channels (16 bytes): half as many store instructions, and those are what the stream stalls on
                constexpr int J = S >> 1;
                const __amdgpu_buffer_rsrc_t rsrcD = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dst + (long long)Lp * 256 * 128), 0, 256 * 128, 0x00020000);
                unsigned pk[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float v[4] = {accp[i][J][0], accp[i][J][1], accp[i][J][2], accp[i][J][3]};
                    if constexpr (ADD != 0) {
                        h5_u32x2 a = ad[J % H5_ADN][i];
                        if constexpr (ADD == 2) {
                            // mask byte k = 2i + (g >> 1) of the loaded word covers channels 8k..8k+7 of this wave's 32; this lane's
                            // four channels are its low (g even) or high (g odd) nibble.  v_bfe_i32 turns a bit into an all-ones mask.
                            const unsigned bits = adm[J % H5_ADN] >> ((2 * i + (g >> 1)) * 8 + (g & 1) * 4);
                            const unsigned m0 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 0, 1), m1 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 1, 1);
                            const unsigned m2 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 2, 1), m3 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 3, 1);
                            a[0] &= (m0 & 0x0000ffffu) | (m1 & 0xffff0000u);
                            a[1] &= (m2 & 0x0000ffffu) | (m3 & 0xffff0000u);
                        }
                        v[0] += __uint_as_float(a[0] << 16); v[1] += __uint_as_float(a[0] & 0xffff0000u);
                        v[2] += __uint_as_float(a[1] << 16); v[3] += __uint_as_float(a[1] & 0xffff0000u);
                    }
                    pk[i][0] = pack_bf16x2(v[0], v[1]); pk[i][1] = pack_bf16x2(v[2], v[3]);
                    if constexpr (BST != 0) {
                        // g = the STORED bf16 value where the consuming BatchNorm's ReLU passed; sums of g and of g * x
                        const unsigned bits = bxm[J % H5_ADN] >> ((2 * i + (g >> 1)) * 8 + (g & 1) * 4);
                        const unsigned m0 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 0, 1), m1 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 1, 1);
                        const unsigned m2 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 2, 1), m3 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 3, 1);
                        const h5_u32x2 xv = bx[J % H5_ADN][i];
                        const float g0 = __uint_as_float((pk[i][0] << 16) & m0), g1 = __uint_as_float(pk[i][0] & 0xffff0000u & m1);
                        const float g2 = __uint_as_float((pk[i][1] << 16) & m2), g3 = __uint_as_float(pk[i][1] & 0xffff0000u & m3);
                        ssum[i][0] += g0; ssum[i][1] += g1; ssum[i][2] += g2; ssum[i][3] += g3;
                        ssq[i][0] = fmaf(g0, __uint_as_float(xv[0] << 16), ssq[i][0]); ssq[i][1] = fmaf(g1, __uint_as_float(xv[0] & 0xffff0000u), ssq[i][1]);
                        ssq[i][2] = fmaf(g2, __uint_as_float(xv[1] << 16), ssq[i][2]); ssq[i][3] = fmaf(g3, __uint_as_float(xv[1] & 0xffff0000u), ssq[i][3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ssum[i][r] += v[r]; ssq[i][r] += v[r] * v[r]; }
                    }
                }
                const h5_u32x2 lo = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);   // rows: (X0,Y0,X2,Y2) / (X1,Y1,X3,Y3)
                const h5_u32x2 hi = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                const h5_u32x4 outv = {lo[0], hi[0], lo[1], hi[1]};
                __builtin_amdgcn_raw_buffer_store_b128(outv, rsrcD, voffS, J * 2048, 0);
                store_b128_guard(outv);
                ad_issue(std::integral_constant<int, J + H5_AD>{}, Lp);
            }
        } else if constexpr (S == 16) {
            if constexpr (MODE == 0 || BST != 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ssum[i][r] = row16_sum(ssum[i][r]); ssq[i][r] = row16_sum(ssq[i][r]); }
            }
        } else {
            if ((MODE == 0 || BST != 0) && t == 0) {         // four lanes (g = 0..3): 2 x 16 bytes of sums and of sums of squares each
                const long long blk = 2LL * Lp + ph;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int co = ch * 32 + i * 16 + g * 4;
                    *(float4*)(p.stat + blk * 64 + co) = make_float4(ssum[i][0], ssum[i][1], ssum[i][2], ssum[i][3]);
                    *(float4*)(p.stat + ((long long)p.n_mblocks + blk) * 64 + co) = make_float4(ssq[i][0], ssq[i][1], ssq[i][2], ssq[i][3]);
                }
            }
        }
    };

    int L = h5_xcd_remap(blockIdx.x, NB);
    if (L >= p.n_tiles) return;
    h5_static_for<0, H5_KH>([&](auto kc) { halo_piece(kc, 0, L, true); });

    f32x4_t accA[2][8], accB[2][8];                          // accumulators of the even / odd tiles of this workgroup

    // One tile from halo buffer BUF.  Software pipeline: while its 18 MFMA batches run, the 11 halo pieces of the NEXT tile are
    // requested (into the buffer the previous tile has left) and the epilogue of the PREVIOUS tile is worked off, one slice per batch.
    auto tile = [&](auto bufc, auto prevc, const int L, const bool has_next) {
        constexpr int BUF = decltype(bufc)::value;
        constexpr bool PREV = decltype(prevc)::value;
        f32x4_t (&acc)[2][8] = BUF ? accB : accA;
        f32x4_t (&accp)[2][8] = BUF ? accA : accB;
        H5_STAMP(0);
        h5_wait_vmcnt<0>();                                  // this wave's pieces of the halo (and, first tile, of the filter) have landed
        __builtin_amdgcn_s_barrier();                        // ... everybody's; and every wave has left the taps of the previous tile
        H5_STAMP(1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        // 18 MFMA batches (tap U, K half H) of 16; the 10 fragment reads of batch s+1 are in flight while batch s runs
        uint4 wf[2][2], pf[2][8];
        auto issue_reads = [&](auto sc) {
            constexpr int S = decltype(sc)::value, U = S >> 1, H = S & 1, A = U / 3, B = U % 3, SET = S & 1;
            h5_static_for<0, 2>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                if constexpr (U < 4) wf[SET][I] = h5_read16<U * H5_WT_BYTES + I * 2048>(wa[H]);
                else wf[SET][I] = h5_read16<(U - 4) * H5_WT_BYTES + I * 2048>(wb[H]);
            });
            h5_static_for<0, 8>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                pf[SET][J] = h5_read16<((A + (J >> 1)) * PITCH + (J & 1) * 16) * 128>(pa[BUF][B][H]);
            });
        };
        issue_reads(std::integral_constant<int, 0>{});
        if constexpr (PREV) epi_prefetch(L - NB);
        h5_static_for<0, 18>([&](auto sc) {
            constexpr int S = decltype(sc)::value, SET = S & 1;
            if constexpr (S + 1 < 18) {
                issue_reads(std::integral_constant<int, S + 1>{});
                h5_wait_lgkmcnt<10>();                      // (in-order returns) the reads of batch S have landed
            } else {
                h5_wait_lgkmcnt<0>();
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = mma_chunk<bf16_tag>(wf[SET][i], pf[SET][j], acc[i][j]);
            if constexpr (PREV) epi_slice(sc, accp, L - NB);
            if constexpr (PREV) {
                // ask the scheduler to thread the slice's vector instructions through the MFMAs (in program order they would
                // run after the batch, with the MFMA pipe idle): one MFMA, then up to two VALU, sixteen times
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
            }
            if constexpr (S < H5_KH) halo_piece(std::integral_constant<int, S>{}, BUF ^ 1, has_next ? L + NB : L, has_next);
        });
        H5_STAMP(3);
    };

    // first tile (nothing to drain), then alternate the buffers; the last tile's epilogue runs on its own
    bool has_next = L + NB < p.n_tiles;
    tile(std::integral_constant<int, 0>{}, std::false_type{}, L, has_next);
    int parity = 0;
    while (has_next) {
        L += NB;
        has_next = L + NB < p.n_tiles;
        tile(std::integral_constant<int, 1>{}, std::true_type{}, L, has_next);
        parity = 1;
        if (!has_next) break;
        L += NB;
        has_next = L + NB < p.n_tiles;
        tile(std::integral_constant<int, 0>{}, std::true_type{}, L, has_next);
        parity = 0;
    }
    epi_prefetch(L);
    if (parity == 0) h5_static_for<0, 18>([&](auto sc) { epi_slice(sc, accA, L); });
    else h5_static_for<0, 18>([&](auto sc) { epi_slice(sc, accB, L); });
    // (the last tile's dead halo requests are LDS-DMA writes nothing above has waited for: do not let the wave end with one in flight)
    h5_wait_vmcnt<0>();
    H5_STAMP(4); H5_STAMP(5);
#endif
}

// returns 1 if the kernel handled the call: bf16, 64 -> 64 channels, 32x32 maps, one weight set, no pooled addend
int fb_conv3x3_halo5_takes(const fb_conv_args* a) {
    static const bool disabled = getenv("FB_DISABLE_HALO5") != nullptr;
    if (disabled || a->dtype != FB_BF16) return 0;
    if (a->R != 3 || a->S != 3 || a->stride != 1 || a->pad != 1) return 0;
    if (a->Hs != 32 || a->Ws != 32 || a->Hd != 32 || a->Wd != 32 || a->Cs != 64 || a->Cd != 64) return 0;
    if (a->imgs_per_wset > 0 && a->imgs_per_wset < a->n_img) return 0;          // per-chunk weight sets: the filter would not stay resident
    if (a->addend && a->addend_mode != 1) return 0;
    if (a->addend_mask && (!a->addend || a->mode != 1)) return 0;
    if (a->mode == 0 && !a->stat_partial) return 0;                            // (the forward variant always writes statistics)
    if (a->bst_x && (a->mode != 1 || !a->bst_mask || !a->stat_partial)) return 0;
    return 1;
}

int fb_try_conv3x3_halo5(const fb_conv_args* a, hipStream_t st) {
    if (!fb_conv3x3_halo5_takes(a)) return 0;
    const int n_cu = fb_persistent_cus();
    Halo5Params p;
    p.src = (const char*)a->src; p.wgt = (const char*)a->wgt; p.dst = (char*)a->dst; p.addend = (const char*)a->addend;
    p.stat = (a->mode == 0 || a->bst_x) ? a->stat_partial : nullptr;
    p.addend_mask = (const unsigned char*)a->addend_mask;
    p.bst_x = (const char*)a->bst_x; p.bst_mask = (const unsigned char*)a->bst_mask;
    p.n_img = a->n_img; p.mode = a->mode; p.addend_mode = a->addend ? 1 : 0;
    p.n_tiles = a->n_img * 4;
    p.n_mblocks = p.n_tiles * 2;
#ifdef FB_H5_TRACE
    extern long long* g_h5_trace;
    p.trace = g_h5_trace;
#endif
    const dim3 grid(p.n_tiles < n_cu ? p.n_tiles : n_cu);
    if (a->mode == 0) {
        hipLaunchKernelGGL((conv3x3s1_c64_halo5_kernel<0, 0>), grid, dim3(256), 0, st, p);
    } else if (a->bst_x) {
        if (a->addend && a->addend_mask) hipLaunchKernelGGL((conv3x3s1_c64_halo5_kernel<1, 2, 1>), grid, dim3(256), 0, st, p);
        else if (a->addend) hipLaunchKernelGGL((conv3x3s1_c64_halo5_kernel<1, 1, 1>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3x3s1_c64_halo5_kernel<1, 0, 1>), grid, dim3(256), 0, st, p);
    } else if (a->addend && a->addend_mask) {
        hipLaunchKernelGGL((conv3x3s1_c64_halo5_kernel<1, 2>), grid, dim3(256), 0, st, p);
    } else if (a->addend) {
        hipLaunchKernelGGL((conv3x3s1_c64_halo5_kernel<1, 1>), grid, dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL((conv3x3s1_c64_halo5_kernel<1, 0>), grid, dim3(256), 0, st, p);
    }
    return 1;
}
