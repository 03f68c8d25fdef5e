// Implicit-GEMM convolution (forward / input-gradient), LDS-direct variant with near-zero per-iteration address math.
//
// PMC on the register-staged kernel (profiles/r1_pmc_notes.md) showed it was bound by VALU/SALU *issue*, not by memory or
// MFMA: ~130 VALU + 75 SALU per K-step against 32 MFMAs, almost all of it implicit-GEMM addressing.  This variant keeps
// the tiling/fragment layout/swizzle/epilogue of conv_igemm.hip (128 pixels x 64|128 channels per workgroup, D[co][pixel],
// K-step = 128 bytes of channels of one tap) and removes the per-iteration arithmetic:
//   * tile rows go HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR round trip, no ds_write pass).  The per-lane
//     32-bit row offset (voffset) is computed once per TAP; the channel-slice/tap advance is a wave-uniform SGPR soffset.
//     Zero padding needs no select on pointers and no zero page: padded rows use an out-of-range voffset and the buffer
//     bounds check returns zeros.  The XOR swizzle sits on the SOURCE side (lane L fetches chunk (L&7)^(row&7)); the LDS
//     image stays lane-linear as the instruction requires.
//   * fragment reads are inline-asm ds_read_b128 with compile-time immediate offsets from four per-thread base addresses
//     (no per-iteration VALU), waited for with counted lgkmcnt.  Being asm they are also invisible to hipcc's
//     "pending LDS-DMA => vmcnt(0) before every ds_read" rule, so the loads of tile t+1 really overlap the MFMAs of tile t.
//   * 1-D grid with an XCD-aware remap: the co-tiles of one pixel tile are adjacent on one XCD and share the gathered rows in L2.
#include "common.h"
#include "conv_params.h"

#include <type_traits>

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);     // keep MFMAs below the wait (cdna_hip_programming.md 5.4 rule 18)
}
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
template <int OFF> __device__ __forceinline__ uint4 lds_read16(unsigned byte_addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ int xcd_remap1d(int b, int n) {
    const int q = n >> 3, r = n & 7, xcd = b & 7, slot = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
#define FB_OOB 0x80000000u

// STAGES 2: the loads of K-step t+1 are in flight while step t is multiplied (64 KiB: two workgroups per CU); STAGES 1: one 32 KiB (24 KiB)
// stage -- the waves wait for their own loads, and three resident workgroups per CU cover each other's waits, prologues and epilogues
template <typename T, int BN_CO, int STAGES = 2>
__global__ __launch_bounds__(256) void conv_igemm_v3_kernel(const ConvParams p, const int mblocks, const int n_co) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (buffer-resource builtins do not exist there)
    constexpr int EB = ET<T>::EB;
    constexpr int BKE = 128 / EB;
    constexpr int WROWS = BN_CO / 32;
    constexpr int FI = BN_CO / 32, FJ = 4;
    constexpr int TILE_BYTES = (128 + BN_CO) * 128;
    __shared__ __attribute__((aligned(16))) char lds[STAGES * TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_co = wave >> 1, wave_px = wave & 1;
    // block coordinates go through integer divisions (VALU): readfirstlane makes them provably wave-uniform again, otherwise
    // hipcc wraps every buffer op that depends on them in a waterfall loop (cdna_hip_programming.md T20)
    const int L = xcd_remap1d(blockIdx.x, gridDim.x);
    const int co_blk = __builtin_amdgcn_readfirstlane(L % n_co), t2 = L / n_co;
    // the four parity classes of a stride-2 input gradient read the same dY pixels (with 0/1-pixel offsets): class is the
    // fast index so that they run side by side on one XCD and share them in L2 (PMC: dY was fetched four times from HBM)
    const int classes = (int)gridDim.x / (mblocks * n_co);
    const int cls = __builtin_amdgcn_readfirstlane(t2 % classes), mblk = __builtin_amdgcn_readfirstlane(t2 / classes);
    const int cpy = (p.os == 2) ? (cls >> 1) : 0, cpx = (p.os == 2) ? (cls & 1) : 0;
    const int lrow = tid >> 3;
    const int chunk = (tid & 7) ^ (lrow & 7);        // logical chunk this lane fetches (source-side swizzle)

    int a_pix[4], a_y[4], a_x[4];
    const int qHW = p.qH * p.qW;
    // source pixels are counted from the tile's first image and the descriptor starts there: the 32-bit offsets stay small whatever the
    // tensor's size (chunk groups beyond 2^31 bytes per tensor)
    const int n_first = __builtin_amdgcn_readfirstlane((mblk * 128) / qHW);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = mblk * 128 + lrow + 32 * i;
        if (m < p.M) {
            const int n = m / qHW, rem = m - n * qHW, qy = rem / p.qW, qx = rem - qy * p.qW;
            a_y[i] = qy * p.ss; a_x[i] = qx * p.ss;
            a_pix[i] = (n - n_first) * p.Hs * p.Ws + a_y[i] * p.Ws + a_x[i];
        } else {
            a_pix[i] = 0; a_y[i] = -(1 << 28); a_x[i] = 0;
        }
    }
    const int wset = __builtin_amdgcn_readfirstlane(((mblk * 128) / qHW) / p.imgs_per_wset);
    const int taps = p.R * p.S;
    const int row_b = p.Cs * EB;                                   // bytes of one pixel / one (co, tap) weight row
    const long long img_b = (long long)p.Hs * p.Ws * row_b;
    const long long left_b = (long long)(p.n_img - n_first) * img_b;
    const __amdgpu_buffer_rsrc_t rsrcA =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + (long long)n_first * img_b), 0, (int)(left_b < 0x7fffffffLL ? left_b : 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wgt + (long long)wset * p.wset_stride_bytes), 0, p.Cd * taps * row_b, 0x00020000);
    unsigned voffW[WROWS];
#pragma unroll
    for (int i = 0; i < WROWS; ++i) voffW[i] = (unsigned)((co_blk * BN_CO + lrow + 32 * i) * taps * row_b + chunk * 16);
    const bool ctail = (p.Cs % BKE) != 0;                           // bf16 with Cs = 32 (mod 64): upper half of the K-step is padding

    // ---- tap enumeration (block-uniform) --------------------------------------------------------------------------------
    auto tap_valid = [&](int r, int s, int& dy, int& dx) -> bool {
        if (p.mode == 0) { dy = r - p.pad; dx = s - p.pad; return true; }
        if (p.os == 1) { dy = p.pad - r; dx = p.pad - s; return true; }
        const int vy = cpy + p.pad - r, vx = cpx + p.pad - s;
        if ((vy & 1) || (vx & 1)) return false;
        dy = vy >> 1; dx = vx >> 1;
        return true;
    };
    const int kc = (p.Cs + BKE - 1) / BKE;
    int n_valid = 0;
    for (int r = 0; r < p.R; ++r) for (int s = 0; s < p.S; ++s) { int dy, dx; n_valid += tap_valid(r, s, dy, dx) ? 1 : 0; }
    const int n_iter = n_valid * kc;

    unsigned voffA[4];
    int cur_r = 0, cur_s = -1, cur_c = kc, cur_t = 0;
    auto advance = [&]() {               // next (tap, channel slice); per-tap work: the four row offsets
        if (++cur_c >= kc) {
            cur_c = 0;
            int dy = 0, dx = 0;
            do { if (++cur_s >= p.S) { cur_s = 0; ++cur_r; } } while (cur_r < p.R && !tap_valid(cur_r, cur_s, dy, dx));
            cur_t = cur_r * p.S + cur_s;
            const int dpix = dy * p.Ws + dx;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sy = a_y[i] + dy, sx = a_x[i] + dx;
                const bool ok = (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
                voffA[i] = ok ? (unsigned)((a_pix[i] + dpix) * row_b + chunk * 16) : FB_OOB;
            }
        }
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    auto issue = [&](int stage) {          // stage is wave-uniform; LDS destinations are SGPR arithmetic -> M0
        char* tile = lds + stage * TILE_BYTES + wave * 1024;
        const int c0b = cur_c * 128;                                  // byte offset of the channel slice
        const int soffW = cur_t * row_b + c0b;
        if (ctail && (c0b + 64 >= row_b)) {                           // rare: bf16 slice with only 32 valid channels
            const bool cbad = c0b + chunk * 16 >= row_b;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(tile + i * 4096), 16,
                                                         cbad ? FB_OOB : voffA[i], c0b, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(tile + i * 4096), 16, voffA[i], c0b, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WROWS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)(tile + (16 + 4 * i) * 1024), 16, voffW[i],
                                                     soffW, 0, 0);
    };

    f32x4_t acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float hs_src = 1.f, hs_wgt = 1.f, hs_inv = 1.f;         // f32h: per-tensor power-of-two scales
    if constexpr (is_hsplit<T>::value) {
        const int img0 = (mblk * 128) / qHW;                 // (a 128-pixel block never straddles two chunks)
        hs_src = fb_pow2_scale(p.amax_src[img0 / p.amax_imgs]); hs_wgt = fb_pow2_scale(p.amax_wgt[wset]);
        hs_inv = 1.f / (hs_src * hs_wgt);
    }

    const unsigned pc0 = ((lane >> 4) ^ (lane & 7)) * 16, pc1 = (((lane >> 4) + 4) ^ (lane & 7)) * 16;
    const unsigned wb = lds0 + (128 + wave_co * (BN_CO / 2) + (lane & 15)) * 128;
    const unsigned pb = lds0 + (wave_px * 64 + (lane & 15)) * 128;

    if (n_iter > 0) {
        if constexpr (STAGES == 2) {
            advance(); issue(0);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        for (int it = 0; it < n_iter; ++it) {
            const int cur = STAGES == 2 ? (it & 1) : 0;
            if constexpr (STAGES == 2) {
                if (it + 1 < n_iter) { advance(); issue(cur ^ 1); }
            } else {
                advance(); issue(0);
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
            const unsigned so = cur * TILE_BYTES;
            const unsigned w0 = wb + pc0 + so, w1 = wb + pc1 + so, p0 = pb + pc0 + so, p1 = pb + pc1 + so;
            uint4 wf0[FI], pf0[FJ], wf1[FI], pf1[FJ];
            static_for<0, FI>([&](auto i) { wf0[decltype(i)::value] = lds_read16<decltype(i)::value * 2048>(w0); });
            static_for<0, FJ>([&](auto j) { pf0[decltype(j)::value] = lds_read16<decltype(j)::value * 2048>(p0); });
            static_for<0, FI>([&](auto i) { wf1[decltype(i)::value] = lds_read16<decltype(i)::value * 2048>(w1); });
            static_for<0, FJ>([&](auto j) { pf1[decltype(j)::value] = lds_read16<decltype(j)::value * 2048>(p1); });
            if constexpr (is_hsplit<T>::value) {     // fp32 operands as two scaled fp16 pieces each, three MFMAs per fragment pair (common.h)
                wait_lgkmcnt<0>();
                split2h_t sw[FI], sp[FJ];
#pragma unroll
                for (int i = 0; i < FI; ++i) { sw[i].h = __builtin_bit_cast(f16x8_t, wf0[i]); sw[i].l = __builtin_bit_cast(f16x8_t, wf1[i]); }   // planes (fb_weight_prep)
#pragma unroll
                for (int j = 0; j < FJ; ++j) sp[j] = split_h2x8(pf0[j], pf1[j], hs_src);
#pragma unroll
                for (int i = 0; i < FI; ++i)
#pragma unroll
                    for (int j = 0; j < FJ; ++j) acc[i][j] = mma_split3h(sw[i], sp[j], acc[i][j], hs_inv);
            } else if constexpr (is_split<T>::value) {      // fp32 operands as three bf16 pieces each, six MFMAs per fragment pair (common.h)
                wait_lgkmcnt<0>();
                split3_t sp[FJ];
#pragma unroll
                for (int j = 0; j < FJ; ++j) sp[j] = split_f32x8(pf0[j], pf1[j]);
                // one weight fragment at a time, FJ zero-started chains in flight, the split of fragment i + 1 threaded through the MFMAs of fragment i
                // (round 5; before: all eight splits, then 96 MFMAs as 16 chains of six DEPENDENT products)
                split3_t sw = split_f32x8(wf0[0], wf1[0]);
                static_for<0, FI>([&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    split3_t swn = sw;
                    if constexpr (I + 1 < FI) {
                        swn = split_f32x8(wf0[I + 1], wf1[I + 1]);
                        mma_split6_row_mix<FJ, 2>(sw, sp, acc[I]);
                    } else {
                        mma_split6_row<FJ>(sw, sp, acc[I]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    sw = swn;
                });
            } else {
            wait_lgkmcnt<FI + FJ>();                 // first half has landed (LDS returns in order)
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) acc[i][j] = mma_chunk<T>(wf0[i], pf0[j], acc[i][j]);
            wait_lgkmcnt<0>();
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) acc[i][j] = mma_chunk<T>(wf1[i], pf1[j], acc[i][j]);
            }
            if constexpr (STAGES == 2) wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
    }

    // ---- epilogue: (+addend) -> dst, per-channel partial statistics ---------------------------------------------------------
    float ssum[FI][4], ssq[FI][4];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
        const int m = mblk * 128 + wave_px * 64 + j * 16 + (lane & 15);
        const bool valid = m < p.M;                     // (rows past M were zero-filled: their accumulators are 0)
        const int mc = valid ? m : p.M - 1;
        const int n = mc / qHW, rem = mc - n * qHW, qy = rem / p.qW, qx = rem - qy * p.qW;
        const int oy = qy * p.os + cpy, ox = qx * p.os + cpx;
        const long long pix = ((long long)n * p.Hd + oy) * p.Wd + ox;
        unsigned pk[FI][2];
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int co = co_blk * BN_CO + wave_co * (BN_CO / 2) + i * 16 + (lane >> 4) * 4;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.addend_mode != 0 && valid) {
                const long long apix = p.addend_mode == 1 ? pix : ((long long)n * (p.Hd >> 1) + (oy >> 1)) * (p.Wd >> 1) + (ox >> 1);
                const float sc = p.addend_mode == 1 ? 1.f : 0.25f;
                const char* ap = p.addend + (apix * p.Cd + co) * EB;
                // addend_mask (round 6; identity blocks whose input gradient no specialised kernel takes: every Bottleneck block in fp32 storage -- the regulariser's
                // passes -- and the 512-channel ones in bf16): the addend is the gradient entering the block, counted where the ReLU bit of the block's output is set
                // (one byte per 16-byte vector of that tensor: 4 fp32 / 8 bf16 channels); d * (out > 0) is never materialised.  Same sums as adding the masked tensor.
                unsigned mb = 0xffu;
                if (p.addend_mask != nullptr) {
                    const long long e = apix * p.Cd + co;
                    mb = EB == 4 ? p.addend_mask[e >> 2] : (unsigned)p.addend_mask[e >> 3] >> (co & 4);
                }
                if constexpr (EB == 4) {
                    const float4 a = *(const float4*)ap;
                    v[0] += (mb & 1u) ? sc * a.x : 0.f; v[1] += (mb & 2u) ? sc * a.y : 0.f; v[2] += (mb & 4u) ? sc * a.z : 0.f; v[3] += (mb & 8u) ? sc * a.w : 0.f;
                } else {
                    const uint2 a = *(const uint2*)ap;
                    v[0] += (mb & 1u) ? sc * __uint_as_float(a.x << 16) : 0.f; v[1] += (mb & 2u) ? sc * __uint_as_float(a.x & 0xffff0000u) : 0.f;
                    v[2] += (mb & 4u) ? sc * __uint_as_float(a.y << 16) : 0.f; v[3] += (mb & 8u) ? sc * __uint_as_float(a.y & 0xffff0000u) : 0.f;
                }
            }
            if constexpr (EB == 4) {
                if (valid) *(float4*)(p.dst + (pix * p.Cd + co) * EB) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                pk[i][0] = pack_bf16x2(v[0], v[1]); pk[i][1] = pack_bf16x2(v[2], v[3]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[i][r] += v[r]; ssq[i][r] += v[r] * v[r]; }
        }
        if constexpr (EB == 2) {
            // a lane holds channels 4g..4g+3 of each 16-channel fragment (8 bytes); v_permlane16_swap exchanges the odd 16-lane rows of
            // fragment i with the even rows of fragment i+1: every lane then owns 8 consecutive channels ({0,16,8,24}[g] of the pair's
            // 32) and one 16-byte store per pair replaces two 8-byte ones (the epilogue is bound by store issue).  All lanes take part
            // in the swap; only the store is predicated.
#pragma unroll
            for (int i = 0; i < FI; i += 2) {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                const u32x2_t lo = __builtin_amdgcn_permlane16_swap(pk[i][0], pk[i + 1][0], false, false);
                const u32x2_t hi = __builtin_amdgcn_permlane16_swap(pk[i][1], pk[i + 1][1], false, false);
                const int g = lane >> 4;
                const int co = co_blk * BN_CO + wave_co * (BN_CO / 2) + i * 16 + (g & 1) * 16 + (g >> 1) * 8;
                if (valid) *(uint4*)(p.dst + (pix * p.Cd + co) * EB) = make_uint4(lo[0], hi[0], lo[1], hi[1]);
            }
        }
    }
    if (p.stat != nullptr) {
        float* red = (float*)lds;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = ssum[i][r], b = ssq[i][r];
                a = row16_sum(a); b = row16_sum(b);
                if ((lane & 15) == 0) {
                    const int col = wave_co * (BN_CO / 2) + i * 16 + (lane >> 4) * 4 + r;
                    red[(wave_px * BN_CO + col) * 2] = a; red[(wave_px * BN_CO + col) * 2 + 1] = b;
                }
            }
        __syncthreads();
        if (tid < BN_CO) {
            const float a = red[tid * 2] + red[(BN_CO + tid) * 2];
            const float b = red[tid * 2 + 1] + red[(BN_CO + tid) * 2 + 1];
            const long long blk = (long long)cls * mblocks + mblk;
            p.stat[blk * p.Cd + co_blk * BN_CO + tid] = a;
            p.stat[((long long)p.n_mblocks + blk) * p.Cd + co_blk * BN_CO + tid] = b;
        }
    }
#endif
}

template <typename T> static void launch(const ConvParams& p, int classes, hipStream_t st) {
    const int mblocks = (p.M + 127) / 128;
    if constexpr (std::is_same<T, bf16_tag>::value) {
        // bf16: one stage, three workgroups per CU (12 544 images, stride-2 forward 64->128 / 128->256 / 256->512: 936 / 587 / 493 us against
        // 1001 / 669 / 561 with two stages and two workgroups); FB_IGEMM_STAGES=2 selects the double-buffered form (read per call: tests compare)
        const char* e = fb_getenv_experimental("FB_IGEMM_STAGES");
        if (!(e && atoi(e) == 2)) {
            if (p.Cd % 128 == 0 && (long long)mblocks * (p.Cd / 128) * classes >= 512) {
                const int n_co = p.Cd / 128;
                hipLaunchKernelGGL((conv_igemm_v3_kernel<T, 128, 1>), dim3(mblocks * n_co * classes), dim3(256), 0, st, p, mblocks, n_co);
            } else {
                const int n_co = p.Cd / 64;
                hipLaunchKernelGGL((conv_igemm_v3_kernel<T, 64, 1>), dim3(mblocks * n_co * classes), dim3(256), 0, st, p, mblocks, n_co);
            }
            return;
        }
    }
    if (p.Cd % 128 == 0 && (long long)mblocks * (p.Cd / 128) * classes >= 512) {
        const int n_co = p.Cd / 128;
        hipLaunchKernelGGL((conv_igemm_v3_kernel<T, 128>), dim3(mblocks * n_co * classes), dim3(256), 0, st, p, mblocks, n_co);
    } else {
        const int n_co = p.Cd / 64;
        hipLaunchKernelGGL((conv_igemm_v3_kernel<T, 64>), dim3(mblocks * n_co * classes), dim3(256), 0, st, p, mblocks, n_co);
    }
}

int fb_igemm_glds_fits(const ConvParams& p, int dtype) {
    const int EB = dtype == FB_F32 ? 4 : 2;
    // a 128-pixel tile addresses the images it covers (+ the next one) from its own descriptor base
    const long long bytesA = (long long)(128 / (p.qH * p.qW) + 2) * p.Hs * p.Ws * p.Cs * EB, bytesW = (long long)p.Cd * p.R * p.S * p.Cs * EB;
    return !(bytesA >= (1LL << 31) || bytesW >= (1LL << 31) || (long long)p.n_img * p.qH * p.qW >= (1LL << 31));
}

// returns 1 if launched, 0 if the tensors are too large for 32-bit buffer offsets (caller falls back to conv_igemm.hip)
int fb_launch_igemm_glds(const ConvParams& p, int classes, int dtype, hipStream_t st) {
    if (!fb_igemm_glds_fits(p, dtype)) return 0;
    if (dtype == FB_F32) {
        if (p.amax_src && p.amax_wgt) launch<f32h_tag>(p, classes, st);
        else if (fb_f32_split_enabled()) launch<f32s_tag>(p, classes, st);
        else launch<float>(p, classes, st);
    }
    else launch<bf16_tag>(p, classes, st);
    return 1;
}
