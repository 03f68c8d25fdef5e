// Implicit-GEMM convolution (forward / input gradient) for fp32 values whose OPERANDS ARRIVE PRE-SPLIT: every fp32 value as its three bf16 pieces
// (x = h + m + l exactly: common.h, "bf16x6"), written once by the kernel that produced the tensor -- FB_F32P storage ("planes").
//
// Why (round 6): the fp32-storage convolutions (conv_igemm_glds.hip <f32s_tag>) split every fragment inside the kernel, 44 VALU per 8 values and lane, and
// every fragment is split by both waves that multiply it: per 32-deep K-step a wave issues 352 VALU (1408 cycles of the vector pipe) next to 96 MFMAs
// (1536 cycles of the matrix pipe), and the two only overlap in part -- 0.36-0.40 of the matrix peak, the same fraction in every fp32 kernel of the
// regulariser's passes (BASELINE configs 3 and 5).  With the pieces in memory a K-step is LDS-DMA + fragment reads + MFMAs, like the bf16 kernel with
// three reads per fragment and six products per fragment pair; the split happens ONCE per element, in the HBM-bound pass that writes the tensor.
//
// Layout of a planes tensor (activations [pixel][C], weights [(co, tap)][C]; C a multiple of 32): a row is 6 C bytes -- C high pieces, C middle pieces, C low
// pieces (bf16) -- and inside every group of 32 channels the pieces sit in the K-SLOT ORDER of the fp32 kernels' fragment reads (value j of the group at slot
// 8 * (j % 16 / 4) + 4 * (j / 16) + j % 4: lane group g of an MFMA step multiplies channels {4g .. 4g + 3, 16 + 4g .. 16 + 4g + 3}, there from two 16-byte fp32
// chunks, here from ONE 16-byte chunk per plane) -- so that this kernel adds the same products in the same order as the <f32s_tag> kernels: SAME BITS.
//
// Tile: 128 pixels x 64 | 128 output channels, K-step = 32 channels of one tap = three 64-byte pieces per row; LDS image per plane and operand: rows of 64 bytes,
// 16-byte chunk c of row r at chunk position c ^ ((r >> 2) & 3) (source-side swizzle: conflict-free ds_read_b128 for 16 rows x one chunk); one LDS stage
// (36 | 48 KiB), three workgroups per CU cover each other's waits (as the bf16 implicit GEMM).  Epilogue = conv_igemm_glds.hip's fp32 one (addend, addend
// through the ReLU bitmask, BatchNorm partial sums).
#include "common.h"
#include "conv_params.h"

#include <type_traits>

namespace {
template <int N> __device__ __forceinline__ void p3_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void p3_wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
typedef __attribute__((ext_vector_type(4))) unsigned p3_u32x4;
template <int OFF> __device__ __forceinline__ bf16x8_t p3_lds_read16(unsigned byte_addr) {
    p3_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return __builtin_bit_cast(bf16x8_t, v);
}
template <int I, int N, typename F> __device__ __forceinline__ void p3_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); p3_static_for<I + 1, N>(f); }
}
__device__ __forceinline__ int p3_xcd_remap1d(int b, int n) {
    const int q = n >> 3, r = n & 7, xcd = b & 7, slot = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
constexpr unsigned P3_OOB = 0x80000000u;
}  // namespace

template <int BN_CO, int STAGES>
__global__ __launch_bounds__(256) void conv_igemm_p3_kernel(const ConvParams p, const int mblocks, const int n_co) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int FI = BN_CO / 32, FJ = 4;
    constexpr int PIECES_B = BN_CO / 64;                    // 16-row LDS-DMA pieces per wave, plane and K-step: filter rows (pixel rows: 2)
    constexpr int PLANE_A = 128 * 64, PLANE_B = BN_CO * 64;
    constexpr int TILE_BYTES = 3 * (PLANE_A + PLANE_B);
    __shared__ __attribute__((aligned(16))) char lds[(STAGES == 0 ? 1 : STAGES) * TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_co = wave >> 1, wave_px = wave & 1;
    const int L = p3_xcd_remap1d(blockIdx.x, gridDim.x);
    const int co_blk = __builtin_amdgcn_readfirstlane(L % n_co), t2 = L / n_co;
    const int classes = (int)gridDim.x / (mblocks * n_co);
    const int cls = __builtin_amdgcn_readfirstlane(t2 % classes), mblk = __builtin_amdgcn_readfirstlane(t2 / classes);
    const int cpy = (p.os == 2) ? (cls >> 1) : 0, cpx = (p.os == 2) ? (cls & 1) : 0;
    // LDS-DMA: a wave instruction fills 16 rows x 64 bytes of one plane; lane -> row (lane >> 2) of the piece, chunk position lane & 3, which holds the
    // logical chunk (lane & 3) ^ ((row >> 2) & 3) = (lane & 3) ^ ((lane >> 4) & 3)   (pieces start at multiples of 16 rows)
    const int drow = lane >> 2;
    const int dchunk = (lane & 3) ^ ((lane >> 4) & 3);

    int a_pix[2], a_y[2], a_x[2];
    const int qHW = p.qH * p.qW;
    const int n_first = __builtin_amdgcn_readfirstlane((mblk * 128) / qHW);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = mblk * 128 + (wave * 2 + i) * 16 + drow;
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
    const int plane_b = p.Cs * 2;                                  // bytes of one plane of a row
    const int row_b = 3 * plane_b;                                 // bytes of one pixel / one (co, tap) weight row
    const long long img_b = (long long)p.Hs * p.Ws * row_b;
    const long long left_b = (long long)(p.n_img - n_first) * img_b;
    const __amdgpu_buffer_rsrc_t rsrcA =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + (long long)n_first * img_b), 0, (int)(left_b < 0x7fffffffLL ? left_b : 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wgt + (long long)wset * p.wset_stride_bytes), 0, p.Cd * taps * row_b, 0x00020000);
    unsigned voffW[PIECES_B];
#pragma unroll
    for (int i = 0; i < PIECES_B; ++i) voffW[i] = (unsigned)((co_blk * BN_CO + (wave * PIECES_B + i) * 16 + drow) * taps * row_b + dchunk * 16);

    auto tap_valid = [&](int r, int s, int& dy, int& dx) -> bool {
        if (p.mode == 0) { dy = r - p.pad; dx = s - p.pad; return true; }
        if (p.os == 1) { dy = p.pad - r; dx = p.pad - s; return true; }
        const int vy = cpy + p.pad - r, vx = cpx + p.pad - s;
        if ((vy & 1) || (vx & 1)) return false;
        dy = vy >> 1; dx = vx >> 1;
        return true;
    };
    const int kc = p.Cs / 32;
    int n_valid = 0;
    for (int r = 0; r < p.R; ++r) for (int s = 0; s < p.S; ++s) { int dy, dx; n_valid += tap_valid(r, s, dy, dx) ? 1 : 0; }
    const int n_iter = n_valid * kc;

    unsigned voffA[2];
    int cur_r = 0, cur_s = -1, cur_c = kc, cur_t = 0;
    auto advance = [&]() {
        if (++cur_c >= kc) {
            cur_c = 0;
            int dy = 0, dx = 0;
            do { if (++cur_s >= p.S) { cur_s = 0; ++cur_r; } } while (cur_r < p.R && !tap_valid(cur_r, cur_s, dy, dx));
            cur_t = cur_r * p.S + cur_s;
            const int dpix = dy * p.Ws + dx;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int sy = a_y[i] + dy, sx = a_x[i] + dx;
                const bool ok = (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
                voffA[i] = ok ? (unsigned)((a_pix[i] + dpix) * row_b + dchunk * 16) : P3_OOB;
            }
        }
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    auto issue = [&](int stage) {
        char* tile = lds + stage * TILE_BYTES;
        const int c0b = cur_c * 64;                                   // byte offset of the channel slice inside a plane
        const int soffW = cur_t * row_b + c0b;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(tile + pl * PLANE_A + (wave * 2 + i) * 1024), 16, voffA[i],
                                                         pl * plane_b + c0b, 0, 0);
#pragma unroll
            for (int i = 0; i < PIECES_B; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)(tile + 3 * PLANE_A + pl * PLANE_B + (wave * PIECES_B + i) * 1024), 16,
                                                         voffW[i], pl * plane_b + soffW, 0, 0);
        }
    };

    f32x4_t acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // fragment reads: row (16 j + (lane & 15)) of the wave's 64 pixels / (16 i + (lane & 15)) of its channels, logical chunk g = lane >> 4
    const unsigned fchunk = (unsigned)(((lane >> 4) ^ ((lane >> 2) & 3)) * 16);
    const unsigned pb = lds0 + (wave_px * 64 + (lane & 15)) * 64 + fchunk;
    const unsigned wb = lds0 + 3 * PLANE_A + (wave_co * (BN_CO / 2) + (lane & 15)) * 64 + fchunk;

    if (n_iter > 0) {
        if constexpr (STAGES == 0) {
            // ONE LDS stage, software-pipelined through the registers: a K-step's fragments (24 x 16 bytes per lane) are all read before its first MFMA, so the
            // stage is free while the step is multiplied -- the NEXT step's LDS-DMA is issued right behind the reads and flies under the 96 MFMAs
            advance(); issue(0);
            for (int it = 0; it < n_iter; ++it) {
                p3_wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();                       // this step's rows have landed (every wave's)
                split3_t sp[FJ], sw[FI];
                p3_static_for<0, FJ>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    sp[j].h = p3_lds_read16<j * 1024>(pb); sp[j].m = p3_lds_read16<PLANE_A + j * 1024>(pb); sp[j].l = p3_lds_read16<2 * PLANE_A + j * 1024>(pb);
                });
                p3_static_for<0, FI>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    sw[i].h = p3_lds_read16<i * 1024>(wb); sw[i].m = p3_lds_read16<PLANE_B + i * 1024>(wb); sw[i].l = p3_lds_read16<2 * PLANE_B + i * 1024>(wb);
                });
                p3_wait_lgkmcnt<0>();
                __builtin_amdgcn_s_barrier();                       // every wave holds its fragments: the stage may be overwritten
                if (it + 1 < n_iter) { advance(); issue(0); }
                p3_static_for<0, FI>([&](auto ic) { constexpr int I = decltype(ic)::value; mma_split6_row<FJ>(sw[I], sp, acc[I]); });
            }
        } else {
        if constexpr (STAGES == 2) {
            advance(); issue(0);
            p3_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        for (int it = 0; it < n_iter; ++it) {
            const int cur = STAGES == 2 ? (it & 1) : 0;
            if constexpr (STAGES == 2) {
                if (it + 1 < n_iter) { advance(); issue(cur ^ 1); }
            } else {
                advance(); issue(0);
                p3_wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
            const unsigned so = cur * TILE_BYTES;
            split3_t sp[FJ], sw[FI];
            p3_static_for<0, FJ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                sp[j].h = p3_lds_read16<j * 1024>(pb + so); sp[j].m = p3_lds_read16<PLANE_A + j * 1024>(pb + so); sp[j].l = p3_lds_read16<2 * PLANE_A + j * 1024>(pb + so);
            });
            p3_static_for<0, FI>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                sw[i].h = p3_lds_read16<i * 1024>(wb + so); sw[i].m = p3_lds_read16<PLANE_B + i * 1024>(wb + so); sw[i].l = p3_lds_read16<2 * PLANE_B + i * 1024>(wb + so);
            });
            // one weight fragment at a time against the FJ pixel fragments: FJ zero-started chains of six products (mma_split6_row: the order of the <f32s_tag> kernel)
            p3_static_for<0, FI>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                p3_wait_lgkmcnt<3 * (FI - 1 - I)>();             // (LDS returns in order: the pixel fragments and weight fragments 0 .. I have landed)
                mma_split6_row<FJ>(sw[I], sp, acc[I]);
            });
            if constexpr (STAGES == 2) p3_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        }
    }

    // ---- epilogue: (+addend) -> dst (fp32), per-channel partial statistics: conv_igemm_glds.hip's ------------------------------------------------------
    float ssum[FI][4], ssq[FI][4];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
        const int m = mblk * 128 + wave_px * 64 + j * 16 + (lane & 15);
        const bool valid = m < p.M;
        const int mc = valid ? m : p.M - 1;
        const int n = mc / qHW, rem = mc - n * qHW, qy = rem / p.qW, qx = rem - qy * p.qW;
        const int oy = qy * p.os + cpy, ox = qx * p.os + cpx;
        const long long pix = ((long long)n * p.Hd + oy) * p.Wd + ox;
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int co = co_blk * BN_CO + wave_co * (BN_CO / 2) + i * 16 + (lane >> 4) * 4;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.addend_mode != 0 && valid) {
                const long long apix = p.addend_mode == 1 ? pix : ((long long)n * (p.Hd >> 1) + (oy >> 1)) * (p.Wd >> 1) + (ox >> 1);
                const float sc = p.addend_mode == 1 ? 1.f : 0.25f;
                const float4 a = *(const float4*)(p.addend + (apix * p.Cd + co) * 4);
                unsigned mb = 0xffu;
                if (p.addend_mask != nullptr) mb = p.addend_mask[(apix * p.Cd + co) >> 2];
                v[0] += (mb & 1u) ? sc * a.x : 0.f; v[1] += (mb & 2u) ? sc * a.y : 0.f; v[2] += (mb & 4u) ? sc * a.z : 0.f; v[3] += (mb & 8u) ? sc * a.w : 0.f;
            }
            if (valid) *(float4*)(p.dst + (pix * p.Cd + co) * 4) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[i][r] += v[r]; ssq[i][r] += v[r] * v[r]; }
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

// returns 1 if launched, 0 if a tile's images do not fit 32-bit buffer offsets
int fb_launch_igemm_planes(const ConvParams& p, int classes, hipStream_t st) {
    const long long bytesA = (long long)(128 / (p.qH * p.qW) + 2) * p.Hs * p.Ws * p.Cs * 6, bytesW = (long long)p.Cd * p.R * p.S * p.Cs * 6;
    if (bytesA >= (1LL << 31) || bytesW >= (1LL << 31) || (long long)p.n_img * p.qH * p.qW >= (1LL << 31)) return 0;
    const int mblocks = (p.M + 127) / 128;
    static const int stages = getenv("FB_P3_STAGES") ? atoi(getenv("FB_P3_STAGES")) : 1;
    static const bool narrow = getenv("FB_P3_NARROW") != nullptr;
    const bool wide = !narrow && p.Cd % 128 == 0 && (long long)mblocks * (p.Cd / 128) * classes >= 512;
    const int n_co = p.Cd / (wide ? 128 : 64);
    const dim3 grid(mblocks * n_co * classes);
    if (stages == 2) {
        if (wide) hipLaunchKernelGGL((conv_igemm_p3_kernel<128, 2>), grid, dim3(256), 0, st, p, mblocks, n_co);
        else hipLaunchKernelGGL((conv_igemm_p3_kernel<64, 2>), grid, dim3(256), 0, st, p, mblocks, n_co);
    } else if (stages == 0) {
        if (wide) hipLaunchKernelGGL((conv_igemm_p3_kernel<128, 0>), grid, dim3(256), 0, st, p, mblocks, n_co);
        else hipLaunchKernelGGL((conv_igemm_p3_kernel<64, 0>), grid, dim3(256), 0, st, p, mblocks, n_co);
    } else {
        if (wide) hipLaunchKernelGGL((conv_igemm_p3_kernel<128, 1>), grid, dim3(256), 0, st, p, mblocks, n_co);
        else hipLaunchKernelGGL((conv_igemm_p3_kernel<64, 1>), grid, dim3(256), 0, st, p, mblocks, n_co);
    }
    return 1;
}

// ---- fp32 rows -> planes rows (activations [pixel][C], weight copies [(co, tap)][C]) ---------------------------------------------------------------------------------
// one thread per 16-byte fp32 vector (4 channels): vector v of a 32-channel group (channels 4v .. 4v + 3) goes to bytes 16 (v % 4) + 8 (v / 4) of the group's 64-byte piece
__global__ __launch_bounds__(256) void planes_from_f32_kernel(const uint4* __restrict__ x, char* __restrict__ out, long long n_vec, int cvec) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / cvec;
        const int v = (int)(i - row * cvec);
        uint2 h, m, l;
        split_f32x4(x[i], h, m, l);
        const int C2 = cvec * 8;                                    // bytes of one plane of a row
        char* d = out + row * (3LL * C2) + (v >> 3) * 64 + (v & 3) * 16 + ((v >> 2) & 1) * 8;
        *(uint2*)d = h; *(uint2*)(d + C2) = m; *(uint2*)(d + 2 * C2) = l;
    }
}
// planes rows -> fp32 rows (exact: x = h + m + l); tests and consumers that want the plain tensor
__global__ __launch_bounds__(256) void planes_to_f32_kernel(const char* __restrict__ in, uint4* __restrict__ x, long long n_vec, int cvec) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / cvec;
        const int v = (int)(i - row * cvec);
        const int C2 = cvec * 8;
        const char* s = in + row * (3LL * C2) + (v >> 3) * 64 + (v & 3) * 16 + ((v >> 2) & 1) * 8;
        const uint2 h = *(const uint2*)s, m = *(const uint2*)(s + C2), l = *(const uint2*)(s + 2 * C2);
        float o[4];
        const unsigned hw[2] = {h.x, h.y}, mw[2] = {m.x, m.y}, lw[2] = {l.x, l.y};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            o[2 * q] = (__uint_as_float(hw[q] << 16) + __uint_as_float(mw[q] << 16)) + __uint_as_float(lw[q] << 16);
            o[2 * q + 1] = (__uint_as_float(hw[q] & 0xffff0000u) + __uint_as_float(mw[q] & 0xffff0000u)) + __uint_as_float(lw[q] & 0xffff0000u);
        }
        x[i] = make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3]));
    }
}

extern "C" int fb_planes_from_f32(const float* x, void* planes, int64_t n_rows, int32_t C, void* stream) {
    if (!x || !planes) FB_FAIL(FB_ERR_ARG, "fb_planes_from_f32: null pointer");
    if (C % 32 != 0) FB_FAIL(FB_ERR_SHAPE, "fb_planes_from_f32: C=%d must be a multiple of 32", C);
    const long long n_vec = (long long)n_rows * (C / 4);
    const int blocks = (int)((n_vec + 255) / 256 < 16384 ? (n_vec + 255) / 256 : 16384);
    if (blocks > 0) hipLaunchKernelGGL(planes_from_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (char*)planes, n_vec, C / 4);
    FB_CHECK_LAUNCH("fb_planes_from_f32");
    return FB_OK;
}
extern "C" int fb_planes_to_f32(const void* planes, float* x, int64_t n_rows, int32_t C, void* stream) {
    if (!x || !planes) FB_FAIL(FB_ERR_ARG, "fb_planes_to_f32: null pointer");
    if (C % 32 != 0) FB_FAIL(FB_ERR_SHAPE, "fb_planes_to_f32: C=%d must be a multiple of 32", C);
    const long long n_vec = (long long)n_rows * (C / 4);
    const int blocks = (int)((n_vec + 255) / 256 < 16384 ? (n_vec + 255) / 256 : 16384);
    if (blocks > 0) hipLaunchKernelGGL(planes_to_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char*)planes, (uint4*)x, n_vec, C / 4);
    FB_CHECK_LAUNCH("fb_planes_to_f32");
    return FB_OK;
}
