// Shared device/host helpers for libfbengine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fb_engine.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

extern thread_local char fb_err_buf[512];
#define FB_FAIL(code, ...)                                   \
    do {                                                     \
        snprintf(fb_err_buf, sizeof(fb_err_buf), __VA_ARGS__); \
        return (code);                                       \
    } while (0)
#define FB_CHECK_LAUNCH(name)                                                                   \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess) FB_FAIL(FB_ERR_LAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- bf16 <-> f32 (round to nearest even, NaN preserved) ------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);      // v_cvt_pk_bf16_f32 (round to nearest even) on gfx950
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}

// Element-type traits.  A "chunk" is 16 bytes: 4 f32 or 8 bf16.
template <typename T> struct ET;
template <> struct ET<float> {
    static constexpr int EB = 4, VEC = 4;
    __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
struct bf16_tag {};
template <> struct ET<bf16_tag> {
    static constexpr int EB = 2, VEC = 8;
    __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// One 16-byte operand chunk per lane for each of A and B -> accumulate into a 16x16 fp32 fragment.
//   bf16: one v_mfma_f32_16x16x32_bf16 (lane group g = lane>>4 supplies k = 8g..8g+7)
//   f32 : four v_mfma_f32_16x16x4_f32 (MFMA e sums k = {4g+e}); exact f32 fma chains.
template <typename T> __device__ __forceinline__ f32x4_t mma_chunk(const uint4& a, const uint4& b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t mma_chunk<bf16_tag>(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mma_chunk<float>(const uint4& a, const uint4& b, f32x4_t c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
}

// ---- fp32 operands on the bf16 matrix pipe: three-way split, six products ("bf16x6") -------------------------------------
// The exact-f32 MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 rate.  An fp32 value is EXACTLY the sum of three bf16
// pieces x = h + m + l (8 significand bits each: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); both subtractions are exact in
// fp32), so a product a*b = sum of nine piece products, each of them exact in fp32.  Keeping the six largest (hh, hm, mh, hl, lh,
// mm; the dropped ml + lm + ll are below 2^-23 |ab|) and accumulating in the MFMA's fp32 accumulators gives fp32-class products at
// 6/16 of the f32-MFMA time.  The finite-difference regulariser needs exactly that: emulating operand storage in the float64 oracle
// shows that 16 significand bits (three products) put 14 % of error on the regularised gradient where fp32 puts 3.5 %
// (tools/split_precision_experiment.py), because round(w + eps v) - round(w) must resolve a perturbation of 1e-4 |w|.
// f32s_tag: fp32 STORAGE (all elementwise kernels see plain float), split arithmetic inside the convolution kernels.
struct f32s_tag {};
template <> struct ET<f32s_tag> : ET<float> {};
struct split3_t { bf16x8_t h, m, l; };
// eight fp32 values of one lane (two 16-byte chunks = its K-slots of a 32-deep step) -> three bf16x8 operands
__device__ __forceinline__ split3_t split_f32x8(const uint4& c0, const uint4& c1) {
    const unsigned w[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    unsigned hp[4], mp[4], lp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = __uint_as_float(w[2 * q]), x1 = __uint_as_float(w[2 * q + 1]);
        hp[q] = pack_bf16x2(x0, x1);
        const float r0 = x0 - __uint_as_float(hp[q] << 16), r1 = x1 - __uint_as_float(hp[q] & 0xffff0000u);
        mp[q] = pack_bf16x2(r0, r1);
        const float s0 = r0 - __uint_as_float(mp[q] << 16), s1 = r1 - __uint_as_float(mp[q] & 0xffff0000u);
        lp[q] = pack_bf16x2(s0, s1);
    }
    split3_t o;
    o.h = __builtin_bit_cast(bf16x8_t, make_uint4(hp[0], hp[1], hp[2], hp[3]));
    o.m = __builtin_bit_cast(bf16x8_t, make_uint4(mp[0], mp[1], mp[2], mp[3]));
    o.l = __builtin_bit_cast(bf16x8_t, make_uint4(lp[0], lp[1], lp[2], lp[3]));
    return o;
}
// four fp32 values (one 16-byte chunk) -> the 8-byte bf16x4 pieces of the three planes (weight-gradient kernels split once per loaded
// element when they store a tile to LDS and read the planes with the transposed bf16 fragment reads)
__device__ __forceinline__ void split_f32x4(const uint4& c, uint2& h, uint2& m, uint2& l) {
    const unsigned w[4] = {c.x, c.y, c.z, c.w};
    unsigned hp[2], mp[2], lp[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float x0 = __uint_as_float(w[2 * q]), x1 = __uint_as_float(w[2 * q + 1]);
        hp[q] = pack_bf16x2(x0, x1);
        const float r0 = x0 - __uint_as_float(hp[q] << 16), r1 = x1 - __uint_as_float(hp[q] & 0xffff0000u);
        mp[q] = pack_bf16x2(r0, r1);
        const float s0 = r0 - __uint_as_float(mp[q] << 16), s1 = r1 - __uint_as_float(mp[q] & 0xffff0000u);
        lp[q] = pack_bf16x2(s0, s1);
    }
    h = make_uint2(hp[0], hp[1]); m = make_uint2(mp[0], mp[1]); l = make_uint2(lp[0], lp[1]);
}
// the six products of two already split operands (bf16x8 per plane), straight into the accumulator
__device__ __forceinline__ f32x4_t mma_planes6(const uint4 (&a)[3], const uint4 (&b)[3], f32x4_t c) {
    const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, a[0]), am = __builtin_bit_cast(bf16x8_t, a[1]), al = __builtin_bit_cast(bf16x8_t, a[2]);
    const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, b[0]), bm = __builtin_bit_cast(bf16x8_t, b[1]), bl = __builtin_bit_cast(bf16x8_t, b[2]);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
    return c;
}

// Six bf16 MFMAs, smallest terms first, into a ZERO-started accumulator; the 32-deep partial sum is then added to the running
// accumulator with v_add_f32 (round to nearest even).  Measured on one 16x16 block with K = 4608 against float64
// (tools/mfma_accum_probe.hip): accumulating the six products straight into the running sum has the L2 error of the exact-f32 MFMA
// chain (1.1e-6 vs 1.2e-6) but twice its one-sided bias (-3.4e-7: the matrix pipe truncates small addends against a large
// accumulator), and that bias is what a BN-normalised gradient amplifies (chunk gradient 1e-3..5e-3 from float64 instead of
// 2.5e-6..2e-3); zero-started, the same products give 2.2e-7 with a bias of -2.5e-8 -- five times tighter than the exact-f32 chain,
// whose rounding happens every 4 terms instead of every 32.  All nine products instead of six change nothing (2.2e-7).
__device__ __forceinline__ f32x4_t mma_split6(const split3_t& a, const split3_t& b, f32x4_t c) {
    f32x4_t t = {0.f, 0.f, 0.f, 0.f};
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.m, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.h, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.m, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, t, 0, 0, 0);
    c += t;
    asm volatile("" : "+v"(c));             // pin the sum here: LLVM otherwise sinks the whole add chain below the K loop and spills every t
    return c;
}
// One A fragment against NJ B fragments: NJ zero-started chains, interleaved product by product (independent MFMAs back to back)
template <int NJ> __device__ __forceinline__ void mma_split6_row(const split3_t& a, const split3_t (&b)[NJ], f32x4_t (&c)[NJ]) {
    f32x4_t t[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b[j].h, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b[j].l, t[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b[j].m, t[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b[j].h, t[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b[j].m, t[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b[j].h, t[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        c[j] += t[j];
        asm volatile("" : "+v"(c[j]));      // pin the sum here: LLVM otherwise sinks the whole add chain below the K loop and spills every t
    }
}
// The same with `valu_per_mfma` independent vector instructions of the caller (issued in program order BEFORE this call: the split of the next
// operand) threaded through the 6 NJ MFMAs: the request sits between the MFMAs and the pinned sums -- an `asm volatile` ends the scheduler's
// region, so a request behind the pins would find no MFMA to interleave with
template <int NJ, int VPM> __device__ __forceinline__ void mma_split6_row_mix(const split3_t& a, const split3_t (&b)[NJ], f32x4_t (&c)[NJ]) {
    f32x4_t t[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b[j].h, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b[j].l, t[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b[j].m, t[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b[j].h, t[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b[j].m, t[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b[j].h, t[j], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 6 * NJ; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        c[j] += t[j];
        asm volatile("" : "+v"(c[j]));
    }
}
template <typename T> struct is_split { static constexpr bool value = false; };
template <> struct is_split<f32s_tag> { static constexpr bool value = true; };

// ---- fp32 operands on the fp16 matrix pipe: scaled two-way split, three products ("fp16x2") -------------------------------
// Two fp16 pieces carry 22 significand bits: x*s = h + l with h = fp16(x*s), l = fp16(x*s - h) (the subtraction is exact in fp32).  The float64
// oracle with 22-bit operands puts 3.7e-2 of error on the regularised gradient where fp32 puts 3.5e-2 and 16 bits 1.4e-1
// (tools/split_precision_experiment.py), so three products hh + hl + lh (the dropped ll is below 2^-22 |ab|) do the job of the six
// bf16 ones at half the matrix time.  fp16 has 5 exponent bits, hence the scale: s = 2^e per TENSOR, chosen from the tensor's largest
// magnitude (fb_absmax, one pass) so that it lands in [2^14, 2^15); elements below 2^-25 of the maximum lose their low piece, nothing
// overflows.  Powers of two commute with every rounding; the 32-deep partial sum is multiplied by 2^-(e_a + e_b) in the v_fma_f32 that
// adds it to the running accumulator (zero-started chains as above).
// f32h_tag: fp32 STORAGE, fp16x2 arithmetic inside the convolution kernels (selected per call by fb_conv_args.amax_src / amax_wgt).
struct f32h_tag {};
template <> struct ET<f32h_tag> : ET<float> {};
template <typename T> struct is_hsplit { static constexpr bool value = false; };
template <> struct is_hsplit<f32h_tag> { static constexpr bool value = true; };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
struct split2h_t { f16x8_t h, l; };
__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){lo, hi}, f16x2_t));
}
// 2^e with the largest magnitude m of the tensor scaled into [2^14, 2^15); e clamped so that 2^-(e_a + e_b) stays a normal float
__device__ __forceinline__ float fb_pow2_scale(float m) {
    const int ex = (int)((__float_as_uint(m) >> 23) & 0xffu);
    int e = ex == 0 ? 0 : 14 - (ex - 127);
    e = e < -100 ? -100 : (e > 60 ? 60 : e);
    return __uint_as_float((unsigned)(e + 127) << 23);
}
__device__ __forceinline__ split2h_t split_h2x8(const uint4& c0, const uint4& c1, float s) {
    const unsigned w[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    unsigned hp[4], lp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = __uint_as_float(w[2 * q]) * s, x1 = __uint_as_float(w[2 * q + 1]) * s;
        hp[q] = pack_f16x2(x0, x1);
        const f16x2_t hv = __builtin_bit_cast(f16x2_t, hp[q]);
        lp[q] = pack_f16x2(x0 - (float)hv[0], x1 - (float)hv[1]);
    }
    split2h_t o;
    o.h = __builtin_bit_cast(f16x8_t, make_uint4(hp[0], hp[1], hp[2], hp[3]));
    o.l = __builtin_bit_cast(f16x8_t, make_uint4(lp[0], lp[1], lp[2], lp[3]));
    return o;
}
// four fp32 values -> the 8-byte fp16x4 pieces of the two planes (weight-gradient kernels split when they store a tile to LDS)
__device__ __forceinline__ void split_h2x4(const uint4& c, float s, uint2& h, uint2& l) {
    const unsigned w[4] = {c.x, c.y, c.z, c.w};
    unsigned hp[2], lp[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float x0 = __uint_as_float(w[2 * q]) * s, x1 = __uint_as_float(w[2 * q + 1]) * s;
        hp[q] = pack_f16x2(x0, x1);
        const f16x2_t hv = __builtin_bit_cast(f16x2_t, hp[q]);
        lp[q] = pack_f16x2(x0 - (float)hv[0], x1 - (float)hv[1]);
    }
    h = make_uint2(hp[0], hp[1]); l = make_uint2(lp[0], lp[1]);
}
__device__ __forceinline__ f32x4_t fma4(const f32x4_t& t, float k, f32x4_t c) {
    c[0] = fmaf(t[0], k, c[0]); c[1] = fmaf(t[1], k, c[1]); c[2] = fmaf(t[2], k, c[2]); c[3] = fmaf(t[3], k, c[3]);
    return c;
}
// three fp16 MFMAs, smallest terms first, zero-started; c += 2^-(e_a + e_b) * partial
__device__ __forceinline__ f32x4_t mma_split3h(const split2h_t& a, const split2h_t& b, f32x4_t c, float inv) {
    f32x4_t t = {0.f, 0.f, 0.f, 0.f};
    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l, b.h, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b.l, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b.h, t, 0, 0, 0);
    c = fma4(t, inv, c);
    asm volatile("" : "+v"(c));
    return c;
}
template <int NJ> __device__ __forceinline__ void mma_split3h_row(const split2h_t& a, const split2h_t (&b)[NJ], f32x4_t (&c)[NJ], float inv) {
    f32x4_t t[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l, b[j].h, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b[j].l, t[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) t[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b[j].h, t[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        c[j] = fma4(t[j], inv, c[j]);
        asm volatile("" : "+v"(c[j]));
    }
}
// the three products of two already split operands, straight into the (scaled) accumulator: weight-gradient kernels unscale once at the end
__device__ __forceinline__ f32x4_t mma_planes3h_acc(const uint4 (&a)[2], const uint4 (&b)[2], f32x4_t c) {
    const f16x8_t ah = __builtin_bit_cast(f16x8_t, a[0]), al = __builtin_bit_cast(f16x8_t, a[1]);
    const f16x8_t bh = __builtin_bit_cast(f16x8_t, b[0]), bl = __builtin_bit_cast(f16x8_t, b[1]);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
    return c;
}
// the three products of two already split operands (fp16x8 per plane), zero-started; c += inv * partial
__device__ __forceinline__ f32x4_t mma_planes3h(const uint4 (&a)[2], const uint4 (&b)[2], f32x4_t c, float inv) {
    const f16x8_t ah = __builtin_bit_cast(f16x8_t, a[0]), al = __builtin_bit_cast(f16x8_t, a[1]);
    const f16x8_t bh = __builtin_bit_cast(f16x8_t, b[0]), bl = __builtin_bit_cast(f16x8_t, b[1]);
    f32x4_t t = {0.f, 0.f, 0.f, 0.f};
    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, t, 0, 0, 0);
    return fma4(t, inv, c);
}
// FB_F32_EXACT=1 (environment, read once): fp32 convolutions on the exact-f32 MFMA instead of the split path (A/B and reference)
bool fb_f32_split_enabled();
// the value of an environment switch that selects an EXPERIMENTAL kernel form (one that lost its A/B): nullptr unless FB_EXPERIMENTAL=1 is set as well (runtime.cpp)
bool fb_experimental();
const char* fb_getenv_experimental(const char* name);
// CUs the persistent kernels size their grids for (runtime.cpp: the device's CU count minus FB_CU_RESERVE)
int fb_persistent_cus();

// 128-bit buffer stores with an SGPR offset: keep the four data registers alive for a few wait states after the store.
// The register allocator likes to reuse a data register in the instruction right behind such a store (seen: v_cndmask writing the next
// halo request's offset into dword 1 of the vector just stored).  LLVM adds no wait state there -- its hazard table exempts MUBUF stores
// with a register soffset, as the ISA manual does -- yet on gfx950, whenever a second stream kept the memory pipeline busy, the NEW value
// now and then reached memory in place of the stored dword: one bf16 of ~1e38 (the low half of a byte offset) in a 50 MB gradient every
// few launches, then Inf / NaN down the backward chain.  tools/race_probe.py found it (50 % of 3-step runs differed from the one-stream
// trace; 0 of 192 with the guard); profiles/r3_notes.md has the hunt.
typedef __attribute__((ext_vector_type(4))) unsigned fb_store_u32x4;
#ifndef FB_STORE_GUARD_NOPS
#define FB_STORE_GUARD_NOPS 3          /* s_nop operand: N + 1 wait states (A/B builds: tools/build_variant.py) */
#endif
__device__ __forceinline__ void store_b128_guard(fb_store_u32x4 data) {
#if FB_STORE_GUARD_NOPS >= 0
#ifdef FB_STORE_GUARD_NOMEM
    asm volatile("s_nop %1" ::"v"(data), "n"(FB_STORE_GUARD_NOPS));
#else
    asm volatile("s_nop %1" ::"v"(data), "n"(FB_STORE_GUARD_NOPS) : "memory");
#endif
#endif
}

// dx of a BatchNorm backward, dx = c_dy * dy + c_x * x + c_0 (coefficients from fb_bn_bwd_finalize): ONE spelling with explicit fused multiply-adds,
// shared by every kernel that evaluates it (fb_bn_bwd_apply / _apply2, the stem weight gradient's loader, fb_bn_bwd_fused) -- left to the
// compiler's contraction the same source line rounded differently in two instantiations of one template (fp32, last bit)
__device__ __forceinline__ float fb_bn_dx(float c_dy, float c_x, float c_0, float dy, float x) { return fmaf(c_dy, dy, fmaf(c_x, x, c_0)); }

// Sum over the 16 lanes of a DPP row (lanes 16k..16k+15), result in every lane: four v_add_f32 with row_ror modifiers --
// no LDS traffic (``__shfl_xor`` lowers to ds_bpermute_b32, which queues behind the fragment reads of the co-resident workgroup).
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}

// deterministic block reduction of `NV` values per thread (sum), result valid on thread 0; smem >= NV*nwaves floats
template <int NV> __device__ __forceinline__ void block_reduce_sum(float (&v)[NV], float* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_xor(v[i], off);
    }
    if (lane == 0)
        for (int i = 0; i < NV; ++i) smem[i * nw + wave] = v[i];
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 0; i < NV; ++i) {
            float s = 0.f;
            for (int w = 0; w < nw; ++w) s += smem[i * nw + w];
            v[i] = s;
        }
    __syncthreads();
}
