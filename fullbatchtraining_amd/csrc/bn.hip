// Training-mode BatchNorm pieces around the conv kernels (HBM-bound, NHWC, 16-byte vector lanes over channels).
//   fwd : conv epilogue partial sums -> fb_bn_fwd_finalize -> fb_bn_apply (normalise + affine + residual + ReLU)
//   bwd : fb_bn_bwd_reduce (sum dy, sum dy*xhat with the ReLU mask fused) -> fb_bn_bwd_finalize -> fb_bn_bwd_apply
// All reductions are two-stage with a fixed order (no atomics): results are bit-reproducible run to run, which the
// finite-difference regulariser relies on (both passes see identically ordered sums).
#include "common.h"
#include "profile.h"

// Two-stage fixed-order sum of the per-pixel-block partials of one (group, channel): blockDim.x threads = 16 channels x SEGS segments
// (SEGS = blockDim.x / 16: 16 or 64); segment s adds blocks s, s+SEGS, ... in double, then the SEGS segment sums are added in order.
// Valid on threads with seg == 0.  64 segments for long columns: with 16 a thread walked 64+ dependent loads (13-37 us per launch).
__device__ __forceinline__ bool partial_sum2(const float* __restrict__ part, int n_mblocks, int blocks_per_group, int C, int g,
                                             int& c, double& s, double& q) {
    __shared__ double red[2][64][16];
    const int segs = (int)blockDim.x >> 4;
    const int cl = threadIdx.x & 15, seg = threadIdx.x >> 4;
    c = blockIdx.x * 16 + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
        const float* ps = part + ((long long)g * blocks_per_group) * C + c;
        const float* pq = part + ((long long)n_mblocks + (long long)g * blocks_per_group) * C + c;
        for (int k = seg; k < blocks_per_group; k += segs) { a += (double)ps[(long long)k * C]; b += (double)pq[(long long)k * C]; }
    }
    red[0][seg][cl] = a; red[1][seg][cl] = b;
    __syncthreads();
    if (seg != 0 || c >= C) return false;
    s = 0.0; q = 0.0;
    for (int k = 0; k < segs; ++k) { s += red[0][k][cl]; q += red[1][k][cl]; }
    return true;
}
static inline int bn_finalize_threads(int blocks_per_group) { return blocks_per_group >= 256 ? 1024 : 256; }

// ---------------------------------------------------------------------------------------------------------------------
__global__ void bn_fwd_finalize_kernel(const float* __restrict__ part, int n_mblocks, int blocks_per_group, int C, double inv_count,
                                       const float* __restrict__ gamma, const float* __restrict__ beta, long long pstride, float eps,
                                       float* __restrict__ mean_tab, float* __restrict__ var_tab, int ch_total, int ch_off,
                                       float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ invstd_out) {
    const int g = blockIdx.y;
    int c; double s, q;
    if (!partial_sum2(part, n_mblocks, blocks_per_group, C, g, c, s, q)) return;
    const double mean = s * inv_count;
    double var = q * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float gm = gamma[(long long)g * pstride + c], bt = beta[(long long)g * pstride + c];
    const float sc = gm * invstd;
    mean_tab[(long long)g * ch_total + ch_off + c] = (float)mean;
    var_tab[(long long)g * ch_total + ch_off + c] = (float)var;
    scale[(long long)g * C + c] = sc;
    shift[(long long)g * C + c] = bt - (float)mean * sc;
    invstd_out[(long long)g * C + c] = invstd;
}

extern "C" int fb_bn_fwd_finalize(const float* stat_partial, int32_t n_mblocks, int32_t n_groups, int32_t C, double count,
                                  const float* gamma, const float* beta, int64_t param_group_stride, float eps, float* mean_tab,
                                  float* var_tab, int32_t ch_total, int32_t ch_off, float* scale, float* shift, float* invstd,
                                  void* stream) {
    if (!stat_partial || !gamma || !beta || !mean_tab || !var_tab || !scale || !shift || !invstd) FB_FAIL(FB_ERR_ARG, "fb_bn_fwd_finalize: null pointer");
    if (n_mblocks % n_groups != 0) FB_FAIL(FB_ERR_SHAPE, "fb_bn_fwd_finalize: %d pixel blocks not divisible by %d groups", n_mblocks, n_groups);
    dim3 grid((C + 15) / 16, n_groups);
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, grid, dim3(bn_finalize_threads(n_mblocks / n_groups)), 0, (hipStream_t)stream, stat_partial, n_mblocks, n_mblocks / n_groups, C,
                       1.0 / count, gamma, beta, (long long)param_group_stride, eps, mean_tab, var_tab, ch_total, ch_off, scale, shift, invstd);
    FB_CHECK_LAUNCH("fb_bn_fwd_finalize");
    return FB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int RES>   // RES: 0 none, 1 plain residual, 2 residual with its own BN affine
__global__ void bn_apply_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, const float* __restrict__ scale,
                                const float* __restrict__ shift, const uint4* __restrict__ res, const float* __restrict__ rscale,
                                const float* __restrict__ rshift, long long n_vec, int cvec, long long vec_per_group, int C, int relu,
                                unsigned char* __restrict__ mask_out, long long valid_vec) {
    constexpr int V = ET<T>::VEC;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (long long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cvec) * V;
        const long long g = i / vec_per_group;
        if (i - g * vec_per_group >= valid_vec) {           // padding pixels of a ragged statistics group: exact zeros, mask clear
            y[i] = make_uint4(0, 0, 0, 0);
            if (mask_out) mask_out[i] = 0;
            continue;
        }
        float xv[V], o[V];
        unsigned m = 0;
        ET<T>::unpack(x[i], xv);
        const float* sc = scale + g * C + c0; const float* sh = shift + g * C + c0;
        float rv[V];
        if (RES) ET<T>::unpack(res[i], rv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float v = xv[k] * sc[k] + sh[k];
            if (RES == 1) v += rv[k];
            if (RES == 2) v += rv[k] * rscale[g * C + c0 + k] + rshift[g * C + c0 + k];
            o[k] = relu ? fmaxf(v, 0.f) : v;
            m |= (v > 0.f ? 1u : 0u) << k;
        }
        y[i] = ET<T>::pack(o);
        if (mask_out) mask_out[i] = (unsigned char)m;     // ReLU mask, one byte per 16-byte vector (backward reads 1/16 of y)
    }
}

// "Span" variants of the elementwise kernels (used whenever the channel vectors of a pixel divide 256): a workgroup owns a run of
// consecutive 16-byte vectors of ONE statistics group, so a thread keeps its channel vector for the whole run and its
// coefficients stay in registers -- the grid-stride form above re-reads 16-48 coefficient dwords per 16-byte vector through
// the same texture path as the data (1.5-3x its bytes).  BN_SPAN_U vectors per thread are in flight per loop trip.
template <int V> __device__ __forceinline__ void load_coef(const float* __restrict__ p, float* o) {
#pragma unroll
    for (int k = 0; k < V; k += 4) { const float4 v = *(const float4*)(p + k); o[k] = v.x; o[k + 1] = v.y; o[k + 2] = v.z; o[k + 3] = v.w; }
}
constexpr int BN_SPAN_U = 2;        // vectors in flight per thread and loop trip
// streaming accesses of the span kernels: every tensor is far larger than L2 + Infinity Cache and is touched once per kernel
typedef __attribute__((ext_vector_type(4))) unsigned bn_u32x4_t;
__device__ __forceinline__ uint4 ld_stream(const uint4* p) {
#ifdef FB_NO_NT
    return *p;
#else
    const bn_u32x4_t v = __builtin_nontemporal_load((const bn_u32x4_t*)p);
    return make_uint4(v[0], v[1], v[2], v[3]);
#endif
}
// loads of the backward REDUCE pass: the apply pass re-reads the same (dout, x) rows right afterwards; -DFB_BN_REDUCE_PLAIN keeps them
// cacheable (tools/bn_mall_experiment.py: sub-batched reduce -> apply out of the Infinity Cache)
__device__ __forceinline__ uint4 ld_reduce(const uint4* p) {
#ifdef FB_BN_REDUCE_PLAIN
    return *p;
#else
    return ld_stream(p);
#endif
}
__device__ __forceinline__ void st_stream(uint4* p, const uint4& v) {
#ifdef FB_NO_NT
    *p = v;
#else
    __builtin_nontemporal_store((bn_u32x4_t){v.x, v.y, v.z, v.w}, (bn_u32x4_t*)p);
#endif
}

// Largest magnitude of what a streaming kernel wrote (scale source of the fp16x2 convolutions, fb_absmax semantics; the caller zeroes the
// slot): wave maximum, then one atomic per wave ONLY if it would raise the value -- after the first few workgroups almost none does
// Two levels, no atomics: every workgroup stores the maximum of its run to ws[group][run] (an atomic on one address per workgroup -- or
// even an agent-scope load of it -- serialises on this multi-die part: 2.3x the kernel time measured), amax_finish_kernel folds the
// runs of a group.
__device__ __forceinline__ void amax_commit(float* ws, float am) {
    __shared__ float red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) ws[(long long)blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(256) void amax_finish_kernel(const float* __restrict__ ws, int runs, float* __restrict__ out) {
    float m = 0.f;
    for (int i = threadIdx.x; i < runs; i += 256) m = fmaxf(m, ws[(long long)blockIdx.x * runs + i]);
    __shared__ float red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// floats of scratch fb_bn_apply / fb_bn_bwd_apply need for amax_out
extern "C" int64_t fb_ws_bn_amax_floats(int64_t n_pixels, int32_t C, int64_t pixels_per_group) {
    const long long vpg = pixels_per_group * (C / 4);
    const long long groups = (n_pixels + pixels_per_group - 1) / pixels_per_group;
    return ((vpg + 511) / 512) * groups;
}

template <typename T, int RES>
__global__ __launch_bounds__(256) void bn_apply_span_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const uint4* __restrict__ res,
                                                            const float* __restrict__ rscale, const float* __restrict__ rshift, long long n_vec,
                                                            int cvec, long long vec_per_group, int C, int relu, unsigned char* __restrict__ mask_out,
                                                            int span, long long valid_vec, uint4* __restrict__ pool_out, float* __restrict__ amax) {
    constexpr int V = ET<T>::VEC;
    float am = 0.f;
    const long long g = blockIdx.y, base = g * vec_per_group;
    const long long lim = n_vec - base < vec_per_group ? n_vec - base : vec_per_group;
    const long long lo = (long long)blockIdx.x * span;
    const long long hi = lo + span < lim ? lo + span : lim;
    const int c0 = (int)(threadIdx.x % cvec) * V;
    float sc[V], sh[V], rsc[RES == 2 ? V : 1], rsh[RES == 2 ? V : 1];
    load_coef<V>(scale + g * C + c0, sc); load_coef<V>(shift + g * C + c0, sh);
    if constexpr (RES == 2) { load_coef<V>(rscale + g * C + c0, rsc); load_coef<V>(rshift + g * C + c0, rsh); }
    x += base; y += base;
    if (RES) res += base;
    if (mask_out) mask_out += base;
    auto one = [&](const uint4& xr, const uint4& rr, long long i) {
        float xv[V], rv[V], o[V];
        unsigned m = 0;
        ET<T>::unpack(xr, xv);
        if (RES) ET<T>::unpack(rr, rv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float v = xv[k] * sc[k] + sh[k];
            if (RES == 1) v += rv[k];
            if (RES == 2) v += rv[k] * rsc[k] + rsh[k];
            o[k] = relu ? fmaxf(v, 0.f) : v;
            m |= (v > 0.f ? 1u : 0u) << k;
        }
        uint4 packed = ET<T>::pack(o);
        if (i >= valid_vec) { packed = make_uint4(0, 0, 0, 0); m = 0; }      // padding pixels of a ragged statistics group
        else if (amax) {
#pragma unroll
            for (int k = 0; k < V; ++k) am = fmaxf(am, fabsf(o[k]));
        }
        st_stream(y + i, packed);
        if (mask_out) mask_out[i] = (unsigned char)m;
        return packed;
    };
    long long i = lo + threadIdx.x;
    for (; i + (BN_SPAN_U - 1) * 256 < hi; i += BN_SPAN_U * 256) {
        uint4 xr[BN_SPAN_U], rr[BN_SPAN_U], outp[BN_SPAN_U];
#pragma unroll
        for (int u = 0; u < BN_SPAN_U; ++u) { xr[u] = ld_stream(x + i + u * 256); if (RES) rr[u] = ld_stream(res + i + u * 256); }
#pragma unroll
        for (int u = 0; u < BN_SPAN_U; ++u) outp[u] = one(xr[u], rr[u], i + u * 256);
        if (pool_out) {
            // AvgPool2d(2,2) of the block output for the next block's shortcut (reference resnets.py:149), fused: the host passes
            // pool_out only when a run is exactly two image rows (W * cvec == 256, span == 512), so the two vectors of a thread are
            // vertical neighbours and the horizontal neighbour sits cvec lanes away.  Pooled from the STORED (rounded) values in the
            // order of avgpool2_fwd_kernel: bit-identical to the separate kernel, one full read of the activation less.
            static_assert(BN_SPAN_U == 2, "the fused pooling pairs the two vectors of a thread");
            float r0[V], r1[V], acc[V];
            ET<T>::unpack(outp[0], r0); ET<T>::unpack(outp[1], r1);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float n0 = __shfl_xor(r0[k], cvec), n1 = __shfl_xor(r1[k], cvec);
                acc[k] = (((r0[k] + n0) + r1[k]) + n1) * 0.25f;
            }
            const int px = (int)threadIdx.x / cvec;
            if ((px & 1) == 0) pool_out[((base + lo) >> 9) * 128 + (px >> 1) * cvec + (threadIdx.x % cvec)] = ET<T>::pack(acc);
        }
    }
    for (; i < hi; i += 256) { uint4 rr = make_uint4(0, 0, 0, 0); if (RES) rr = res[i]; one(x[i], rr, i); }
    if (amax) amax_commit(amax, am);                          // (uniform branch) one value per run; folded per statistics group afterwards
}

// run length of a span kernel: 512 vectors (one trip of two vectors per thread).  Alone on the device longer runs are a little faster
// (4096 vectors, four in flight: 5.7-6.5 TB/s), inside the step short ones win: 245.5 vs 249.6 ms/step (same-box A/B of 512 / 1024 /
// 2048 / 4096 / 8192-vector runs with 1, 2, 4, 8 vectors in flight; profiles/r1_pmc_notes.md)
static inline int bn_span(long long) { return 512; }

template <typename T>
static bool launch_bn_apply(const void* x, void* y, const float* scale, const float* shift, const void* res, const float* rscale,
                            const float* rshift, int64_t n_pixels, int C, int64_t ppg, int64_t valid_ppg, int relu, unsigned char* mask_out, uint4* pool_out,
                            float* amax_out, float* amax_ws, hipStream_t st) {
    const int cvec = C / ET<T>::VEC;
    const long long n_vec = n_pixels * cvec, vpg = ppg * cvec;
    const long long valid_vec = (valid_ppg > 0 && valid_ppg < ppg ? valid_ppg : ppg) * cvec;
    if (256 % cvec == 0 && vpg > 0) {
        const int span = bn_span(n_vec);
        const dim3 grid((unsigned)((vpg + span - 1) / span), (unsigned)((n_vec + vpg - 1) / vpg));
        if (!res) hipLaunchKernelGGL((bn_apply_span_kernel<T, 0>), grid, dim3(256), 0, st, (const uint4*)x, (uint4*)y, scale, shift, nullptr, nullptr, nullptr, n_vec, cvec, vpg, C, relu, mask_out, span, valid_vec, pool_out, amax_out ? amax_ws : nullptr);
        else if (!rscale) hipLaunchKernelGGL((bn_apply_span_kernel<T, 1>), grid, dim3(256), 0, st, (const uint4*)x, (uint4*)y, scale, shift, (const uint4*)res, nullptr, nullptr, n_vec, cvec, vpg, C, relu, mask_out, span, valid_vec, pool_out, amax_out ? amax_ws : nullptr);
        else hipLaunchKernelGGL((bn_apply_span_kernel<T, 2>), grid, dim3(256), 0, st, (const uint4*)x, (uint4*)y, scale, shift, (const uint4*)res, rscale, rshift, n_vec, cvec, vpg, C, relu, mask_out, span, valid_vec, pool_out, amax_out ? amax_ws : nullptr);
        if (amax_out) hipLaunchKernelGGL(amax_finish_kernel, dim3(grid.y), dim3(256), 0, st, amax_ws, (int)grid.x, amax_out);
        return true;
    }
    const int blocks = (int)((n_vec + 255) / 256 < 8192 ? (n_vec + 255) / 256 : 8192);
    if (!res) hipLaunchKernelGGL((bn_apply_kernel<T, 0>), dim3(blocks), dim3(256), 0, st, (const uint4*)x, (uint4*)y, scale, shift, nullptr, nullptr, nullptr, n_vec, cvec, vpg, C, relu, mask_out, valid_vec);
    else if (!rscale) hipLaunchKernelGGL((bn_apply_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, (const uint4*)x, (uint4*)y, scale, shift, (const uint4*)res, nullptr, nullptr, n_vec, cvec, vpg, C, relu, mask_out, valid_vec);
    else hipLaunchKernelGGL((bn_apply_kernel<T, 2>), dim3(blocks), dim3(256), 0, st, (const uint4*)x, (uint4*)y, scale, shift, (const uint4*)res, rscale, rshift, n_vec, cvec, vpg, C, relu, mask_out, valid_vec);
    return false;                                           // (the grid-stride form does not track the largest magnitude)
}

// fused AvgPool2d(2,2) output of fb_bn_apply: a 512-vector run of the span kernel must be exactly two image rows
extern "C" int32_t fb_bn_apply_can_pool(int32_t C, int32_t W, int64_t pixels_per_group, int32_t dtype) {
    return dtype == FB_BF16 && W > 0 && (W & 1) == 0 && (long long)W * C == 2048 && C % 8 == 0 && 256 % (C / 8) == 0 && pixels_per_group % (2 * W) == 0;
}

extern "C" int fb_bn_apply(const void* x, void* y, const float* scale, const float* shift, const void* res, const float* rscale,
                           const float* rshift, int64_t n_pixels, int32_t C, int64_t pixels_per_group, int64_t valid_pixels_per_group,
                           int32_t relu, void* mask_out, void* pool_out, int32_t pool_W, int32_t dtype, float* amax_out, float* amax_ws, void* stream) {
    if (!x || !y || !scale || !shift) FB_FAIL(FB_ERR_ARG, "fb_bn_apply: null pointer");
    if (amax_out && (dtype != FB_F32 || !amax_ws)) FB_FAIL(FB_ERR_ARG, "fb_bn_apply: amax_out is for fp32 tensors and needs amax_ws (fb_ws_bn_amax_floats)");
    const int64_t n_groups = (n_pixels + pixels_per_group - 1) / pixels_per_group;
    if (C % 8 != 0) FB_FAIL(FB_ERR_SHAPE, "fb_bn_apply: C=%d must be a multiple of 8", C);
    if (pool_out && !fb_bn_apply_can_pool(C, pool_W, pixels_per_group, dtype))
        FB_FAIL(FB_ERR_UNSUPPORTED, "fb_bn_apply: fused 2x2 average pooling needs bf16, W * C == 2048 and whole row pairs per group (C=%d W=%d)", C, pool_W);
    const int32_t info[FB_PROF_INFO] = {(int32_t)(n_pixels / 128), C, (int32_t)(pixels_per_group / 128), dtype, res ? 1 : 0, mask_out ? 1 : 0, 0, pool_out ? 1 : 0, 0, 0, 0};
    const int prof = fb_prof_begin(FB_PROF_BN_APPLY, (hipStream_t)stream, info);
    bool tracked = false;
    if (dtype == FB_F32) tracked = launch_bn_apply<float>(x, y, scale, shift, res, rscale, rshift, n_pixels, C, pixels_per_group, valid_pixels_per_group, relu, (unsigned char*)mask_out, (uint4*)pool_out, amax_out, amax_ws, (hipStream_t)stream);
    else launch_bn_apply<bf16_tag>(x, y, scale, shift, res, rscale, rshift, n_pixels, C, pixels_per_group, valid_pixels_per_group, relu, (unsigned char*)mask_out, (uint4*)pool_out, nullptr, nullptr, (hipStream_t)stream);
    fb_prof_end(prof, (hipStream_t)stream);
    FB_CHECK_LAUNCH("fb_bn_apply");
    if (amax_out && !tracked) return fb_absmax((const float*)y, pixels_per_group * C, (int32_t)n_groups, pixels_per_group * C, 1, amax_out, stream);
    return FB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void bn_running_update_kernel(float* __restrict__ rm, float* __restrict__ rv, const float* __restrict__ mean_tab,
                                         const float* __restrict__ var_tab, int n_passes, long long pass_stride,
                                         const float* __restrict__ unbias, int n_groups, int ch_total, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ch_total) return;
    float m = rm[c], v = rv[c];
    const float ub = unbias[c], keep = 1.f - momentum;
    for (int g = 0; g < n_groups; ++g)
        for (int p = 0; p < n_passes; ++p) {
            const long long o = p * pass_stride + (long long)g * ch_total + c;
            m = keep * m + momentum * mean_tab[o];
            v = keep * v + momentum * (var_tab[o] * ub);
        }
    rm[c] = m; rv[c] = v;
}

extern "C" int fb_bn_running_update(float* running_mean, float* running_var, const float* mean_tab, const float* var_tab, int32_t n_passes,
                                    int64_t pass_stride, const float* unbias, int32_t n_groups, int32_t ch_total, float momentum,
                                    void* stream) {
    if (!running_mean || !running_var || !mean_tab || !var_tab || !unbias) FB_FAIL(FB_ERR_ARG, "fb_bn_running_update: null pointer");
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((ch_total + 255) / 256), dim3(256), 0, (hipStream_t)stream, running_mean,
                       running_var, mean_tab, var_tab, n_passes, (long long)pass_stride, unbias, n_groups, ch_total, momentum);
    FB_CHECK_LAUNCH("fb_bn_running_update");
    return FB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// backward reduce: workgroup = PB pixels (128 .. 1024, one statistics group) x all channels; thread -> (pixel sub-row, channel vector).
// With 256 % cvec == 0 the vectors of the workgroup's pixels are one linear run and a thread keeps its channel vector: mean / invstd
// live in registers, four (dy, x, mask) triples are in flight per trip.  One partial row per workgroup, summed in fixed order by
// fb_bn_bwd_finalize.  fb_bn_bwd_reduce_rows() tells the caller how many rows that is.
static inline int bn_reduce_pixels(long long n_pixels, long long ppg) {
    int pb = 128;
    while (pb < 1024 && ppg % (2 * pb) == 0 && n_pixels / (2 * pb) >= 2048) pb *= 2;
    return pb;
}
extern "C" int32_t fb_bn_bwd_reduce_rows(int64_t n_pixels, int64_t pixels_per_group) {
    if (n_pixels <= 0 || pixels_per_group <= 0) return 0;
    const int pb = bn_reduce_pixels(n_pixels, pixels_per_group);
    return (int32_t)((n_pixels + pb - 1) / pb);
}

// DUAL: TWO BatchNorms share the incoming gradient and its ReLU mask (a downsampling block: the BatchNorm of conv2 and the one of the shortcut
// convolution both take the gradient of the block output): one pass over dout reduces for both -- sum dy is common, sum dy * xhat per BatchNorm.
// Same thread / pixel order as the single form: each BatchNorm's sums are bit-identical to its own fb_bn_bwd_reduce.
struct BnReduceB { const uint4* x; const float* invstd; float* partial; int ch_off; };
template <typename T, bool DUAL = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const uint4* __restrict__ dout, const uint4* __restrict__ y,
                                                            const unsigned char* __restrict__ mask,
                                                            const uint4* __restrict__ x, const float* __restrict__ mean_tab,
                                                            const float* __restrict__ invstd, int ch_total, int ch_off,
                                                            float* __restrict__ partial, long long n_pixels, int C, long long ppg,
                                                            int n_mblocks, int PB, const BnReduceB B) {
    constexpr int V = ET<T>::VEC;
    extern __shared__ float sm[];   // [rows][C][2]
    const int cvec = C / V;
    const long long p0 = (long long)blockIdx.x * PB;
    const long long g = p0 / ppg;
    const long long p1 = p0 + PB < n_pixels ? p0 + PB : n_pixels;
    auto mask_at = [&](long long i) -> unsigned {
        if (mask) return mask[i];
        if (!y) return 0xffu;
        float yv[V];
        ET<T>::unpack(y[i], yv);
        unsigned mk = 0;
#pragma unroll
        for (int k = 0; k < V; ++k) mk |= (yv[k] > 0.f ? 1u : 0u) << k;
        return mk;
    };
    int rows;
    if (256 % cvec == 0) {
        rows = 256 / cvec;
        const int cv = threadIdx.x % cvec, row = threadIdx.x / cvec, c0 = cv * V;
        float mu[V], is[V], s1[V], s2[V];
        float muB[DUAL ? V : 1], isB[DUAL ? V : 1], s2B[DUAL ? V : 1];
#pragma unroll
        for (int k = 0; k < V; ++k) { mu[k] = mean_tab[g * ch_total + ch_off + c0 + k]; is[k] = invstd[g * C + c0 + k]; s1[k] = 0.f; s2[k] = 0.f; }
        if constexpr (DUAL) {
#pragma unroll
            for (int k = 0; k < V; ++k) { muB[k] = mean_tab[g * ch_total + B.ch_off + c0 + k]; isB[k] = B.invstd[g * C + c0 + k]; s2B[k] = 0.f; }
        }
        auto one = [&](const uint4& dr, const uint4& xr, const uint4& xbr, unsigned mk) {
            float d[V], xv[V];
            ET<T>::unpack(dr, d); ET<T>::unpack(xr, xv);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float dy = ((mk >> k) & 1u) ? d[k] : 0.f;
                s1[k] += dy; s2[k] += dy * ((xv[k] - mu[k]) * is[k]);
            }
            if constexpr (DUAL) {
                float xb[V];
                ET<T>::unpack(xbr, xb);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float dy = ((mk >> k) & 1u) ? d[k] : 0.f;
                    s2B[k] += dy * ((xb[k] - muB[k]) * isB[k]);
                }
            }
        };
        const long long hi = p1 * cvec;
        long long i = p0 * cvec + threadIdx.x;
        for (; i + (BN_SPAN_U - 1) * 256 < hi; i += BN_SPAN_U * 256) {
            uint4 dr[BN_SPAN_U], xr[BN_SPAN_U], xbr[BN_SPAN_U];
            unsigned mk[BN_SPAN_U];
#pragma unroll
            for (int u = 0; u < BN_SPAN_U; ++u) {
                dr[u] = ld_reduce(dout + i + u * 256); xr[u] = ld_reduce(x + i + u * 256); mk[u] = mask_at(i + u * 256);
                if constexpr (DUAL) xbr[u] = ld_reduce(B.x + i + u * 256); else xbr[u] = xr[u];
            }
#pragma unroll
            for (int u = 0; u < BN_SPAN_U; ++u) one(dr[u], xr[u], xbr[u], mk[u]);
        }
        for (; i < hi; i += 256) { if constexpr (DUAL) one(dout[i], x[i], B.x[i], mask_at(i)); else one(dout[i], x[i], x[i], mask_at(i)); }
#pragma unroll
        for (int k = 0; k < V; ++k) { sm[(row * C + c0 + k) * 2] = s1[k]; sm[(row * C + c0 + k) * 2 + 1] = s2[k]; }
        if constexpr (DUAL) {
#pragma unroll
            for (int k = 0; k < V; ++k) sm[2 * rows * C + row * C + c0 + k] = s2B[k];          // [rows][C] behind the [rows][C][2] table
        }
    } else {                        // channel counts whose vectors do not divide the workgroup: one sub-row, channel loop
        rows = 256 / cvec > 0 ? 256 / cvec : 1;
        const int lanes = cvec < 256 ? cvec : 256;
        for (int cv = threadIdx.x % lanes; cv < cvec; cv += lanes) {
            const int row = threadIdx.x / lanes;
            const int c0 = cv * V;
            float mu[V], is[V], s1[V], s2[V];
#pragma unroll
            for (int k = 0; k < V; ++k) { mu[k] = mean_tab[g * ch_total + ch_off + c0 + k]; is[k] = invstd[g * C + c0 + k]; s1[k] = 0.f; s2[k] = 0.f; }
            if (row < rows) {
                for (long long pidx = p0 + row; pidx < p1; pidx += rows) {
                    const long long i = pidx * cvec + cv;
                    float d[V], xv[V];
                    ET<T>::unpack(dout[i], d); ET<T>::unpack(x[i], xv);
                    const unsigned mk = mask_at(i);
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        const float dy = ((mk >> k) & 1u) ? d[k] : 0.f;
                        s1[k] += dy; s2[k] += dy * ((xv[k] - mu[k]) * is[k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < V; ++k) { sm[(row * C + c0 + k) * 2] = s1[k]; sm[(row * C + c0 + k) * 2 + 1] = s2[k]; }
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < rows; ++r) { a += sm[(r * C + c) * 2]; b += sm[(r * C + c) * 2 + 1]; }
        partial[(long long)blockIdx.x * C + c] = a;
        partial[((long long)n_mblocks + blockIdx.x) * C + c] = b;
        if constexpr (DUAL) {
            float bb = 0.f;
            for (int r = 0; r < rows; ++r) bb += sm[2 * rows * C + r * C + c];
            B.partial[(long long)blockIdx.x * C + c] = a;
            B.partial[((long long)n_mblocks + blockIdx.x) * C + c] = bb;
        }
    }
}

extern "C" int fb_bn_bwd_reduce(const void* dout, const void* y, const void* mask, const void* x, const float* mean_tab, const float* invstd,
                                int32_t ch_total, int32_t ch_off, float* partial, int64_t n_pixels, int32_t C,
                                int64_t pixels_per_group, int32_t dtype, void* stream) {
    if (!dout || !x || !mean_tab || !invstd || !partial) FB_FAIL(FB_ERR_ARG, "fb_bn_bwd_reduce: null pointer");
    if (pixels_per_group % 128 != 0) FB_FAIL(FB_ERR_SHAPE, "fb_bn_bwd_reduce: pixels_per_group=%lld must be a multiple of 128", (long long)pixels_per_group);
    const int V = dtype == FB_F32 ? 4 : 8;
    if (C % V != 0) FB_FAIL(FB_ERR_SHAPE, "fb_bn_bwd_reduce: C=%d", C);
    const int cvec = C / V, rows = 256 / cvec > 0 ? 256 / cvec : 1;
    const int PB = bn_reduce_pixels(n_pixels, pixels_per_group);
    const int n_mblocks = fb_bn_bwd_reduce_rows(n_pixels, pixels_per_group);
    const size_t smem = (size_t)rows * C * 2 * sizeof(float);
    if (smem > 64 * 1024) FB_FAIL(FB_ERR_UNSUPPORTED, "fb_bn_bwd_reduce: C=%d too large", C);
    const int32_t info[FB_PROF_INFO] = {(int32_t)(n_pixels / 128), C, (int32_t)(pixels_per_group / 128), dtype, 0, (mask || y) ? 1 : 0, 0, 0, 0, 0, 0};
    const int prof = fb_prof_begin(FB_PROF_BN_BWD_REDUCE, (hipStream_t)stream, info);
    if (dtype == FB_F32)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<float>), dim3(n_mblocks), dim3(256), smem, (hipStream_t)stream, (const uint4*)dout, (const uint4*)y,
                           (const unsigned char*)mask, (const uint4*)x, mean_tab, invstd, ch_total, ch_off, partial, (long long)n_pixels, C, (long long)pixels_per_group, n_mblocks, PB, BnReduceB{});
    else
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_tag>), dim3(n_mblocks), dim3(256), smem, (hipStream_t)stream, (const uint4*)dout, (const uint4*)y,
                           (const unsigned char*)mask, (const uint4*)x, mean_tab, invstd, ch_total, ch_off, partial, (long long)n_pixels, C, (long long)pixels_per_group, n_mblocks, PB, BnReduceB{});
    fb_prof_end(prof, (hipStream_t)stream);
    FB_CHECK_LAUNCH("fb_bn_bwd_reduce");
    return FB_OK;
}

extern "C" int fb_bn_bwd_reduce2(const void* dout, const void* mask, const void* x_a, const float* invstd_a, int32_t ch_off_a, float* partial_a,
                                 const void* x_b, const float* invstd_b, int32_t ch_off_b, float* partial_b, const float* mean_tab, int32_t ch_total,
                                 int64_t n_pixels, int32_t C, int64_t pixels_per_group, int32_t dtype, void* stream) {
    if (!dout || !x_a || !x_b || !mean_tab || !invstd_a || !invstd_b || !partial_a || !partial_b) FB_FAIL(FB_ERR_ARG, "fb_bn_bwd_reduce2: null pointer");
    if (pixels_per_group % 128 != 0) FB_FAIL(FB_ERR_SHAPE, "fb_bn_bwd_reduce2: pixels_per_group=%lld must be a multiple of 128", (long long)pixels_per_group);
    const int V = dtype == FB_F32 ? 4 : 8;
    if (C % V != 0 || 256 % (C / V) != 0) FB_FAIL(FB_ERR_UNSUPPORTED, "fb_bn_bwd_reduce2: C=%d (the channel vectors of a pixel must divide 256)", C);
    const int cvec = C / V, rows = 256 / cvec;
    const int PB = bn_reduce_pixels(n_pixels, pixels_per_group);
    const int n_mblocks = fb_bn_bwd_reduce_rows(n_pixels, pixels_per_group);
    const size_t smem = (size_t)rows * C * 3 * sizeof(float);
    const int32_t info[FB_PROF_INFO] = {(int32_t)(n_pixels / 128), C, (int32_t)(pixels_per_group / 128), dtype, 1 /* dual */, mask ? 1 : 0, 0, 0, 0, 0, 0};
    const int prof = fb_prof_begin(FB_PROF_BN_BWD_REDUCE, (hipStream_t)stream, info);
    const BnReduceB B{(const uint4*)x_b, invstd_b, partial_b, ch_off_b};
    if (dtype == FB_F32)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, true>), dim3(n_mblocks), dim3(256), smem, (hipStream_t)stream, (const uint4*)dout, (const uint4*)nullptr,
                           (const unsigned char*)mask, (const uint4*)x_a, mean_tab, invstd_a, ch_total, ch_off_a, partial_a, (long long)n_pixels, C, (long long)pixels_per_group, n_mblocks, PB, B);
    else
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_tag, true>), dim3(n_mblocks), dim3(256), smem, (hipStream_t)stream, (const uint4*)dout, (const uint4*)nullptr,
                           (const unsigned char*)mask, (const uint4*)x_a, mean_tab, invstd_a, ch_total, ch_off_a, partial_a, (long long)n_pixels, C, (long long)pixels_per_group, n_mblocks, PB, B);
    fb_prof_end(prof, (hipStream_t)stream);
    FB_CHECK_LAUNCH("fb_bn_bwd_reduce2");
    return FB_OK;
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int n_mblocks, int blocks_per_group, int C, double inv_count,
                                       const float* __restrict__ scale, const float* __restrict__ mean_tab, const float* __restrict__ invstd,
                                       int ch_total, int ch_off, float* __restrict__ dgamma, float* __restrict__ dbeta, long long gstride,
                                       float* __restrict__ coef, int raw_x) {
    const int g = blockIdx.y;
    int c; double s1, s2;
    if (!partial_sum2(part, n_mblocks, blocks_per_group, C, g, c, s1, s2)) return;
    if (raw_x)                                              // partials of dy*x from a convolution epilogue: sum dy*xhat = invstd*(sum dy*x - mean*sum dy)
        s2 = (double)invstd[(long long)g * C + c] * (s2 - (double)mean_tab[(long long)g * ch_total + ch_off + c] * s1);
    dbeta[(long long)g * gstride + c] = (float)s1;
    dgamma[(long long)g * gstride + c] = (float)s2;
    // dx = scale*(dy - s1/M - xhat*s2/M) = c_dy*dy + c_x*x + c_0
    const double sc = scale[(long long)g * C + c], is = invstd[(long long)g * C + c], mu = mean_tab[(long long)g * ch_total + ch_off + c];
    const double cx = -sc * is * s2 * inv_count;
    float* o = coef + ((long long)g * C + c) * 3;
    o[0] = (float)sc; o[1] = (float)cx; o[2] = (float)(-sc * s1 * inv_count - cx * mu);
}

extern "C" int fb_bn_bwd_finalize(const float* partial, int32_t n_mblocks, int32_t n_groups, int32_t C, double count, const float* scale,
                                  const float* mean_tab, const float* invstd, int32_t ch_total, int32_t ch_off, float* dgamma,
                                  float* dbeta, int64_t grad_group_stride, float* coef, int32_t raw_x, void* stream) {
    if (!partial || !scale || !mean_tab || !invstd || !dgamma || !dbeta || !coef) FB_FAIL(FB_ERR_ARG, "fb_bn_bwd_finalize: null pointer");
    if (n_mblocks % n_groups != 0) FB_FAIL(FB_ERR_SHAPE, "fb_bn_bwd_finalize: blocks/groups");
    dim3 grid((C + 15) / 16, n_groups);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, grid, dim3(bn_finalize_threads(n_mblocks / n_groups)), 0, (hipStream_t)stream, partial, n_mblocks, n_mblocks / n_groups, C, 1.0 / count,
                       scale, mean_tab, invstd, ch_total, ch_off, dgamma, dbeta, (long long)grad_group_stride, coef, (int)raw_x);
    FB_CHECK_LAUNCH("fb_bn_bwd_finalize");
    return FB_OK;
}

template <typename T>
__global__ void bn_bwd_apply_kernel(const uint4* __restrict__ dout, const uint4* __restrict__ y, const unsigned char* __restrict__ mask,
                                    const uint4* __restrict__ x,
                                    const float* __restrict__ coef, uint4* __restrict__ dx, uint4* __restrict__ dy_out, long long n_vec,
                                    int cvec, long long vec_per_group, int C) {
    constexpr int V = ET<T>::VEC;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (long long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cvec) * V;
        const long long g = i / vec_per_group;
        float d[V], xv[V], yv[V], o[V], dyv[V];
        ET<T>::unpack(dout[i], d); ET<T>::unpack(x[i], xv);
        unsigned mk = 0xffu;
        if (mask) mk = mask[i];
        else if (y) {
            ET<T>::unpack(y[i], yv);
            mk = 0;
#pragma unroll
            for (int k = 0; k < V; ++k) mk |= (yv[k] > 0.f ? 1u : 0u) << k;
        }
        const float* cf = coef + (g * C + c0) * 3;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float dy = ((mk >> k) & 1u) ? d[k] : 0.f;
            dyv[k] = dy;
            o[k] = fb_bn_dx(cf[3 * k], cf[3 * k + 1], cf[3 * k + 2], dy, xv[k]);
        }
        dx[i] = ET<T>::pack(o);
        if (dy_out) dy_out[i] = ET<T>::pack(dyv);
    }
}

// DUAL: the second BatchNorm of fb_bn_bwd_reduce2 -- dx_b = c_dy' * dy + c_x' * x_b + c_0' from the same (dout, mask) read
struct BnApplyB { const uint4* x; const float* coef; uint4* dx; };
template <typename T, bool DUAL = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_span_kernel(const uint4* __restrict__ dout, const uint4* __restrict__ y,
                                                                const unsigned char* __restrict__ mask, const uint4* __restrict__ x,
                                                                const float* __restrict__ coef, uint4* __restrict__ dx, uint4* __restrict__ dy_out,
                                                                long long n_vec, int cvec, long long vec_per_group, int C, int span, float* __restrict__ amax,
                                                                BnApplyB B) {
    constexpr int V = ET<T>::VEC;
    float am = 0.f;
    const long long g = blockIdx.y, base = g * vec_per_group;
    const long long lim = n_vec - base < vec_per_group ? n_vec - base : vec_per_group;
    const long long lo = (long long)blockIdx.x * span;
    const long long hi = lo + span < lim ? lo + span : lim;
    const int c0 = (int)(threadIdx.x % cvec) * V;
    float cf[3 * V];
    load_coef<3 * V>(coef + (g * C + c0) * 3, cf);
    float cfB[DUAL ? 3 * V : 1];
    if constexpr (DUAL) { load_coef<3 * V>(B.coef + (g * C + c0) * 3, cfB); B.x += base; B.dx += base; }
    dout += base; x += base; dx += base;
    if (mask) mask += base;
    if (y) y += base;
    if (dy_out) dy_out += base;
    auto one = [&](const uint4& dr, const uint4& xr, const uint4& xbr, unsigned mk, long long i) {
        float d[V], xv[V], o[V], dyv[V];
        ET<T>::unpack(dr, d); ET<T>::unpack(xr, xv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float dy = ((mk >> k) & 1u) ? d[k] : 0.f;
            dyv[k] = dy;
            o[k] = fb_bn_dx(cf[3 * k], cf[3 * k + 1], cf[3 * k + 2], dy, xv[k]);
        }
        if (amax) {
#pragma unroll
            for (int k = 0; k < V; ++k) am = fmaxf(am, fabsf(o[k]));
        }
        st_stream(dx + i, ET<T>::pack(o));
        if (dy_out) st_stream(dy_out + i, ET<T>::pack(dyv));
        if constexpr (DUAL) {
            float xb[V], ob[V];
            ET<T>::unpack(xbr, xb);
#pragma unroll
            for (int k = 0; k < V; ++k) ob[k] = fb_bn_dx(cfB[3 * k], cfB[3 * k + 1], cfB[3 * k + 2], dyv[k], xb[k]);
            st_stream(B.dx + i, ET<T>::pack(ob));
        }
    };
    auto mask_at = [&](long long i) -> unsigned {
        if (mask) return mask[i];
        if (!y) return 0xffu;
        float yv[V];
        ET<T>::unpack(y[i], yv);
        unsigned mk = 0;
#pragma unroll
        for (int k = 0; k < V; ++k) mk |= (yv[k] > 0.f ? 1u : 0u) << k;
        return mk;
    };
    long long i = lo + threadIdx.x;
    for (; i + (BN_SPAN_U - 1) * 256 < hi; i += BN_SPAN_U * 256) {
        uint4 dr[BN_SPAN_U], xr[BN_SPAN_U], xbr[BN_SPAN_U];
        unsigned mk[BN_SPAN_U];
#pragma unroll
        for (int u = 0; u < BN_SPAN_U; ++u) {
            dr[u] = ld_stream(dout + i + u * 256); xr[u] = ld_stream(x + i + u * 256); mk[u] = mask_at(i + u * 256);
            if constexpr (DUAL) xbr[u] = ld_stream(B.x + i + u * 256); else xbr[u] = xr[u];
        }
#pragma unroll
        for (int u = 0; u < BN_SPAN_U; ++u) one(dr[u], xr[u], xbr[u], mk[u], i + u * 256);
    }
    for (; i < hi; i += 256) { if constexpr (DUAL) one(dout[i], x[i], B.x[i], mask_at(i), i); else one(dout[i], x[i], x[i], mask_at(i), i); }
    if (amax) amax_commit(amax, am);                          // (uniform branch) one value per run; folded per statistics group afterwards
}

extern "C" int fb_bn_bwd_apply(const void* dout, const void* y, const void* mask, const void* x, const float* coef, void* dx, void* dy_out,
                               int64_t n_pixels, int32_t C, int64_t pixels_per_group, int32_t dtype, float* amax_out, float* amax_ws, void* stream) {
    if (!dout || !x || !coef || !dx) FB_FAIL(FB_ERR_ARG, "fb_bn_bwd_apply: null pointer");
    if (amax_out && (dtype != FB_F32 || !amax_ws)) FB_FAIL(FB_ERR_ARG, "fb_bn_bwd_apply: amax_out is for fp32 tensors and needs amax_ws (fb_ws_bn_amax_floats)");
    const int64_t n_groups = (n_pixels + pixels_per_group - 1) / pixels_per_group;
    const int V = dtype == FB_F32 ? 4 : 8;
    const int cvec = C / V;
    const long long n_vec = n_pixels * cvec, vpg = pixels_per_group * cvec;
    const int32_t info[FB_PROF_INFO] = {(int32_t)(n_pixels / 128), C, (int32_t)(pixels_per_group / 128), dtype, 0, (mask || y) ? 1 : 0, dy_out ? 1 : 0, 0, 0, 0, 0};
    if (256 % cvec == 0 && vpg > 0) {
        const int prof = fb_prof_begin(FB_PROF_BN_BWD_APPLY, (hipStream_t)stream, info);
        const int span = bn_span(n_vec);
        const dim3 grid((unsigned)((vpg + span - 1) / span), (unsigned)((n_vec + vpg - 1) / vpg));
        if (dtype == FB_F32)
            hipLaunchKernelGGL((bn_bwd_apply_span_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)dout, (const uint4*)y,
                               (const unsigned char*)mask, (const uint4*)x, coef, (uint4*)dx, (uint4*)dy_out, n_vec, cvec, vpg, C, span, amax_out ? amax_ws : nullptr, BnApplyB{});
        else
            hipLaunchKernelGGL((bn_bwd_apply_span_kernel<bf16_tag>), grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)dout, (const uint4*)y,
                               (const unsigned char*)mask, (const uint4*)x, coef, (uint4*)dx, (uint4*)dy_out, n_vec, cvec, vpg, C, span, nullptr, BnApplyB{});
        fb_prof_end(prof, (hipStream_t)stream);
        if (amax_out) hipLaunchKernelGGL(amax_finish_kernel, dim3(grid.y), dim3(256), 0, (hipStream_t)stream, amax_ws, (int)grid.x, amax_out);
        FB_CHECK_LAUNCH("fb_bn_bwd_apply");
        return FB_OK;
    }
    const int blocks = (int)((n_vec + 255) / 256 < 8192 ? (n_vec + 255) / 256 : 8192);
    if (dtype == FB_F32)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)dout, (const uint4*)y,
                           (const unsigned char*)mask, (const uint4*)x, coef, (uint4*)dx, (uint4*)dy_out, n_vec, cvec, vpg, C);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_tag>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)dout, (const uint4*)y,
                           (const unsigned char*)mask, (const uint4*)x, coef, (uint4*)dx, (uint4*)dy_out, n_vec, cvec, vpg, C);
    FB_CHECK_LAUNCH("fb_bn_bwd_apply");
    if (amax_out) return fb_absmax((const float*)dx, pixels_per_group * C, (int32_t)n_groups, pixels_per_group * C, 1, amax_out, stream);
    return FB_OK;
}


/* The apply step of two BatchNorms that share the incoming gradient and its ReLU mask (fb_bn_bwd_reduce2): one read of dout for both dx tensors. */
extern "C" int fb_bn_bwd_apply2(const void* dout, const void* mask, const void* x_a, const float* coef_a, void* dx_a, const void* x_b, const float* coef_b,
                                void* dx_b, int64_t n_pixels, int32_t C, int64_t pixels_per_group, int32_t dtype, void* stream) {
    if (!dout || !x_a || !coef_a || !dx_a || !x_b || !coef_b || !dx_b) FB_FAIL(FB_ERR_ARG, "fb_bn_bwd_apply2: null pointer");
    const int V = dtype == FB_F32 ? 4 : 8;
    if (C % V != 0 || 256 % (C / V) != 0 || pixels_per_group <= 0) FB_FAIL(FB_ERR_UNSUPPORTED, "fb_bn_bwd_apply2: C=%d (the channel vectors of a pixel must divide 256)", C);
    const int cvec = C / V;
    const long long n_vec = n_pixels * cvec, vpg = pixels_per_group * cvec;
    const int32_t info[FB_PROF_INFO] = {(int32_t)(n_pixels / 128), C, (int32_t)(pixels_per_group / 128), dtype, 1 /* dual */, mask ? 1 : 0, 0, 0, 0, 0, 0};
    const int prof = fb_prof_begin(FB_PROF_BN_BWD_APPLY, (hipStream_t)stream, info);
    const int span = bn_span(n_vec);
    const dim3 grid((unsigned)((vpg + span - 1) / span), (unsigned)((n_vec + vpg - 1) / vpg));
    const BnApplyB B{(const uint4*)x_b, coef_b, (uint4*)dx_b};
    if (dtype == FB_F32)
        hipLaunchKernelGGL((bn_bwd_apply_span_kernel<float, true>), grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)dout, (const uint4*)nullptr,
                           (const unsigned char*)mask, (const uint4*)x_a, coef_a, (uint4*)dx_a, (uint4*)nullptr, n_vec, cvec, vpg, C, span, (float*)nullptr, B);
    else
        hipLaunchKernelGGL((bn_bwd_apply_span_kernel<bf16_tag, true>), grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)dout, (const uint4*)nullptr,
                           (const unsigned char*)mask, (const uint4*)x_a, coef_a, (uint4*)dx_a, (uint4*)nullptr, n_vec, cvec, vpg, C, span, (float*)nullptr, B);
    fb_prof_end(prof, (hipStream_t)stream);
    FB_CHECK_LAUNCH("fb_bn_bwd_apply2");
    return FB_OK;
}
