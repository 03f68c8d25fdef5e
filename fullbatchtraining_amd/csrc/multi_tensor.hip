// Fused multi-tensor passes over the flat fp32 parameter / gradient arena (P ~ 11.17 M floats for ResNet-18).
// Pure HBM streaming: float4 lanes, grid-stride, FB_MT_BLOCKS blocks; every reduction is two-stage with a fixed
// order.  Replaces the torch._foreach_* / pow(2).sum() / SGD loops of reference training.py:45-47,162,198-211,
// modules.py:215-240 and torch.optim.SGD.step.
#include "common.h"

static inline int mt_blocks(int64_t n) {
    const int64_t b = (n / 4 + 255) / 256;
    return (int)(b < 1 ? 1 : (b > FB_MT_BLOCKS ? FB_MT_BLOCKS : b));
}

// second stage: out[g*NV + v] = sum_b ws[(g*NV+v)*FB_MT_BLOCKS + b] in double, fixed order
// (one 256-thread workgroup per output: thread t adds partials t, t+256, ..., then a fixed binary tree -- one thread walking all
// FB_MT_BLOCKS partials took 90 us per launch)
__global__ __launch_bounds__(256) void mt_finalize_kernel(const float* __restrict__ ws, float* __restrict__ out, int nblocks, int count) {
    __shared__ double sm[256];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += (double)ws[(long long)i * FB_MT_BLOCKS + b];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) sm[threadIdx.x] += sm[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[i] = (float)sm[0];
}

// ---- |scale*x[g] + add_scale*add|^2 ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mt_sqnorm_kernel(const float* __restrict__ x, long long gstride, long long n, float scale,
                                                        const float* __restrict__ add, float add_scale, float* __restrict__ ws) {
    __shared__ float sm[8];
    const int g = blockIdx.y;
    const float* xg = x + (long long)g * gstride;
    float acc[1] = {0.f};
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = ((const float4*)xg)[i];
        float a = v.x * scale, b = v.y * scale, c = v.z * scale, d = v.w * scale;
        if (add) { const float4 p = ((const float4*)add)[i]; a += add_scale * p.x; b += add_scale * p.y; c += add_scale * p.z; d += add_scale * p.w; }
        acc[0] += a * a + b * b + c * c + d * d;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long long i = n4 * 4; i < n; ++i) { const float a = xg[i] * scale + (add ? add_scale * add[i] : 0.f); acc[0] += a * a; }
    block_reduce_sum<1>(acc, sm);
    if (threadIdx.x == 0) ws[(long long)g * FB_MT_BLOCKS + blockIdx.x] = acc[0];
}

extern "C" int fb_mt_sqnorm(const float* x, int64_t group_stride, int32_t n_groups, int64_t n, float scale, const float* add, float add_scale,
                            float* out, float* ws, void* stream) {
    if (!x || !out || !ws) FB_FAIL(FB_ERR_ARG, "fb_mt_sqnorm: null pointer");
    if (((uintptr_t)x & 15) || ((uintptr_t)add & 15) || (group_stride & 3)) FB_FAIL(FB_ERR_ARG, "fb_mt_sqnorm: 16-byte alignment required");
    const int nb = mt_blocks(n);
    hipLaunchKernelGGL(mt_sqnorm_kernel, dim3(nb, n_groups), dim3(256), 0, (hipStream_t)stream, x, (long long)group_stride, (long long)n, scale, add,
                       add_scale, ws);
    hipLaunchKernelGGL(mt_finalize_kernel, dim3(n_groups), dim3(256), 0, (hipStream_t)stream, ws, out, nb, n_groups);
    FB_CHECK_LAUNCH("fb_mt_sqnorm");
    return FB_OK;
}

// ---- running mean over the chunks of a group batch (+ fused per-chunk squared norms) --------------------------------------
// Up to 8 chunks per launch are folded into the running mean in registers (static unroll keeps the per-chunk norm
// accumulators out of scratch); the host wrapper walks larger batches 8 at a time.
// (skip: up to four ranges [lo, hi) of 16-byte vectors that the pass leaves alone -- the layers whose running mean comes from their group sum,
// fb_mt_accumulate_sum; their arena rows hold nothing)
struct MtSkip { long long lo[4], hi[4]; };
template <bool SQ>
__global__ __launch_bounds__(256) void mt_accumulate_kernel(float* __restrict__ avg, const float* __restrict__ g, long long gstride, int n_groups,
                                                            long long n, int counter0, float* __restrict__ ws, const MtSkip skip) {
    constexpr int NG = 8;
    __shared__ float sm[8 * NG];
    const long long n4 = n / 4;
    float sq[NG], inv[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) { sq[j] = 0.f; inv[j] = (float)(1.0 / (double)(counter0 + j + 1)); }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        if ((i >= skip.lo[0] && i < skip.hi[0]) || (i >= skip.lo[1] && i < skip.hi[1]) || (i >= skip.lo[2] && i < skip.hi[2]) || (i >= skip.lo[3] && i < skip.hi[3])) continue;
        float4 a = ((float4*)avg)[i];
#pragma unroll
        for (int j = 0; j < NG; ++j)
            if (j < n_groups) {
                const float4 v = ((const float4*)(g + (long long)j * gstride))[i];
                if (SQ) sq[j] += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                a.x += (v.x - a.x) * inv[j]; a.y += (v.y - a.y) * inv[j]; a.z += (v.z - a.z) * inv[j]; a.w += (v.w - a.w) * inv[j];
            }
        ((float4*)avg)[i] = a;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long long i = n4 * 4; i < n; ++i) {
            float a = avg[i];
#pragma unroll
            for (int j = 0; j < NG; ++j)
                if (j < n_groups) {
                    const float v = g[(long long)j * gstride + i];
                    if (SQ) sq[j] += v * v;
                    a += (v - a) * inv[j];
                }
            avg[i] = a;
        }
    if (SQ) {
        block_reduce_sum<NG>(sq, sm);
        if (threadIdx.x == 0)
#pragma unroll
            for (int j = 0; j < NG; ++j)
                if (j < n_groups) ws[(long long)j * FB_MT_BLOCKS + blockIdx.x] = sq[j];
    }
}

static int mt_accumulate_impl(float* avg, const float* g, int64_t group_stride, int32_t n_groups, int64_t n, int32_t counter0, float* sq_out,
                              float* ws, const MtSkip& skip, void* stream) {
    if (!avg || !g) FB_FAIL(FB_ERR_ARG, "fb_mt_accumulate: null pointer");
    if (sq_out && !ws) FB_FAIL(FB_ERR_ARG, "fb_mt_accumulate: workspace required for norms");
    if (((uintptr_t)avg & 15) || ((uintptr_t)g & 15) || (group_stride & 3)) FB_FAIL(FB_ERR_ARG, "fb_mt_accumulate: 16-byte alignment required");
    const int nb = mt_blocks(n);
    for (int j0 = 0; j0 < n_groups; j0 += 8) {
        const int ng = n_groups - j0 < 8 ? n_groups - j0 : 8;
        const float* gj = g + (long long)j0 * group_stride;
        if (sq_out)
            hipLaunchKernelGGL((mt_accumulate_kernel<true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, avg, gj, (long long)group_stride, ng,
                               (long long)n, counter0 + j0, ws + (long long)j0 * FB_MT_BLOCKS, skip);
        else
            hipLaunchKernelGGL((mt_accumulate_kernel<false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, avg, gj, (long long)group_stride, ng,
                               (long long)n, counter0 + j0, ws, skip);
    }
    if (sq_out) hipLaunchKernelGGL(mt_finalize_kernel, dim3(n_groups), dim3(256), 0, (hipStream_t)stream, ws, sq_out, nb, n_groups);
    FB_CHECK_LAUNCH("fb_mt_accumulate");
    return FB_OK;
}

extern "C" int fb_mt_accumulate(float* avg, const float* g, int64_t group_stride, int32_t n_groups, int64_t n, int32_t counter0, float* sq_out,
                                float* ws, void* stream) {
    return mt_accumulate_impl(avg, g, group_stride, n_groups, n, counter0, sq_out, ws, MtSkip{{0, 0, 0, 0}, {0, 0, 0, 0}}, stream);
}

// fb_mt_accumulate over [0, n) except up to four ranges [skip_lo[k], skip_hi[k]) (multiples of 4 floats; lo == hi: unused): neither read nor
// written, and they do not enter the norms -- the layers that take their mean from fb_mt_accumulate_sum (ABI v12)
extern "C" int fb_mt_accumulate_skip(float* avg, const float* g, int64_t group_stride, int32_t n_groups, int64_t n, int32_t counter0, float* sq_out,
                                     float* ws, int64_t lo0, int64_t hi0, int64_t lo1, int64_t hi1, int64_t lo2, int64_t hi2, int64_t lo3, int64_t hi3,
                                     void* stream) {
    const int64_t lo[4] = {lo0, lo1, lo2, lo3}, hi[4] = {hi0, hi1, hi2, hi3};
    MtSkip skip;
    for (int k = 0; k < 4; ++k) {
        if (lo[k] < 0 || hi[k] < lo[k] || hi[k] > n || (lo[k] & 3) || (hi[k] & 3)) FB_FAIL(FB_ERR_ARG, "fb_mt_accumulate_skip: range %d = [%lld, %lld) of %lld", k, (long long)lo[k], (long long)hi[k], (long long)n);
        skip.lo[k] = lo[k] / 4; skip.hi[k] = hi[k] / 4;
    }
    return mt_accumulate_impl(avg, g, group_stride, n_groups, n, counter0, sq_out, ws, skip, stream);
}

// ---- running mean advanced by a whole batch of chunks from their sum (chunk-chained weight gradients) ------------------------------------
__global__ __launch_bounds__(256) void mt_accumulate_sum_kernel(float* __restrict__ avg, const float* __restrict__ gsum, long long n, float n_groups, float inv) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float a = avg[i];
        avg[i] = a + (gsum[i] - n_groups * a) * inv;
    }
}

extern "C" int fb_mt_accumulate_sum(float* avg, const float* gsum, int64_t n, int32_t counter0, int32_t n_groups, void* stream) {
    if (!avg || !gsum) FB_FAIL(FB_ERR_ARG, "fb_mt_accumulate_sum: null pointer");
    if (n_groups < 1 || counter0 < 0) FB_FAIL(FB_ERR_ARG, "fb_mt_accumulate_sum: n_groups=%d counter0=%d", n_groups, counter0);
    hipLaunchKernelGGL(mt_accumulate_sum_kernel, dim3(mt_blocks(n)), dim3(256), 0, (hipStream_t)stream, avg, gsum, (long long)n, (float)n_groups,
                       (float)(1.0 / (double)(counter0 + n_groups)));
    FB_CHECK_LAUNCH("fb_mt_accumulate_sum");
    return FB_OK;
}

// ---- finite-difference perturbation: theta[g] = theta0 + sign*eps_n[g]*(s*g[g] + acc*pre) ---------------------------------
__global__ void mt_epsn_kernel(const float* __restrict__ vnorm2, float eps, float* __restrict__ eps_n, int n_groups) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_groups) eps_n[g] = eps / sqrtf(vnorm2[g]);
}
__global__ __launch_bounds__(256) void mt_fd_perturb_kernel(const float* __restrict__ theta0, const float* __restrict__ g, long long gstride,
                                                            long long n, float s, float sign, const float* __restrict__ eps_n,
                                                            const float* __restrict__ pre, float acc, float* __restrict__ out) {
    const int grp = blockIdx.y;
    const float alpha = sign * eps_n[grp];
    const float* gg = g + (long long)grp * gstride; float* og = out + (long long)grp * gstride;
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 t = ((const float4*)theta0)[i], v = ((const float4*)gg)[i];
        float4 w = make_float4(s * v.x, s * v.y, s * v.z, s * v.w);
        if (pre) { const float4 q = ((const float4*)pre)[i]; w.x += acc * q.x; w.y += acc * q.y; w.z += acc * q.z; w.w += acc * q.w; }
        ((float4*)og)[i] = make_float4(t.x + alpha * w.x, t.y + alpha * w.y, t.z + alpha * w.z, t.w + alpha * w.w);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long long i = n4 * 4; i < n; ++i) og[i] = theta0[i] + alpha * (s * gg[i] + (pre ? acc * pre[i] : 0.f));
}

extern "C" int fb_mt_fd_perturb(const float* theta0, const float* g, int64_t group_stride, int32_t n_groups, int64_t n, float s, float eps,
                                float sign, const float* vnorm2, float* eps_n, const float* pre, float acc, float* theta_out, void* stream) {
    if (!theta0 || !g || !vnorm2 || !eps_n || !theta_out) FB_FAIL(FB_ERR_ARG, "fb_mt_fd_perturb: null pointer");
    hipLaunchKernelGGL(mt_epsn_kernel, dim3((n_groups + 63) / 64), dim3(64), 0, (hipStream_t)stream, vnorm2, eps, eps_n, n_groups);
    hipLaunchKernelGGL(mt_fd_perturb_kernel, dim3(mt_blocks(n), n_groups), dim3(256), 0, (hipStream_t)stream, theta0, g, (long long)group_stride,
                       (long long)n, s, sign, eps_n, pre, acc, theta_out);
    FB_CHECK_LAUNCH("fb_mt_fd_perturb");
    return FB_OK;
}

// ---- vhp = (ga-gb)/eps_n ; gt = g + cf*vhp ; running mean ------------------------------------------------------------------
__global__ __launch_bounds__(256) void mt_fd_combine_kernel(float* __restrict__ avg, const float* __restrict__ g, const float* __restrict__ ga,
                                                            const float* __restrict__ gb, long long gstride, int n_groups, long long n,
                                                            const float* __restrict__ eps_n, float cf, int counter0) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float a = avg[i];
        for (int j = 0; j < n_groups; ++j) {
            const long long o = (long long)j * gstride + i;
            const float vhp = (ga[o] - gb[o]) / eps_n[j];
            const float gt = g[o] + cf * vhp;
            a += (gt - a) * (float)(1.0 / (double)(counter0 + j + 1));
        }
        avg[i] = a;
    }
}

extern "C" int fb_mt_fd_combine_accumulate(float* avg, const float* g, const float* ga, const float* gb, int64_t group_stride, int32_t n_groups,
                                           int64_t n, const float* eps_n, float cf, int32_t counter0, void* stream) {
    if (!avg || !g || !ga || !gb || !eps_n) FB_FAIL(FB_ERR_ARG, "fb_mt_fd_combine_accumulate: null pointer");
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(mt_fd_combine_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, avg, g, ga, gb,
                       (long long)group_stride, n_groups, (long long)n, eps_n, cf, counter0);
    FB_CHECK_LAUNCH("fb_mt_fd_combine_accumulate");
    return FB_OK;
}

// ---- the same in two steps, for per-chunk clipping of the regularised gradient (hyp.batch_clip) ------------------------------
__global__ __launch_bounds__(256) void mt_fd_combine_inplace_kernel(float* g, const float* ga, const float* gb,      // (gb may alias g)
                                                                    long long gstride, long long n, const float* __restrict__ eps_n, float cf) {
    const int j = blockIdx.y;
    const float e = eps_n[j];
    float* gj = g + (long long)j * gstride; const float* a = ga + (long long)j * gstride; const float* b = gb + (long long)j * gstride;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float vhp = (a[i] - b[i]) / e;            // same expression (and rounding) as mt_fd_combine_kernel
        gj[i] = gj[i] + cf * vhp;
    }
}
extern "C" int fb_mt_fd_combine(float* g, const float* ga, const float* gb, int64_t group_stride, int32_t n_groups, int64_t n, const float* eps_n,
                                float cf, void* stream) {
    if (!g || !ga || !gb || !eps_n) FB_FAIL(FB_ERR_ARG, "fb_mt_fd_combine: null pointer");
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(mt_fd_combine_inplace_kernel, dim3((unsigned)(nb < 1024 ? nb : 1024), n_groups), dim3(256), 0, (hipStream_t)stream, g, ga, gb,
                       (long long)group_stride, (long long)n, eps_n, cf);
    FB_CHECK_LAUNCH("fb_mt_fd_combine");
    return FB_OK;
}
__global__ __launch_bounds__(256) void mt_chunk_clip_kernel(float* __restrict__ g, long long gstride, long long n, const float* __restrict__ sq,
                                                            float clip, float* __restrict__ clipped) {
    const int j = blockIdx.y;
    const float norm = sqrtf(sq[j]);
    const bool hit = norm > clip;
    if (blockIdx.x == 0 && threadIdx.x == 0 && clipped) clipped[j] = hit ? 1.f : 0.f;
    if (!hit) return;
    const float coef = clip / (norm + 1e-6f);
    float* gj = g + (long long)j * gstride;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) gj[i] *= coef;
}
extern "C" int fb_mt_chunk_clip(float* g, int64_t group_stride, int32_t n_groups, int64_t n, const float* sq, float clip, float* clipped,
                                void* stream) {
    if (!g || !sq) FB_FAIL(FB_ERR_ARG, "fb_mt_chunk_clip: null pointer");
    if (!(clip >= 0.f)) FB_FAIL(FB_ERR_ARG, "fb_mt_chunk_clip: clip=%g must be non-negative", (double)clip);
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(mt_chunk_clip_kernel, dim3((unsigned)(nb < 1024 ? nb : 1024), n_groups), dim3(256), 0, (hipStream_t)stream, g,
                       (long long)group_stride, (long long)n, sq, clip, clipped);
    FB_CHECK_LAUNCH("fb_mt_chunk_clip");
    return FB_OK;
}

// ---- |a|^2 and |b|^2 ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mt_norms2_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, float* __restrict__ ws) {
    __shared__ float sm[16];
    float acc[2] = {0.f, 0.f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float x = a[i]; acc[0] += x * x;
        if (b) { const float y = b[i]; acc[1] += y * y; }
    }
    block_reduce_sum<2>(acc, sm);
    if (threadIdx.x == 0) { ws[blockIdx.x] = acc[0]; ws[FB_MT_BLOCKS + blockIdx.x] = acc[1]; }
}

extern "C" int fb_mt_norms2(const float* a, const float* b, int64_t n, float* out, float* ws, void* stream) {
    if (!a || !out || !ws) FB_FAIL(FB_ERR_ARG, "fb_mt_norms2: null pointer");
    const int64_t want = (n + 255) / 256;
    const int nb = (int)(want < 1 ? 1 : (want > FB_MT_BLOCKS ? FB_MT_BLOCKS : want));
    hipLaunchKernelGGL(mt_norms2_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, (long long)n, ws);
    hipLaunchKernelGGL(mt_finalize_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, ws, out, nb, 2);
    FB_CHECK_LAUNCH("fb_mt_norms2");
    return FB_OK;
}

// ---- clip + SGD (weight decay, momentum, dampening, Nesterov) ---------------------------------------------------------------
__global__ __launch_bounds__(256) void mt_clip_sgd_kernel(float* __restrict__ theta, float* __restrict__ grad, float* __restrict__ mom, long long n,
                                                          const float* __restrict__ gnorm2, float grad_clip, float lr, float wd, float mu,
                                                          float damp, int nesterov, int first) {
    float coef = 1.f;
    if (grad_clip >= 0.f) {
        const float norm = sqrtf(gnorm2[0]);
        if (norm > grad_clip) coef = grad_clip / (norm + 1e-6f);
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float p = theta[i];
        float gr = grad[i];
        if (coef != 1.f) { gr *= coef; grad[i] = gr; }
        float d = gr + wd * p;
        if (mu != 0.f) {
            const float buf = first ? d : mu * mom[i] + (1.f - damp) * d;
            mom[i] = buf;
            d = nesterov ? d + mu * buf : buf;
        }
        theta[i] = p - lr * d;
    }
}

extern "C" int fb_mt_clip_sgd(float* theta, float* grad, float* mom, int64_t n, const float* gnorm2, float grad_clip, float lr,
                              float weight_decay, float momentum, float dampening, int32_t nesterov, int32_t first_step, void* stream) {
    if (!theta || !grad || (!mom && momentum != 0.f)) FB_FAIL(FB_ERR_ARG, "fb_mt_clip_sgd: null pointer");
    if (grad_clip >= 0.f && !gnorm2) FB_FAIL(FB_ERR_ARG, "fb_mt_clip_sgd: clipping needs the device norm");
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(mt_clip_sgd_kernel, dim3((unsigned)(nb < 4096 ? (nb < 1 ? 1 : nb) : 4096)), dim3(256), 0, (hipStream_t)stream, theta, grad, mom,
                       (long long)n, gnorm2, grad_clip, lr, weight_decay, momentum, dampening, nesterov, first_step);
    FB_CHECK_LAUNCH("fb_mt_clip_sgd");
    return FB_OK;
}

__global__ void mt_scale_kernel(float* __restrict__ x, long long n, float a) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] *= a;
}
extern "C" int fb_mt_scale(float* x, int64_t n, float a, void* stream) {
    if (!x) FB_FAIL(FB_ERR_ARG, "fb_mt_scale: null pointer");
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(mt_scale_kernel, dim3((unsigned)(nb < 4096 ? (nb < 1 ? 1 : nb) : 4096)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, a);
    FB_CHECK_LAUNCH("fb_mt_scale");
    return FB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Sharpness-aware minimisation around the full-batch closure (reference additional_optimizers/sam.py:56-82): the ascent step
// e_w = g_c * rho / (|g_c| + 1e-12), theta += e_w with g_c the closure's (clipped, training.py:198-206) gradient, and the exact
// way back theta -= e_w.  All scalars come from the device norm: no host round trip between the two gradient evaluations.
__global__ void mt_sam_ascent_kernel(float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ e_w, long long n,
                                     const float* __restrict__ gnorm2, float grad_clip, float rho) {
    const float norm = sqrtf(gnorm2[0]);
    const float coef = (grad_clip >= 0.f && norm > grad_clip) ? grad_clip / (norm + 1e-6f) : 1.f;
    const float scale = rho / (norm * coef + 1e-12f);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float e = (grad[i] * coef) * scale;
        e_w[i] = e;
        theta[i] += e;
    }
}
extern "C" int fb_mt_sam_ascent(float* theta, const float* grad, float* e_w, int64_t n, const float* gnorm2, float grad_clip, float rho,
                                void* stream) {
    if (!theta || !grad || !e_w || !gnorm2) FB_FAIL(FB_ERR_ARG, "fb_mt_sam_ascent: null pointer");
    if (rho < 0.f) FB_FAIL(FB_ERR_ARG, "fb_mt_sam_ascent: rho=%g must be non-negative", (double)rho);
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(mt_sam_ascent_kernel, dim3((unsigned)(nb < 4096 ? (nb < 1 ? 1 : nb) : 4096)), dim3(256), 0, (hipStream_t)stream, theta, grad,
                       e_w, (long long)n, gnorm2, grad_clip, rho);
    FB_CHECK_LAUNCH("fb_mt_sam_ascent");
    return FB_OK;
}
__global__ void mt_sam_restore_kernel(float* __restrict__ theta, const float* __restrict__ e_w, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) theta[i] -= e_w[i];
}
extern "C" int fb_mt_sam_restore(float* theta, const float* e_w, int64_t n, void* stream) {
    if (!theta || !e_w) FB_FAIL(FB_ERR_ARG, "fb_mt_sam_restore: null pointer");
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(mt_sam_restore_kernel, dim3((unsigned)(nb < 4096 ? (nb < 1 ? 1 : nb) : 4096)), dim3(256), 0, (hipStream_t)stream, theta, e_w,
                       (long long)n);
    FB_CHECK_LAUNCH("fb_mt_sam_restore");
    return FB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Options of the closure's gradient modification that are off by default (reference training.py:187-211) and the
// EMA of the evaluated model (training/utils.py:22-29).
// L-infinity clip norm: out[0] = (max |a_i|)^2, so that the consumers of the squared L2 norm (fb_mt_clip_sgd, fb_mt_sam_ascent,
// the statistics) work unchanged on sqrt(out[0]).
__global__ __launch_bounds__(256) void mt_absmax_kernel(const float* __restrict__ a, long long n, float* __restrict__ ws) {
    __shared__ float sm[256];
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(a[i]));
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) ws[blockIdx.x] = sm[0];
}
__global__ void mt_absmax_finalize_kernel(const float* __restrict__ ws, int nb, float* __restrict__ out) {
    __shared__ float sm[64];
    float m = 0.f;
    for (int i = threadIdx.x; i < nb; i += 64) m = fmaxf(m, ws[i]);
    sm[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 64; ++i) m = fmaxf(m, sm[i]);
        out[0] = m * m;
    }
}
extern "C" int fb_mt_absmax2(const float* a, int64_t n, float* out, float* ws, void* stream) {
    if (!a || !out || !ws) FB_FAIL(FB_ERR_ARG, "fb_mt_absmax2: null pointer");
    const int64_t want = (n + 255) / 256;
    const int nb = (int)(want < 1 ? 1 : (want > FB_MT_BLOCKS ? FB_MT_BLOCKS : want));
    hipLaunchKernelGGL(mt_absmax_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, (long long)n, ws);
    hipLaunchKernelGGL(mt_absmax_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, nb, out);
    FB_CHECK_LAUNCH("fb_mt_absmax2");
    return FB_OK;
}

// external norm bias (training.py:188-196): norm_type 1: grad += strength * sign(|theta|^2 - bias^2); else grad += strength * 2 (|theta|^2 -
// bias^2) * theta.  pnorm2 = |theta|^2 over ALL parameters (device scalar); the range [grad, grad + n) is one parameter tensor -- the
// constant of norm_type 1 must not land in the alignment padding of the arena.
__global__ void mt_norm_bias_kernel(float* __restrict__ grad, const float* __restrict__ theta, long long n, const float* __restrict__ pnorm2,
                                    float strength, float bias, int norm_type) {
    const float diff = pnorm2[0] - bias * bias;
    const float c = norm_type == 1 ? strength * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) : strength * (2.f * diff);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        grad[i] += norm_type == 1 ? c : c * theta[i];
}
extern "C" int fb_mt_norm_bias(float* grad, const float* theta, int64_t n, const float* pnorm2, float strength, float bias, int32_t norm_type,
                               void* stream) {
    if (!grad || !theta || !pnorm2) FB_FAIL(FB_ERR_ARG, "fb_mt_norm_bias: null pointer");
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(mt_norm_bias_kernel, dim3((unsigned)(nb < 4096 ? (nb < 1 ? 1 : nb) : 4096)), dim3(256), 0, (hipStream_t)stream, grad, theta,
                       (long long)n, pnorm2, strength, bias, norm_type);
    FB_CHECK_LAUNCH("fb_mt_norm_bias");
    return FB_OK;
}

// ema = momentum * ema + one_minus * src, two roundings and an add like the reference's tensor expression (no fma contraction)
__global__ void mt_ema_kernel(float* __restrict__ ema, const float* __restrict__ src, long long n, float momentum, float one_minus) {
#pragma clang fp contract(off)     // hipcc contracts a*b + c*d into an fma by default (and __fmul_rn / __fadd_rn are plain operators)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float a = momentum * ema[i], b = one_minus * src[i];
        ema[i] = a + b;
    }
}
extern "C" int fb_mt_ema(float* ema, const float* src, int64_t n, float momentum, float one_minus, void* stream) {
    if (!ema || !src) FB_FAIL(FB_ERR_ARG, "fb_mt_ema: null pointer");
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(mt_ema_kernel, dim3((unsigned)(nb < 4096 ? (nb < 1 ? 1 : nb) : 4096)), dim3(256), 0, (hipStream_t)stream, ema, src, (long long)n,
                       momentum, one_minus);
    FB_CHECK_LAUNCH("fb_mt_ema");
    return FB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Gradient noise of the closure (reference training.py:212-215) acts on the CLIPPED gradient, so the clip that fb_mt_clip_sgd
// otherwise fuses is applied in place first (fb_mt_clip_scale), then fb_mt_grad_noise with noise drawn by the host framework's
// generator (the reference draws torch.randn_like per parameter):
//   mode 0: grad += strength * noise          (p.grad.add_(a * randn))       two roundings, like the tensor expression
//   mode 1: grad *= 1 + strength * noise      (p.grad.mul_(1 + m * randn))
__global__ void mt_clip_scale_kernel(float* __restrict__ grad, long long n, const float* __restrict__ gnorm2, float grad_clip) {
    const float norm = sqrtf(gnorm2[0]);
    if (!(norm > grad_clip)) return;
    const float coef = grad_clip / (norm + 1e-6f);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) grad[i] *= coef;
}
extern "C" int fb_mt_clip_scale(float* grad, int64_t n, const float* gnorm2, float grad_clip, void* stream) {
    if (!grad || !gnorm2) FB_FAIL(FB_ERR_ARG, "fb_mt_clip_scale: null pointer");
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(mt_clip_scale_kernel, dim3((unsigned)(nb < 4096 ? (nb < 1 ? 1 : nb) : 4096)), dim3(256), 0, (hipStream_t)stream, grad, (long long)n,
                       gnorm2, grad_clip);
    FB_CHECK_LAUNCH("fb_mt_clip_scale");
    return FB_OK;
}
__global__ void mt_grad_noise_kernel(float* __restrict__ grad, const float* __restrict__ noise, long long n, float strength, int mode) {
#pragma clang fp contract(off)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float t = strength * noise[i];
        if (mode == 0) grad[i] = grad[i] + t;
        else { const float u = 1.f + t; grad[i] = grad[i] * u; }
    }
}
extern "C" int fb_mt_grad_noise(float* grad, const float* noise, int64_t n, float strength, int32_t mode, void* stream) {
    if (!grad || !noise) FB_FAIL(FB_ERR_ARG, "fb_mt_grad_noise: null pointer");
    if (mode != 0 && mode != 1) FB_FAIL(FB_ERR_ARG, "fb_mt_grad_noise: mode %d", mode);
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(mt_grad_noise_kernel, dim3((unsigned)(nb < 4096 ? (nb < 1 ? 1 : nb) : 4096)), dim3(256), 0, (hipStream_t)stream, grad, noise,
                       (long long)n, strength, mode);
    FB_CHECK_LAUNCH("fb_mt_grad_noise");
    return FB_OK;
}

// general p-norm clip (reference training.py:201-204: the p-norm of the per-tensor p-norms = (sum |g_i|^p)^(1/p)):
// out[0] = norm^2, in the slot of the squared L2 norm like fb_mt_absmax2.  Partial sums in float, second stage in double.
__global__ __launch_bounds__(256) void mt_pnorm_kernel(const float* __restrict__ a, long long n, float p, float* __restrict__ ws) {
    __shared__ float sm[16];
    float acc[1] = {0.f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float x = fabsf(a[i]);
        acc[0] += p == 1.f ? x : (x > 0.f ? powf(x, p) : 0.f);
    }
    block_reduce_sum<1>(acc, sm);
    if (threadIdx.x == 0) ws[blockIdx.x] = acc[0];
}
__global__ void mt_pnorm_finalize_kernel(const float* __restrict__ ws, int nb, double inv_p, float* __restrict__ out) {
    __shared__ double sm[64];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) s += (double)ws[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 64; ++i) s += sm[i];
        const double norm = pow(s, inv_p);
        out[0] = (float)(norm * norm);
    }
}
extern "C" int fb_mt_pnorm2(const float* a, int64_t n, float p, float* out, float* ws, void* stream) {
    if (!a || !out || !ws) FB_FAIL(FB_ERR_ARG, "fb_mt_pnorm2: null pointer");
    if (!(p >= 1.f)) FB_FAIL(FB_ERR_ARG, "fb_mt_pnorm2: p=%g (p >= 1; infinity is fb_mt_absmax2)", (double)p);
    const int64_t want = (n + 255) / 256;
    const int nb = (int)(want < 1 ? 1 : (want > FB_MT_BLOCKS ? FB_MT_BLOCKS : want));
    hipLaunchKernelGGL(mt_pnorm_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, (long long)n, p, ws);
    hipLaunchKernelGGL(mt_pnorm_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, nb, 1.0 / (double)p, out);
    FB_CHECK_LAUNCH("fb_mt_pnorm2");
    return FB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Largest magnitude of an fp32 tensor (scale source of the fp16x2 split, common.h): |x| as a bit pattern is monotone, so the maximum is an
// integer atomic; 16-byte loads, one atomic per workgroup.  n_sets slices of n values, set_stride floats apart (the per-chunk weight sets
// of one layer), share one result.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x0, long long n, long long set_stride, unsigned* __restrict__ out0,
                                                     int per_set) {
    unsigned* out = out0 + (per_set ? blockIdx.y : 0);
    const float* xs = x0 + (long long)blockIdx.y * set_stride;
    const uint4* x = (const uint4*)xs;
    const long long n_vec = (((unsigned long long)xs & 15) == 0) ? n / 4 : 0;     // an unaligned slice takes the scalar path below
    unsigned m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (long long)gridDim.x * 256) {
        const uint4 v = x[i];
        const unsigned a = v.x & 0x7fffffffu, b = v.y & 0x7fffffffu, c = v.z & 0x7fffffffu, d = v.w & 0x7fffffffu;
        const unsigned ab = a > b ? a : b, cd = c > d ? c : d, q = ab > cd ? ab : cd;
        m = m > q ? m : q;
    }
    for (long long i = n_vec * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const unsigned a = __float_as_uint(xs[i]) & 0x7fffffffu;
        m = m > a ? m : a;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)m, off); m = m > o ? m : o; }
    __shared__ unsigned red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a = red[0] > red[1] ? red[0] : red[1], b = red[2] > red[3] ? red[2] : red[3];
        atomicMax(out, a > b ? a : b);
    }
}

extern "C" int fb_absmax(const float* x, int64_t n, int32_t n_sets, int64_t set_stride, int32_t per_set, float* out, void* stream) {
    if (!x || !out || n < 0 || n_sets < 1) FB_FAIL(FB_ERR_ARG, "fb_absmax: bad arguments");
    if (((uintptr_t)x & 3) != 0) FB_FAIL(FB_ERR_ARG, "fb_absmax: x must be 4-byte aligned");
    if (hipMemsetAsync(out, 0, sizeof(float) * (per_set ? n_sets : 1), (hipStream_t)stream) != hipSuccess) FB_FAIL(FB_ERR_LAUNCH, "fb_absmax: memset failed");
    long long blocks = (n / 4 + 256 * 8 - 1) / (256 * 8);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    if (per_set && blocks * n_sets > 8192) blocks = (8192 + n_sets - 1) / n_sets;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks, (unsigned)n_sets), dim3(256), 0, (hipStream_t)stream, x, (long long)n, (long long)set_stride, (unsigned*)out,
                       (int)per_set);
    FB_CHECK_LAUNCH("fb_absmax");
    return FB_OK;
}
