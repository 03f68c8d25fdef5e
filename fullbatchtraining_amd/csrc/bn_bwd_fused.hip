// BatchNorm backward (+ ReLU mask) in ONE pass over (dout, x): reduce -> coefficients -> apply with the operands held ON CHIP in between.
//
// The two-pass form (bn.hip: fb_bn_bwd_reduce -> fb_bn_bwd_finalize -> fb_bn_bwd_apply) reads dout and x twice -- 5 tensor passes per layer,
// 310 of the step's 910 GB -- because the per-channel sums of a statistics group (a chunk: 128 images) must be complete before the first dx can
// be written, and a chunk's (dout, x) is 8-67 MB.  But the register files of an MI355X hold 128 MB (256 CUs x 4 SIMDs x 128 KiB): a CLUSTER of
// workgroups, resident together, can keep a whole chunk's operands in registers across the reduction:
//   phase 1  every workgroup of the cluster loads its slice of the chunk (16 (dout, x) vector pairs per thread = 128 VGPRs), accumulates the
//            masked sums of dy and dy * xhat, writes one partial row and ARRIVES at the chunk's counter (one atomic per workgroup);
//   reduce   the workgroup that arrives last adds the cluster's partial rows in fixed order (double), writes dgamma / dbeta and the three
//            coefficients per channel (the arithmetic of fb_bn_bwd_finalize) and releases the chunk's flags;
//   phase 2  every workgroup computes dx = c0 dy + c1 x + c2 from its registers and streams it out.
// 3 tensor passes instead of 5.  Two workgroups per CU (<= 256 VGPRs), persistent: clusters walk the chunks round-robin, so while one cluster
// waits for its reduction the other workgroup of the CU is loading or storing.  Results do not depend on which workgroup reduces (the order
// of the sum is fixed) nor on how many chunks a launch holds (a chunk's slices depend on (C, pixels per chunk) only).
//
// Co-residency is what makes the wait safe: a cluster never has more workgroups than the device has slots (2 x CUs; the entry point refuses
// otherwise and the caller takes the two-pass form), workgroups are dispatched in index order, and whatever else holds CUs (the weight-gradient
// stream) finishes without waiting for this kernel.  A wait that still exceeds ~2 s sets the error word of the workspace instead of hanging.
#include "common.h"
#include "profile.h"

namespace {
#ifndef FB_BNF_NP
#define FB_BNF_NP 16
#define FB_BNF_WGCU 2
#endif
constexpr int BF_NP = FB_BNF_NP;          // (dout, x) vector pairs per thread
constexpr int BF_WGCU = FB_BNF_WGCU;      // resident workgroups per CU (register budget 512 / BF_WGCU per lane)
typedef __attribute__((ext_vector_type(4))) unsigned bf_u32x4_t;

struct BnBwdFusedParams {
    const uint4* dout; const unsigned char* mask; const uint4* x;
    const float* mean_tab; const float* invstd; const float* scale;
    float* dgamma; float* dbeta; float* coef;
    uint4* dx; uint4* dy_out;
    float* partial;                        // [n_groups][ncw][2][C]
    double* red2;                          // [n_groups][2][C] final sums (the reducer's own scratch, behind the partial rows)
    int* sync;                             // [n_groups][BF_SYNC] counters and flags, [1] error word
    long long gstride, vpg;                // gradient arena stride; 16-byte vectors per statistics group
    double inv_count;
    int n_groups, C, ch_total, ch_off, ncw, n_clusters, poll;
    long long* trace;                      // development: six 100 MHz timestamps per (group, workgroup) -- fb_bn_bwd_fused_trace()
};

// Slices are addressed as  descriptor (slice base) + one lane register (tid * 16) + a scalar offset (j * 4096): 64-bit addresses per load would
// cost 64 VGPRs next to the 128 that hold the data.  Streaming (nt) accesses: every tensor is touched once.
constexpr int BF_NT = 2, BF_SC1 = 16;          // cache-policy bits of the buffer instructions: nt (streaming), sc1 (device-coherent)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bf_rsrc(const void* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint4 bf_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const bf_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, BF_NT);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void bf_st(__amdgpu_buffer_rsrc_t r, int voff, int soff, const uint4& v) {
    const bf_u32x4_t d = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, BF_NT);
    store_b128_guard(d);                                             // (common.h: store data of a wide MUBUF store with a register offset)
}
// Cross-workgroup data (partial rows, coefficients, counters, flags) moves through device-coherent accesses only: agent-scope stores are
// written through to memory, agent-scope loads read it there, and a wave waits for its own stores (vmcnt) before it signals.  No release /
// acquire FENCES: at agent scope they write back and invalidate the whole L2 of the XCD -- issued from every polling workgroup that made the
// kernel 14x slower than the two-pass form (17.4 ms for the 64-channel layer).
__device__ __forceinline__ float bf_ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void bf_st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void bf_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
}  // namespace

// sync words of statistics group g (BF_SYNC ints): [0] arrivals (top level), [1..16] arrivals of the 16 sub-groups, [17] unused,
// [32..63] "reduced" flags (pollers spread over 32 addresses), [64..95] "loaded" flags (the next cluster starts its first chunk behind them)
constexpr int BF_SYNC = 96, BF_SUB = 16;

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BF_WGCU, BF_WGCU))) void bn_bwd_fused_kernel(const BnBwdFusedParams p) {
    constexpr int V = ET<T>::VEC;
    __shared__ __attribute__((aligned(16))) float sm[256 * V * 2];      // [rows][C][2] with rows * C = 256 * V; the reducer's scratch
    __shared__ int last_flag;
    const int tid = threadIdx.x, C = p.C, cvec = C / V, rows = 256 / cvec;
    const int cv = tid % cvec, row = tid / cvec, c0 = cv * V;
    // a cluster = ncw CONSECUTIVE workgroups (consecutive indices go round the XCDs, so it is spread over all of them): a prefix of the grid that is
    // resident always contains whole clusters, whatever else (another stream, another process) holds the remaining slots
    const int cluster = blockIdx.x / p.ncw, wi = blockIdx.x - cluster * p.ncw;
    if (cluster >= p.n_clusters) return;
    int* err = p.sync + (long long)p.n_groups * BF_SYNC;
    const int n_sub = p.ncw % BF_SUB == 0 ? BF_SUB : 1, sub_size = p.ncw / n_sub;

    auto wait_flag = [&](const int* f) {                             // thread 0 polls (relaxed, device-coherent), everybody leaves together
        if (p.poll < 0) return;                                      // (timing experiment FB_BNF_POLL=-1: no waits -- results are wrong)
        if (tid == 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                for (int k = 0; k < p.poll; ++k) __builtin_amdgcn_s_sleep(16);       // ~0.5 us per unit: polls of one address serialise in its memory channel
                if (wall_clock64() - t0 > 200000000LL) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // 2 s at 100 MHz
            }
        }
        __syncthreads();
    };

    bool first = true;
    for (int g = cluster; g < p.n_groups; g += p.n_clusters, first = false) {
        int* sy = p.sync + (long long)g * BF_SYNC;
        long long* tr = p.trace ? p.trace + ((long long)g * p.ncw + wi) * 6 : nullptr;
        auto stamp = [&](int k) { if (tr && tid == 0) tr[k] = wall_clock64(); };
        stamp(0);
        // Clusters run out of phase: cluster k starts its first chunk when cluster k - 1 has LOADED its own, so that afterwards one cluster's
        // wait for its reduction lies under another cluster's loads and stores (started together they would stay in lockstep: everybody
        // loads, everybody waits, everybody stores).
        if (first && cluster > 0) wait_flag(sy - BF_SYNC + 64 + (wi % 32));

        // ---- phase 1: load the slice, masked sums ---------------------------------------------------------------------------------
        const long long slice = (long long)g * p.vpg + (long long)wi * (BF_NP * 256);        // first vector of this workgroup's slice
        constexpr int SLICE_B = BF_NP * 256 * 16;
        const __amdgpu_buffer_rsrc_t rD = bf_rsrc(p.dout + slice, SLICE_B), rX = bf_rsrc(p.x + slice, SLICE_B);
        const __amdgpu_buffer_rsrc_t rM = bf_rsrc(p.mask ? p.mask + slice : (const unsigned char*)p.dout, p.mask ? BF_NP * 256 : 0);
        // register pair j holds the vectors at slot (j + rot) % 16 of the slice, rot = the workgroup's index: every workgroup walks its 64 KiB
        // slice from a different 4 KiB slot.  Walking them all from slot 0 puts the whole cluster on the same few memory channels at any moment
        // (slices are 64 KiB apart): 14-16 us per 128 KiB instead of 5-6.
        const int rot = wi & (BF_NP - 1);
        auto slot = [&](int j) { return (j + rot) & (BF_NP - 1); };
        uint4 dr[BF_NP], xr[BF_NP];
        unsigned mk[BF_NP / 4];
#pragma unroll
        for (int j = 0; j < BF_NP; ++j) { dr[j] = bf_ld(rD, tid * 16, slot(j) * 4096); xr[j ^ (BF_NP / 2)] = bf_ld(rX, tid * 16, slot(j ^ (BF_NP / 2)) * 4096); }
#pragma unroll
        for (int q = 0; q < BF_NP / 4; ++q) {
            mk[q] = 0xffffffffu;
            if (p.mask) {
                mk[q] = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) mk[q] |= (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rM, tid, slot(4 * q + e) * 256, BF_NT) << (8 * e);
            }
        }
        float mu[V], is[V], s1[V], s2[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            mu[k] = p.mean_tab[(long long)g * p.ch_total + p.ch_off + c0 + k];
            is[k] = p.invstd[(long long)g * C + c0 + k];
            s1[k] = 0.f; s2[k] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < BF_NP; ++j) {
            float d[V], xv[V];
            ET<T>::unpack(dr[j], d); ET<T>::unpack(xr[j], xv);
            const unsigned m = mk[j >> 2] >> (8 * (j & 3));
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float dy = ((m >> k) & 1u) ? d[k] : 0.f;
                s1[k] += dy; s2[k] += dy * ((xv[k] - mu[k]) * is[k]);
            }
            __builtin_amdgcn_sched_barrier(0);                       // one pair's temporaries at a time: the 128 data registers leave room for little else
        }
#pragma unroll
        for (int k = 0; k < V; ++k) { sm[(row * C + c0 + k) * 2] = s1[k]; sm[(row * C + c0 + k) * 2 + 1] = s2[k]; }
        __syncthreads();
        stamp(1);                                                    // loaded + summed
        float* prow = p.partial + ((long long)g * p.ncw + wi) * 2 * C;      // [C] sums of dy, [C] sums of dy * xhat
        for (int c = tid; c < C; c += 256) {
            float a = 0.f, b = 0.f;
            for (int r = 0; r < rows; ++r) { a += sm[(r * C + c) * 2]; b += sm[(r * C + c) * 2 + 1]; }
            bf_st_agent(prow + c, a); bf_st_agent(prow + C + c, b);
        }
        bf_stores_done();
        __syncthreads();
        if (tid == 0) {                                              // arrive: sub-group counter, its last arrival at the top counter
            int last = 0;
            if (__hip_atomic_fetch_add(sy + 1 + (n_sub > 1 ? wi / sub_size : 0), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sub_size - 1)
                last = __hip_atomic_fetch_add(sy, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_sub - 1;
            last_flag = last;
        }
        __syncthreads();
        stamp(2);                                                    // arrived

        // ---- reduction by the last arrival: partial rows in fixed order, dgamma / dbeta, coefficients, release ----------------------
        if (last_flag) {
            if (tid < 32) __hip_atomic_store(sy + 64 + tid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // "loaded": the next cluster may start
            // rows are [2C] floats = 2C / 4 vectors; thread (v, seg) adds the vector v of rows seg, seg + segs, ... (device-coherent 16-byte
            // loads, eight in flight), then the segment sums are added in order
            double* red = (double*)sm;
            const int nvec = C / 2;
            const __amdgpu_buffer_rsrc_t rP = bf_rsrc(p.partial + (long long)g * p.ncw * 2 * C, p.ncw * 2 * C * 4);
            for (int vb = 0; vb < nvec; vb += 256) {
                const int nv = nvec - vb < 256 ? nvec - vb : 256, segs = 256 / nv;
                const int v = vb + tid % nv, seg = tid / nv;
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                if (seg < segs) {
                    int w = seg;
                    for (; w + 7 * segs < p.ncw; w += 8 * segs) {
                        bf_u32x4_t t[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) t[u] = __builtin_amdgcn_raw_buffer_load_b128(rP, v * 16, (w + u * segs) * 2 * C * 4, BF_SC1);
#pragma unroll
                        for (int u = 0; u < 8; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[e] += (double)__uint_as_float(t[u][e]);
                    }
                    for (; w < p.ncw; w += segs) {
                        const bf_u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(rP, v * 16, w * 2 * C * 4, BF_SC1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] += (double)__uint_as_float(t[e]);
                    }
                }
                __syncthreads();                                     // (sm: the sums of phase 1 / the previous sweep are consumed)
                if (seg < segs && segs > 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) red[(seg * nv + (v - vb)) * 4 + e] = acc[e];
                }
                __syncthreads();
                if (seg == 0) {
                    if (segs > 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            double q = 0.0;
                            for (int k = 0; k < segs; ++k) q += red[(k * nv + (v - vb)) * 4 + e];
                            acc[e] = q;
                        }
                    }
                    // value index 4v + e of the [2C] row: channel (4v + e) % C, sum of dy (first half) or of dy * xhat; kept as floats for
                    // the coefficient step only when both halves are not in this thread: park them in the partial row 0 (device-coherent)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int idx = 4 * v + e;
                        // doubles in the tail of the row buffer would need another region; the two sums of a channel meet below via LDS
                        ((double*)p.red2)[(long long)g * 2 * C + idx] = acc[e];
                    }
                }
                __syncthreads();
            }
            __threadfence_block();
            bf_stores_done();
            __syncthreads();
            for (int c = tid; c < C; c += 256) {                     // fb_bn_bwd_finalize: dx = scale * (dy - s1 / M - xhat * s2 / M) = c_dy dy + c_x x + c_0
                const double a = ((const double*)p.red2)[(long long)g * 2 * C + c], b = ((const double*)p.red2)[(long long)g * 2 * C + C + c];
                p.dbeta[(long long)g * p.gstride + c] = (float)a;
                p.dgamma[(long long)g * p.gstride + c] = (float)b;
                const double sc = p.scale[(long long)g * C + c], isd = p.invstd[(long long)g * C + c], mud = p.mean_tab[(long long)g * p.ch_total + p.ch_off + c];
                const double cx = -sc * isd * b * p.inv_count;
                float* o = p.coef + ((long long)g * C + c) * 3;
                bf_st_agent(o, (float)sc); bf_st_agent(o + 1, (float)cx); bf_st_agent(o + 2, (float)(-sc * a * p.inv_count - cx * mud));
            }
            bf_stores_done();
            __syncthreads();
            if (tid < 32) __hip_atomic_store(sy + 32 + tid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            wait_flag(sy + 32 + (wi % 32));
        }

        stamp(3);                                                    // coefficients available (the reducer: written)
        if (tr && tid == 0) tr[5] = last_flag;
        // ---- phase 2: dx from the registers ----------------------------------------------------------------------------------------
        const __amdgpu_buffer_rsrc_t rO = bf_rsrc(p.dx + slice, SLICE_B);
        const __amdgpu_buffer_rsrc_t rY = bf_rsrc(p.dy_out ? p.dy_out + slice : p.dx, p.dy_out ? SLICE_B : 0);
        float cf[3 * V];
        {   // 3V coefficients of this thread's channels: device-coherent 16-byte loads, all in flight together (relaxed atomic loads go one by one)
            const __amdgpu_buffer_rsrc_t rC = bf_rsrc(p.coef + ((long long)g * C + c0) * 3, 3 * V * 4);
            bf_u32x4_t t[3 * V / 4];
#pragma unroll
            for (int q = 0; q < 3 * V / 4; ++q) t[q] = __builtin_amdgcn_raw_buffer_load_b128(rC, 0, q * 16, BF_SC1);
#pragma unroll
            for (int q = 0; q < 3 * V / 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) cf[4 * q + e] = __uint_as_float(t[q][e]);
        }
#pragma unroll
        for (int j = 0; j < BF_NP; ++j) {
            float d[V], xv[V], o[V], dyv[V];
            ET<T>::unpack(dr[j], d); ET<T>::unpack(xr[j], xv);
            const unsigned m = mk[j >> 2] >> (8 * (j & 3));
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float dy = ((m >> k) & 1u) ? d[k] : 0.f;
                dyv[k] = dy;
                o[k] = fb_bn_dx(cf[3 * k], cf[3 * k + 1], cf[3 * k + 2], dy, xv[k]);
            }
            bf_st(rO, tid * 16, slot(j) * 4096, ET<T>::pack(o));
            if (p.dy_out) bf_st(rY, tid * 16, slot(j) * 4096, ET<T>::pack(dyv));
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(4);                                                    // stores issued
        __syncthreads();                                             // sm / last_flag are reused by the next chunk
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
static inline int bf_vec(int dtype) { return dtype == FB_F32 ? 4 : 8; }

// workgroups per statistics group (0: this shape is not for the fused kernel)
static int bf_cluster(int64_t n_pixels, int32_t C, int64_t ppg, int32_t dtype) {
    static const bool disabled = getenv("FB_DISABLE_BN_BWD_FUSED") != nullptr;
    if (disabled || n_pixels <= 0 || ppg <= 0 || n_pixels % ppg != 0) return 0;
    const int V = bf_vec(dtype);
    if (C % V != 0) return 0;
    const int cvec = C / V;
    if (cvec > 256 || 256 % cvec != 0) return 0;
    const long long vpg = ppg * cvec;                              // vectors per group
    if (vpg % (BF_NP * 256) != 0) return 0;
    const long long ncw = vpg / (BF_NP * 256);
    const int slots = BF_WGCU * fb_persistent_cus();
    if (ncw > slots) return 0;                                      // the cluster must be resident as a whole
    if (n_pixels / ppg > (1 << 20)) return 0;
    return (int)ncw;
}

static long long* g_bnf_trace = nullptr;
extern "C" void fb_bn_bwd_fused_trace(long long* buf) { g_bnf_trace = buf; }      // development hook (tools/bn_bwd_microbench.py); not part of the ABI header

extern "C" int32_t fb_bn_bwd_fused_supported(int64_t n_pixels, int32_t C, int64_t pixels_per_group, int32_t dtype) {
    return bf_cluster(n_pixels, C, pixels_per_group, dtype) > 0 ? 1 : 0;
}
// floats of `partial` and ints of `sync` the entry point needs
extern "C" int64_t fb_ws_bn_bwd_fused_floats(int64_t n_pixels, int32_t C, int64_t pixels_per_group, int32_t dtype) {
    const int ncw = bf_cluster(n_pixels, C, pixels_per_group, dtype);
    return ncw ? (n_pixels / pixels_per_group) * ((int64_t)ncw * 2 * C + 4 * C) : 0;       // partial rows + the final sums in double
}
extern "C" int64_t fb_ws_bn_bwd_fused_ints(int64_t n_groups) { return n_groups * BF_SYNC + 1; }

extern "C" int fb_bn_bwd_fused(const void* dout, const void* mask, const void* x, const float* mean_tab, const float* invstd, const float* scale,
                               int32_t ch_total, int32_t ch_off, float* dgamma, float* dbeta, int64_t grad_group_stride, float* coef, void* dx,
                               void* dy_out, int64_t n_pixels, int32_t C, int64_t pixels_per_group, double count, int32_t dtype, float* partial,
                               int32_t* sync, void* stream) {
    if (!dout || !x || !mean_tab || !invstd || !scale || !dgamma || !dbeta || !coef || !dx || !partial || !sync) FB_FAIL(FB_ERR_ARG, "fb_bn_bwd_fused: null pointer");
    const int ncw = bf_cluster(n_pixels, C, pixels_per_group, dtype);
    if (ncw == 0) FB_FAIL(FB_ERR_UNSUPPORTED, "fb_bn_bwd_fused: %lld pixels in groups of %lld x %d channels is not for this kernel (fb_bn_bwd_fused_supported)",
                          (long long)n_pixels, (long long)pixels_per_group, C);
    const int n_groups = (int)(n_pixels / pixels_per_group), slots = BF_WGCU * fb_persistent_cus();
    BnBwdFusedParams p;
    p.dout = (const uint4*)dout; p.mask = (const unsigned char*)mask; p.x = (const uint4*)x;
    p.mean_tab = mean_tab; p.invstd = invstd; p.scale = scale; p.dgamma = dgamma; p.dbeta = dbeta; p.coef = coef;
    p.dx = (uint4*)dx; p.dy_out = (uint4*)dy_out; p.partial = partial; p.sync = sync;
    p.red2 = (double*)(partial + (size_t)(n_pixels / pixels_per_group) * ncw * 2 * C);
    p.gstride = grad_group_stride; p.vpg = pixels_per_group * (C / bf_vec(dtype)); p.inv_count = 1.0 / count;
    p.n_groups = n_groups; p.C = C; p.ch_total = ch_total; p.ch_off = ch_off; p.ncw = ncw;
    p.n_clusters = slots / ncw < n_groups ? slots / ncw : n_groups;
    p.trace = g_bnf_trace;
    static const int poll = fb_getenv_experimental("FB_BNF_POLL") ? atoi(fb_getenv_experimental("FB_BNF_POLL")) : 2;
    p.poll = poll != 0 ? poll : 1;
    // arrival counters and flags start at zero; the error word is sticky (the caller reads and clears it: fb_bn_bwd_fused_error)
    if (hipMemsetAsync(sync, 0, sizeof(int32_t) * (size_t)n_groups * BF_SYNC, (hipStream_t)stream) != hipSuccess)
        FB_FAIL(FB_ERR_LAUNCH, "fb_bn_bwd_fused: hipMemsetAsync failed");
    const int32_t info[FB_PROF_INFO] = {(int32_t)(n_pixels / 128), C, (int32_t)(pixels_per_group / 128), dtype, 0, mask ? 1 : 0, dy_out ? 1 : 0, 0, 0, 0, 0};
    const int prof = fb_prof_begin(FB_PROF_BN_BWD_FUSED, (hipStream_t)stream, info);
    const dim3 grid((unsigned)(p.n_clusters * ncw));
    if (dtype == FB_F32) hipLaunchKernelGGL((bn_bwd_fused_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((bn_bwd_fused_kernel<bf16_tag>), grid, dim3(256), 0, (hipStream_t)stream, p);
    fb_prof_end(prof, (hipStream_t)stream);
    FB_CHECK_LAUNCH("fb_bn_bwd_fused");
    return FB_OK;
}
