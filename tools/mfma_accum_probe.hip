// How exact is an fp32 dot product on the bf16 matrix pipe?  One wave computes a 16x16 block of C = A^T B with K = 4608 (the deepest
// reduction of ResNet-18: 3x3x512) from random fp32 operands in four ways and the host compares with float64:
//   exact : v_mfma_f32_16x16x4_f32 chain (the engine's FB_F32_EXACT=1 path)
//   x6    : three-way bf16 split, six v_mfma_f32_16x16x32_bf16 per 32-deep step accumulated straight into the running accumulator
//   x6z   : the six MFMAs of a step start from a ZERO accumulator; the step's partial sum is added to the running one with v_add_f32
//   x9z   : all nine piece products, zero-started
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/mfma_accum_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

__device__ __forceinline__ unsigned pack2(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t)); }
struct split3 { bf16x8_t h, m, l; };
__device__ __forceinline__ split3 split8(const float* x) {
    unsigned hp[4], mp[4], lp[4];
    for (int q = 0; q < 4; ++q) {
        const float x0 = x[2 * q], x1 = x[2 * q + 1];
        hp[q] = pack2(x0, x1);
        const float r0 = x0 - __uint_as_float(hp[q] << 16), r1 = x1 - __uint_as_float(hp[q] & 0xffff0000u);
        mp[q] = pack2(r0, r1);
        const float s0 = r0 - __uint_as_float(mp[q] << 16), s1 = r1 - __uint_as_float(mp[q] & 0xffff0000u);
        lp[q] = pack2(s0, s1);
    }
    split3 o;
    o.h = __builtin_bit_cast(bf16x8_t, make_uint4(hp[0], hp[1], hp[2], hp[3]));
    o.m = __builtin_bit_cast(bf16x8_t, make_uint4(mp[0], mp[1], mp[2], mp[3]));
    o.l = __builtin_bit_cast(bf16x8_t, make_uint4(lp[0], lp[1], lp[2], lp[3]));
    return o;
}
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// A: [K][16] (row k, column m), B: [K][16]; C[m][n] = sum_k A[k][m] B[k][n].  mode 0 exact, 1 x6, 2 x6z, 3 x9z
__global__ void probe(const float* A, const float* B, float* C, int K, int mode) {
    const int lane = threadIdx.x, col = lane & 15, g = lane >> 4;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
        float a[8], b[8];
        for (int e = 0; e < 8; ++e) { a[e] = A[(k0 + 8 * g + e) * 16 + col]; b[e] = B[(k0 + 8 * g + e) * 16 + col]; }
        if (mode == 0) {
            for (int s = 0; s < 8; ++s) {        // slab (k0 + 4s .. +3): lane group g supplies k = k0 + 4s + g
                const float av = A[(k0 + 4 * s + g) * 16 + col], bv = B[(k0 + 4 * s + g) * 16 + col];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
            }
        } else {
            const split3 sa = split8(a), sb = split8(b);
            if (mode == 1) {
                acc = MF(sa.l, sb.h, acc); acc = MF(sa.h, sb.l, acc); acc = MF(sa.m, sb.m, acc);
                acc = MF(sa.m, sb.h, acc); acc = MF(sa.h, sb.m, acc); acc = MF(sa.h, sb.h, acc);
            } else {
                f32x4_t t = {0.f, 0.f, 0.f, 0.f};
                if (mode == 3) { t = MF(sa.l, sb.l, t); t = MF(sa.l, sb.m, t); t = MF(sa.m, sb.l, t); }
                t = MF(sa.l, sb.h, t); t = MF(sa.h, sb.l, t); t = MF(sa.m, sb.m, t);
                t = MF(sa.m, sb.h, t); t = MF(sa.h, sb.m, t); t = MF(sa.h, sb.h, t);
                acc[0] += t[0]; acc[1] += t[1]; acc[2] += t[2]; acc[3] += t[3];
            }
        }
    }
    // D layout of 16x16 MFMA: lane holds rows 4g..4g+3 of column col
    for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + col] = acc[r];
}

int main() {
    const int K = 4608;
    std::vector<float> A(K * 16), B(K * 16);
    srand(7);
    auto rnd = [] { return (float)((rand() / (double)RAND_MAX - 0.5) * 2.0); };
    for (int trial = 0; trial < 2; ++trial) {
        // trial 0: zero-mean operands (cancelling sums, like gradients); trial 1: positive operands (post-ReLU activations x |w|)
        for (auto& v : A) v = trial ? fabsf(rnd()) : rnd();
        for (auto& v : B) v = trial ? fabsf(rnd()) : rnd();
        std::vector<double> ref(256, 0.0);
        for (int k = 0; k < K; ++k)
            for (int m = 0; m < 16; ++m)
                for (int n = 0; n < 16; ++n) ref[m * 16 + n] += (double)A[k * 16 + m] * (double)B[k * 16 + n];
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 256 * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        const char* names[4] = {"exact f32 MFMA", "bf16x6 direct ", "bf16x6 zero-st", "bf16x9 zero-st"};
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
            std::vector<float> C(256);
            hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
            double num = 0, den = 0, bias = 0, scale = 0;
            for (int i = 0; i < 256; ++i) { num += (C[i] - ref[i]) * (C[i] - ref[i]); den += ref[i] * ref[i]; bias += C[i] - ref[i]; scale += fabs(ref[i]); }
            // also in units of the typical magnitude sqrt(sum a^2 b^2)
            printf("trial %d  %s  rel L2 err %.3e   mean signed err / mean |C| %.3e\n", trial, names[mode], sqrt(num / den), bias / scale);
        }
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
