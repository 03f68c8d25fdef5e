// 1x1 convolutions with a short K (64 / 128 / 256 input channels) as streaming kernels: the downsample shortcuts of ResNet-18/34
// (resnets.py:150) and the channel-expanding / -reducing convolutions of the Bottleneck blocks (resnets.py:289-291), forward and input gradient.
//
// These are GEMMs [pixels x K] . [K x Cd] whose K is 1-4 steps of the implicit-GEMM tile kernel: a tile there lives mostly in its
// prologue (first-load latency), its per-K-step barriers and its epilogue (64->128 @16x16: 138 TF/s and 3.2 TB/s of a possible 6; 256->1024
// @14x14: 367 TF/s, 1.8 TB/s -- a third of EITHER ceiling).  At 50-200 FLOP per byte they are HBM-bound, so the kernel is built like the
// streaming stem kernel (conv1x1_k32.hip), not like a GEMM:
//   * the filter never touches LDS: a wave owns CW = 16 | 32 output channels and keeps their K x CW weights in registers for its whole
//     life as the A operands of its MFMAs (K = 256, CW = 32: 64 VGPRs); 8 waves = 128 | 256 channels per workgroup
//   * pixels stream through LDS in sub-tiles of 16 KiB-of-K (64 px x 256 ch ... 256 px x 64 ch, 32 KiB) by `buffer_load ... lds`, double
//     buffered, in the 128-byte-row XOR-swizzled image of the implicit GEMM; every wave reads all pixel fragments (LDS reads at half the
//     MFMA time), ONE barrier per sub-tile
//   * persistent workgroups walk 128-pixel blocks; the stores of sub-tile t-1 are issued after the loads of sub-tile t+1 have been
//     requested and before the MFMAs of sub-tile t, so that the `s_waitcnt vmcnt(0)` that guards the next LDS buffer never waits for a
//     store that has just been issued; BatchNorm partial sums (forward) by DPP row sums straight to global (a wave owns its channels)
//   * workgroups that share a pixel range (different channel tiles) are neighbours on one XCD and share the pixel rows in its L2
#include "common.h"
#include "conv_params.h"

#include <type_traits>

namespace {
typedef __attribute__((ext_vector_type(4))) unsigned s1_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned s1_u32x2;
template <int N> __device__ __forceinline__ void s1_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void s1_wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int OFF> __device__ __forceinline__ uint4 s1_lds_read16(unsigned byte_addr) {
    s1_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
template <int I, int N, typename F> __device__ __forceinline__ void s1_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); s1_static_for<I + 1, N>(f); }
}
__device__ __forceinline__ int s1_xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, xcd = b & 7, slot = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// 16-byte buffer load through inline asm: invisible to hipcc's wait bookkeeping (it neither waits for it nor drains the LDS-DMA queue for it);
// the consumer waits with s1_wait_addend4 below.  FIRST: the descriptor's words may come straight from v_readfirstlane (5 wait states).
template <bool FIRST> __device__ __forceinline__ void s1_addend_load(s1_u32x4& dst, unsigned voff, s1_u32x4 rs) {
    if constexpr (FIRST) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rs) : "memory");
    else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rs) : "memory");
}
// counted wait that names the four destinations it releases (no consumer of them can be scheduled above it)
template <int N> __device__ __forceinline__ void s1_wait_addend4(s1_u32x4& a, s1_u32x4& b, s1_u32x4& c, s1_u32x4& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

struct S1Params {
    const char* src; const char* wgt; char* dst; const char* addend; float* stat;
    long long M; int Cd; int n_blocks; int n_co; int n_workers; int n_mblocks;
};
}  // namespace

// K input channels, CW output channels per wave.  One "unit" = 128 pixels (a statistics block) or one sub-tile of PXT pixels, whichever is larger.
// NWC waves share the channels of the workgroup (NWC x CW of them), the other factor of the 8 waves shares the pixels of a sub-tile.
// NW waves per workgroup (4: two workgroups per CU, one computes while the other waits for its loads and store acknowledgements).
// ADD (input gradient with a same-shape addend, CW = 32): the addend fragments of a sub-tile are requested as 16-byte loads in the layout of the
// STORES (8 consecutive channels per lane, `v_permlane16_swap` brings them back to the accumulator layout) right after the sub-tile's other memory
// operations have been issued, and waited for with a counted `vmcnt` in the epilogue -- their latency runs under the sub-tile's MFMAs.  (As 8-byte
// loads issued where they were consumed, every epilogue drained the whole memory queue first -- the next sub-tile's LDS-DMA and the late stores:
// hipcc waits `vmcnt(0)` for an ordinary load beside LDS-DMA -- and the wave sat out the load's full latency: 333 us against 186 without addend
// on 256 -> 1024 @14x14, 1024 images.)
template <int K, int CW, int NWC, int NW, bool ADD = false>
__global__ __launch_bounds__(NW * 64) void conv1x1_stream_kernel(const S1Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NS = K / 64;                       // 128-byte channel slices per pixel row
    constexpr int PXT = 16384 / K;                   // pixels per sub-tile: 32 KiB of LDS
    constexpr int UNIT = PXT > 128 ? PXT : 128;      // pixels per loop iteration
    constexpr int NSUB = UNIT / PXT;                 // sub-tiles per iteration (1 or 2)
    constexpr int NWP = NW / NWC;                    // pixel shares of a sub-tile
    constexpr int FI = CW / 16, FJ = PXT / 16 / NWP, KK = K / 32;      // FJ: pixel fragments per wave and sub-tile
    static_assert(FJ % 4 == 0 && (NWP == 1 || FJ % 8 == 0), "a wave's pixel share must hold whole 128-pixel statistics blocks");
    constexpr int TILE = PXT * K * 2;                // bytes
    constexpr int NDMA = (PXT / 8) * NS / NW;        // LDS-DMA instructions per wave and sub-tile (32 per sub-tile)
    static_assert(NDMA * NW == 32 && TILE == 32768, "sub-tile geometry");
    __shared__ __attribute__((aligned(16))) char lds[2 * TILE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int L = s1_xcd_remap(blockIdx.x, gridDim.x);
    const int co_blk = __builtin_amdgcn_readfirstlane(L % p.n_co), worker = __builtin_amdgcn_readfirstlane(L / p.n_co);
    const int wc = wave % NWC, wp = wave / NWC;
    const int co0 = co_blk * (NWC * CW) + wc * CW;
    const int px0 = wp * FJ * 16;                    // first pixel of this wave's share inside a sub-tile

    // the wave's filter slice: wgt [Cd][K] bf16, fragment (f, kk) = channels co0 + 16 f + col, inputs 32 kk + 8 g .. + 7
    uint4 wf[FI][KK];
#pragma unroll
    for (int f = 0; f < FI; ++f)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wf[f][kk] = *(const uint4*)(p.wgt + ((long long)(co0 + f * 16 + col) * K + kk * 32 + g * 8) * 2);

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)(p.M * K * 2), 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    // LDS-DMA: instruction q = 4 wave + i moves 8 rows of one slice; lane -> row (lane >> 3), logical chunk (lane & 7) ^ (row & 7)
    const unsigned dma_lane = (unsigned)((lane >> 3) * K * 2 + (((lane & 7) ^ (lane >> 3)) * 16));
    auto issue = [&](long long m0, int stage) {      // m0: first pixel of the sub-tile
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int qi = wave * NDMA + i, rg = qi / NS, s = qi % NS;
            const long long row0 = m0 + rg * 8;
            // rows beyond M: the buffer's range check returns zeros (their accumulators stay 0, their stores are predicated)
            const unsigned voff = (unsigned)(row0 * K * 2) + dma_lane;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + stage * TILE + (s * PXT + rg * 8) * 128), 16,
                                                     voff, s * 128, 0, 0);
        }
    };
    // fragment reads: pixel row j*16 + col of slice kk >> 1, logical chunk g + 4 (kk & 1)
    const unsigned rd0 = lds0 + (px0 + col) * 128 + ((g ^ (col & 7)) * 16), rd1 = lds0 + (px0 + col) * 128 + (((g + 4) ^ (col & 7)) * 16);

    unsigned pk[NSUB][FI][FJ][2];                    // packed bf16 outputs of the previous unit (stores are issued one unit late)
    long long pk_m0 = -1;
    float ssum[FI][4], ssq[FI][4];
#pragma unroll
    for (int f = 0; f < FI; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[f][r] = 0.f; ssq[f][r] = 0.f; }

    auto store_unit = [&](long long m0u) {
#pragma unroll
        for (int sb = 0; sb < NSUB; ++sb)
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                const long long m = m0u + sb * PXT + px0 + j * 16 + col;
                char* row = p.dst + m * p.Cd * 2;
                if constexpr (FI == 2) {             // fragment pair: 8 consecutive channels per lane after the row swap, one 16-byte store
                    const s1_u32x2 lo = __builtin_amdgcn_permlane16_swap(pk[sb][0][j][0], pk[sb][1][j][0], false, false);
                    const s1_u32x2 hi = __builtin_amdgcn_permlane16_swap(pk[sb][0][j][1], pk[sb][1][j][1], false, false);
                    const int co = co0 + (g & 1) * 16 + (g >> 1) * 8;
                    if (m < p.M) __builtin_nontemporal_store((s1_u32x4){lo[0], hi[0], lo[1], hi[1]}, (s1_u32x4*)(row + co * 2));
                } else {
                    if (m < p.M) __builtin_nontemporal_store((s1_u32x2){pk[sb][0][j][0], pk[sb][0][j][1]}, (s1_u32x2*)(row + (co0 + g * 4) * 2));
                }
            }
    };

    constexpr int JG = 4;                             // pixel fragments per accumulator group (64 pixels)
    const long long n_units = (p.M + UNIT - 1) / UNIT;
    static_assert(!ADD || FI == 2, "the addend path pairs two 16-channel fragments per 16-byte load");
    s1_u32x4 ad[ADD ? FJ : 1];                        // addend of this sub-tile, store layout: channels co0 + {0, 16, 8, 24}[g] .. + 7 of pixel j * 16 + col
    const unsigned ad_lane = (unsigned)(((px0 + col) * p.Cd + co0 + (g & 1) * 16 + (g >> 1) * 8) * 2);
    long long unit = worker;
    if (unit < n_units) issue(unit * UNIT, 0);
    int stage = 0;
    for (; unit < n_units; unit += p.n_workers) {
        const long long m0u = unit * UNIT;
#pragma unroll
        for (int sb = 0; sb < NSUB; ++sb) {
            s1_wait_vmcnt<0>();                       // this wave's share of the sub-tile has landed (and the stores of the previous unit)
            __builtin_amdgcn_s_barrier();            // ... everybody's share; everybody has finished reading the other buffer
            // request the next sub-tile into the other buffer
            if (sb + 1 < NSUB) issue(m0u + (sb + 1) * PXT, stage ^ 1);
            else if (unit + p.n_workers < n_units) issue((unit + p.n_workers) * UNIT, stage ^ 1);
            if (sb == 0 && pk_m0 >= 0) { store_unit(pk_m0); pk_m0 = -1; }     // previous unit's outputs leave under this unit's MFMAs
            if constexpr (ADD) {
                // a descriptor per unit (rows past M read zeros; 32-bit offsets whatever the tensor's size); asm loads: hipcc must neither count nor drain them
                const long long left = p.M - m0u;
                const unsigned long long base = (unsigned long long)(p.addend + m0u * p.Cd * 2);
                const s1_u32x4 rs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base),
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffffu)),
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)((left < UNIT ? left : UNIT) * p.Cd * 2)), 0x00020000u};
                s1_static_for<0, FJ>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const unsigned voff = ad_lane + (unsigned)((sb * PXT + j * 16) * p.Cd * 2);
                    s1_addend_load<j == 0>(ad[j], voff, rs);
                });
            }
            const unsigned r0 = rd0 + stage * TILE, r1 = rd1 + stage * TILE;
            s1_static_for<0, FJ / JG>([&](auto jgc) {
                constexpr int jg = decltype(jgc)::value;
                f32x4_t acc[FI][JG];
#pragma unroll
                for (int f = 0; f < FI; ++f)
#pragma unroll
                    for (int j = 0; j < JG; ++j) acc[f][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                uint4 bf[2][JG];
                s1_static_for<0, JG>([&](auto jc) { constexpr int j = decltype(jc)::value; bf[0][j] = s1_lds_read16<(jg * JG + j) * 2048>(r0); });
                s1_static_for<0, KK>([&](auto kkc) {
                    constexpr int kk = decltype(kkc)::value;
                    if constexpr (kk + 1 < KK) {      // fragments of the next K-step are requested before the MFMAs of this one
                        s1_static_for<0, JG>([&](auto jc) {
                            constexpr int j = decltype(jc)::value;
                            bf[(kk + 1) & 1][j] = s1_lds_read16<((kk + 1) >> 1) * PXT * 128 + (jg * JG + j) * 2048>(((kk + 1) & 1) ? r1 : r0);
                        });
                        s1_wait_lgkmcnt<JG>();
                    } else {
                        s1_wait_lgkmcnt<0>();
                    }
#pragma unroll
                    for (int f = 0; f < FI; ++f)
#pragma unroll
                        for (int j = 0; j < JG; ++j)
                            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[f][kk]), __builtin_bit_cast(bf16x8_t, bf[kk & 1][j]),
                                                                                acc[f][j], 0, 0, 0);
                });
                // epilogue of the group: (+ addend) -> packed bf16 kept for the late store; statistics per 128-pixel block
                if constexpr (ADD) {                  // this group's addend fragments have landed: only the later groups' loads may still be in flight
                    constexpr int later = FJ - (jg + 1) * JG;
                    static_assert(JG == 4, "the wait below names four fragments");
                    s1_wait_addend4<later>(ad[ADD ? jg * JG : 0], ad[ADD ? jg * JG + 1 : 0], ad[ADD ? jg * JG + 2 : 0], ad[ADD ? jg * JG + 3 : 0]);
                }
#pragma unroll
                for (int jj = 0; jj < JG; ++jj) {
                    constexpr int jbase = jg * JG;
                    const int j = jbase + jj;
                    const int pix_in_unit = sb * PXT + px0 + j * 16;
                    const long long m = m0u + pix_in_unit + col;
                    unsigned aw[2][2] = {{0u, 0u}, {0u, 0u}};          // addend words in the accumulator layout: [fragment][channel pair]
                    if constexpr (ADD) {
                        const s1_u32x4 a = ad[ADD ? j : 0];
                        const s1_u32x2 lo = __builtin_amdgcn_permlane16_swap(a[0], a[2], false, false);      // the inverse of the store's exchange (an involution)
                        const s1_u32x2 hi = __builtin_amdgcn_permlane16_swap(a[1], a[3], false, false);
                        aw[0][0] = lo[0]; aw[1][0] = lo[1]; aw[0][1] = hi[0]; aw[1][1] = hi[1];
                    }
#pragma unroll
                    for (int f = 0; f < FI; ++f) {
                        float v[4] = {acc[f][jj][0], acc[f][jj][1], acc[f][jj][2], acc[f][jj][3]};
                        if constexpr (ADD) {
                            v[0] += __uint_as_float(aw[f & 1][0] << 16); v[1] += __uint_as_float(aw[f & 1][0] & 0xffff0000u);
                            v[2] += __uint_as_float(aw[f & 1][1] << 16); v[3] += __uint_as_float(aw[f & 1][1] & 0xffff0000u);
                        } else if (p.addend != nullptr && m < p.M) {
                            const uint2 a = *(const uint2*)(p.addend + (m * p.Cd + co0 + f * 16 + g * 4) * 2);
                            v[0] += __uint_as_float(a.x << 16); v[1] += __uint_as_float(a.x & 0xffff0000u);
                            v[2] += __uint_as_float(a.y << 16); v[3] += __uint_as_float(a.y & 0xffff0000u);
                        }
                        pk[sb][f][j][0] = pack_bf16x2(v[0], v[1]); pk[sb][f][j][1] = pack_bf16x2(v[2], v[3]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ssum[f][r] += v[r]; ssq[f][r] += v[r] * v[r]; }
                    }
                    if (((pix_in_unit + 16) & 127) == 0) {      // last fragment of a 128-pixel statistics block
                        if (p.stat != nullptr) {
                            const long long blk = (m0u + pix_in_unit) >> 7;
#pragma unroll
                            for (int f = 0; f < FI; ++f)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float a = row16_sum(ssum[f][r]), b = row16_sum(ssq[f][r]);
                                    if (col == 0 && blk < p.n_mblocks) {
                                        const int co = co0 + f * 16 + g * 4 + r;
                                        p.stat[blk * p.Cd + co] = a;
                                        p.stat[((long long)p.n_mblocks + blk) * p.Cd + co] = b;
                                    }
                                }
                        }
#pragma unroll
                        for (int f = 0; f < FI; ++f)
#pragma unroll
                            for (int r = 0; r < 4; ++r) { ssum[f][r] = 0.f; ssq[f][r] = 0.f; }
                    }
                }
            });
            stage ^= 1;
        }
        pk_m0 = m0u;
    }
    if (pk_m0 >= 0) store_unit(pk_m0);
#endif
}

template <int K, int CW, int NWC, int NW, bool ADD = false> static void s1_launch(const S1Params& p, int grid, hipStream_t st) {
    hipLaunchKernelGGL((conv1x1_stream_kernel<K, CW, NWC, NW, ADD>), dim3(grid), dim3(NW * 64), 0, st, p);
}

// returns 1 if the kernel handled the call: bf16 1x1 convolution (forward or input gradient), 64 / 128 / 256 input channels, output channels a
// multiple of 128, one shared weight set, optional same-shape addend (mode 1), optional BatchNorm partial sums (mode 0)
int fb_try_conv1x1_stream(const fb_conv_args* a, hipStream_t st) {
    static const bool disabled = getenv("FB_DISABLE_CONV1X1_STREAM") != nullptr;
    if (disabled) return 0;
    if (a->R != 1 || a->S != 1 || a->stride != 1 || a->pad != 0 || a->dtype != FB_BF16) return 0;
    if (a->Hs != a->Hd || a->Ws != a->Wd) return 0;
    if (a->Cs != 64 && a->Cs != 128 && a->Cs != 256) return 0;
    if (a->Cd % 128 != 0) return 0;
    if (a->wset_stride != 0 && a->imgs_per_wset > 0 && a->imgs_per_wset < a->n_img) return 0;
    if (a->addend_mask || a->bst_x) return 0;
    if (a->mode == 0 && a->addend) return 0;
    if (a->mode == 1 && (a->stat_partial || (a->addend && a->addend_mode != 1))) return 0;
    const long long M = (long long)a->n_img * a->Hd * a->Wd;
    if (M * a->Cs * 2 >= (1LL << 31) || M * a->Cd * 2 >= (1LL << 40)) return 0;
    // 4-wave workgroups, two per CU (64 KiB of LDS each): a wave owns 32 channels; K >= 128: 4 channel waves = 128 channels per workgroup, every
    // wave reads the whole sub-tile; K = 64 (sub-tiles of 256 pixels, output-dominated traffic): 2 channel waves x 2 pixel halves
    // measured (tools/conv_microbench.py, profiles/r3_notes.md): 8-wave workgroups (one per CU) are 5-10 % faster for the forward shapes, 4-wave
    // ones (two per CU: one computes while the other waits for its loads and store acknowledgements) for most input gradients
    static const int nw_env = fb_getenv_experimental("FB_C1S_NW") ? atoi(fb_getenv_experimental("FB_C1S_NW")) : 0;
    const bool add_asm = !(getenv("FB_C1S_ADD_ASM") && atoi(getenv("FB_C1S_ADD_ASM")) == 0);      // A/B (read per call): the addend by in-place 8-byte loads
    const bool add = a->addend != nullptr && add_asm;
    const int nw = add ? 4 : (nw_env ? nw_env : (a->mode == 0 ? 8 : 4));
    const int CW = (nw == 8 && a->Cs != 64 && a->Cd % 256 != 0) ? 16 : 32;
    const int NWC = nw == 8 ? (a->Cs == 64 ? 4 : 8) : (a->Cs == 64 ? 2 : 4);
    const int pxt = 16384 / a->Cs, unit = pxt > 128 ? pxt : 128;
    S1Params p;
    p.src = (const char*)a->src; p.wgt = (const char*)a->wgt; p.dst = (char*)a->dst; p.addend = (const char*)a->addend; p.stat = a->stat_partial;
    p.M = M; p.Cd = a->Cd;
    p.n_co = a->Cd / (NWC * CW);
    p.n_mblocks = (int)((M + 127) / 128);
    const long long n_units = (M + unit - 1) / unit;
    const int n_cu = fb_persistent_cus();
    long long workers = ((nw == 8 ? 1LL : 2LL) * n_cu) / p.n_co;
    if (workers < 1) workers = 1;
    if (workers > n_units) workers = n_units;
    p.n_workers = (int)workers;
    p.n_blocks = (int)n_units;
    const int grid = p.n_workers * p.n_co;
    if (nw == 8) {
        if (a->Cs == 64) s1_launch<64, 32, 4, 8>(p, grid, st);
        else if (a->Cs == 128) { if (CW == 32) s1_launch<128, 32, 8, 8>(p, grid, st); else s1_launch<128, 16, 8, 8>(p, grid, st); }
        else { if (CW == 32) s1_launch<256, 32, 8, 8>(p, grid, st); else s1_launch<256, 16, 8, 8>(p, grid, st); }
    } else if (add) {
        if (a->Cs == 64) s1_launch<64, 32, 2, 4, true>(p, grid, st);
        else if (a->Cs == 128) s1_launch<128, 32, 4, 4, true>(p, grid, st);
        else s1_launch<256, 32, 4, 4, true>(p, grid, st);
    } else {
        if (a->Cs == 64) s1_launch<64, 32, 2, 4>(p, grid, st);
        else if (a->Cs == 128) s1_launch<128, 32, 4, 4>(p, grid, st);
        else s1_launch<256, 32, 4, 4>(p, grid, st);
    }
    return 1;
}
