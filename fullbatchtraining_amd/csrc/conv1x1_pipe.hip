// 1x1 convolutions with 64 / 128 / 256 input channels, forward and input gradient: the streaming kernel of conv1x1_stream.hip (register-resident
// filter, pixels through LDS by `buffer_load ... lds`, persistent workgroups) with its EPILOGUE SOFTWARE-PIPELINED INTO THE NEXT GROUP'S MFMAs.
//
// What the counters said about the round-3 form (round 5, 256 -> 1024 @14x14, 1024 images, 144 us = 3.5 TB/s, HBM bytes = algorithmic):
// matrix pipe busy a third of the time, 43 % of the wave cycles stalled at issue, 3-4 vector instructions per MFMA.  Per 64-pixel group a wave
// issues 64 MFMAs (~1100 cycles of matrix pipe) and then ~200 vector instructions of epilogue (fp32 sums and sums of squares for the BatchNorm
// statistics, bf16 packing, lane-row exchange, addresses; ~1200 cycles) -- and because the two waves of a SIMD leave the per-sub-tile barrier
// together they run their MFMA phases against each other and then their epilogues against each other: MFMA time + VALU time instead of the
// larger of the two.  Here a wave keeps TWO accumulator sets (2 x 32 registers): while the MFMAs of group n run, the epilogue of group n-1 is
// worked off in slices between them (`sched_group_barrier`: one MFMA, a few VALU), and its stores leave at the end of the section.
//
// Every memory operation of the loop is issued unconditionally (rows past the tensor's end are dropped / read as zeros by the buffer range
// check on per-group descriptors), so the queue of outstanding operations at any point of the code is a compile-time constant and every wait
// is a COUNTED `s_waitcnt vmcnt(N)`: nothing ever drains the queue -- not the addend loads (16-byte loads in the layout of the stores, issued
// a whole group ahead through inline asm that hipcc neither counts nor waits for), not the stores.
//
// Geometry (as before): a wave owns 32 output channels (K x 32 weights in registers) and a 128-pixel statistics block per unit, worked off as two
// GROUPS of 64 pixels x 32 channels (8 accumulator fragments); K = 256: one group per 64-pixel sub-tile, K = 128: two per 128-pixel sub-tile,
// K = 64: the two pixel halves of a 256-pixel sub-tile go to different waves.
#include "common.h"
#include "conv_params.h"

#include <type_traits>

#ifndef P1_LATE
#define P1_LATE 1           /* stores wait for the next barrier (A/B builds: tools/build_variant.py) */
#endif
#ifndef P1_SPREAD
#define P1_SPREAD 1         /* LDS-DMA pieces issued inside the MFMA slices */
#endif

namespace {
typedef __attribute__((ext_vector_type(4))) unsigned p1_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned p1_u32x2;
typedef __attribute__((ext_vector_type(4))) float p1_f32x4;
template <int N> __device__ __forceinline__ void p1_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void p1_wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int OFF> __device__ __forceinline__ uint4 p1_lds_read16(unsigned byte_addr) {
    p1_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
template <int I, int N, typename F> __device__ __forceinline__ void p1_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); p1_static_for<I + 1, N>(f); }
}
__device__ __forceinline__ int p1_xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, xcd = b & 7, slot = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
// 16-byte buffer load through inline asm (invisible to hipcc's wait bookkeeping; the consumer waits with p1_wait_loads4)
__device__ __forceinline__ void p1_load16(p1_u32x4& dst, unsigned voff, p1_u32x4 rs) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rs) : "memory");
}
// one mask byte per lane (the ReLU bitmask of the lane's 16-byte addend vector: bit j <-> channel j), through inline asm like p1_load16
__device__ __forceinline__ void p1_load_byte(unsigned& dst, unsigned voff, p1_u32x4 rs) {
    asm volatile("buffer_load_ubyte %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rs) : "memory");
}
// the addend where its bit is set, zero elsewhere (v_bfe_i32 turns a bit into an all-ones word)
__device__ __forceinline__ p1_u32x4 p1_masked(p1_u32x4 a, unsigned bits) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned lo = (unsigned)__builtin_amdgcn_sbfe((int)bits, 2 * d, 1), hi = (unsigned)__builtin_amdgcn_sbfe((int)bits, 2 * d + 1, 1);
        a[d] &= (lo & 0x0000ffffu) | (hi & 0xffff0000u);
    }
    return a;
}
// counted wait that names the four destinations it releases (no consumer of them can be scheduled above it)
template <int N> __device__ __forceinline__ void p1_wait_loads4(p1_u32x4& a, p1_u32x4& b, p1_u32x4& c, p1_u32x4& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
// descriptor of `rows` rows of `row_bytes` bytes at `base` (rows <= 0: everything out of range: loads return zeros, stores are dropped)
__device__ __forceinline__ p1_u32x4 p1_desc(const char* base, long long rows, int max_rows, int row_bytes) {
    const unsigned long long b = (unsigned long long)base;
    const long long r = rows < 0 ? 0 : (rows > max_rows ? max_rows : rows);
    return (p1_u32x4){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), (unsigned)__builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu)),
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(r * row_bytes)), 0x00020000u};
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t p1_rsrc(const char* base, long long rows, int max_rows, int row_bytes) {
    const long long r = rows < 0 ? 0 : (rows > max_rows ? max_rows : rows);
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(r * row_bytes), 0x00020000);
}
constexpr unsigned P1_OOB = 0x80000000u;

struct P1Params {
    const char* src; const char* wgt; char* dst; const char* addend; float* stat;
    const unsigned char* addend_mask;    // MASK: ReLU bitmask of the addend as fb_bn_apply wrote it (1 byte per 16-byte vector): the addend counts only where its bit is set
    long long M; int Cd; int n_co; int n_workers; int n_mblocks;
    int exp;                             // timing experiments, only in builds with -DFB_C1P_EXPERIMENTS (WRONG results): FB_C1P_EXP & 1 = every store into the first 256 rows, & 2 = every LDS-DMA round from the first 256 rows, & 4 = full 128-byte lines per store instruction (pixel pairs)
};
#ifdef FB_C1P_EXPERIMENTS
#define P1_EXP(p) ((p).exp)
#else
#define P1_EXP(p) 0
#endif
}  // namespace

// K input channels; NWC waves share the channels of the workgroup (32 each), NW / NWC the pixels of a sub-tile; STAT: BatchNorm partial sums
// per 128-pixel block (forward); ADD: same-shape addend (input gradient of the convolution behind a residual branch); MASK: ... taken through its ReLU
// bitmask (the masked gradient d * (out > 0) of reference resnets.py:312-316 without a materialised copy: fb_conv_args.addend_mask)
template <int K, int NWC, int NW, bool STAT, bool ADD, bool MASK = false>
__global__ __launch_bounds__(NW * 64) void conv1x1_pipe_kernel(const P1Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NS = K / 64;                       // 128-byte channel slices per pixel row
    constexpr int PXT = 16384 / K;                   // pixels per sub-tile: 32 KiB of LDS
    constexpr int UNIT = PXT > 128 ? PXT : 128;      // pixels per unit: every wave works off one 128-pixel block of it
    constexpr int NSUB = UNIT / PXT;                 // sub-tiles (LDS-DMA rounds) per unit: 2 (K = 256) or 1
    constexpr int NWP = NW / NWC;                    // pixel shares of a sub-tile
    constexpr int FI = 2, JG = 4, KK = K / 32;
    constexpr int FJ = PXT / 16 / NWP, NG = FJ / JG; // groups (64 pixels) of a wave per sub-tile
    static_assert(NSUB * NG == 2 && FJ * NWP * 16 == PXT, "a wave owns 128 pixels of a unit: two groups");
    constexpr int TILE = 32768, NDMA = 32 / NW;      // LDS-DMA instructions per wave and sub-tile
    // memory operations of one group section, in issue order: [addend loads of the group] ... [statistics stores of the previous group's block] [stores of the previous group]
    static_assert(ADD || !MASK, "a mask needs an addend");
    constexpr int AD = ADD ? (MASK ? 2 * JG : JG) : 0, SS = STAT ? 4 : 0, ST = JG;
    __shared__ __attribute__((aligned(16))) char lds[2 * TILE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int L = p1_xcd_remap(blockIdx.x, gridDim.x);
    const int co_blk = __builtin_amdgcn_readfirstlane(L % p.n_co), worker = __builtin_amdgcn_readfirstlane(L / p.n_co);
    const int wc = wave % NWC, wp = wave / NWC;
    const int co0 = co_blk * (NWC * 32) + wc * 32;
    const int px0 = wp * FJ * 16;                    // first pixel of this wave's share inside a sub-tile

    // the wave's filter slice: wgt [Cd][K] bf16, fragment (f, kk) = channels co0 + 16 f + col, inputs 32 kk + 8 g .. + 7
    uint4 wf[FI][KK];
#pragma unroll
    for (int f = 0; f < FI; ++f)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wf[f][kk] = *(const uint4*)(p.wgt + ((long long)(co0 + f * 16 + col) * K + kk * 32 + g * 8) * 2);

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    // LDS-DMA: instruction q = NDMA wave + i moves 8 rows of one slice; lane -> row (lane >> 3), logical chunk (lane & 7) ^ (row & 7)
    const unsigned dma_lane = (unsigned)((lane >> 3) * K * 2 + (((lane & 7) ^ (lane >> 3)) * 16));
    const int row_b = p.Cd * 2;
    // piece i of a sub-tile's LDS-DMA round (rows past M, or a sub-tile past the end, read zeros: the descriptor covers the rows that exist)
    auto issue_piece = [&](const __amdgpu_buffer_rsrc_t rs, const int i, const int stage) {
        const int qi = wave * NDMA + i, rg = qi / NS, s = qi % NS;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + stage * TILE + (s * PXT + rg * 8) * 128), 16,
                                                 dma_lane + (unsigned)(rg * 8 * K * 2), s * 128, 0, 0);   // (the range check sees the row: voffset; the slice rides in soffset)
    };
    // fragment reads: pixel row j*16 + col of slice kk >> 1, logical chunk g + 4 (kk & 1)
    const unsigned rd0 = lds0 + (px0 + col) * 128 + ((g ^ (col & 7)) * 16), rd1 = lds0 + (px0 + col) * 128 + (((g + 4) ^ (col & 7)) * 16);
    // stores / addend loads of a group: pixel jj * 16 + col of the group, 8 channels co0 + {0, 16, 8, 24}[g] .. + 7 (the lane-row exchange below)
    unsigned voffS[JG];
#pragma unroll
    for (int jj = 0; jj < JG; ++jj) voffS[jj] = (unsigned)((jj * 16 + col) * row_b + (co0 + (g & 1) * 16 + (g >> 1) * 8) * 2);
#ifdef FB_C1P_EXPERIMENTS
    if (P1_EXP(p) & 4)                               // (timing experiment: the lanes of a pixel pair write the two halves of ONE 128-byte line)
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) voffS[jj] = (unsigned)(((jj * 16 + col) & ~1) * row_b + ((co0 / 32) * 2 + (col & 1)) * 64 + ((g & 1) * 16 + (g >> 1) * 8) * 2);
    if (P1_EXP(p) & 8)                               // (the two halves of a line from two CONSECUTIVE store instructions of the same wave)
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) voffS[jj] = (unsigned)(((jj >> 1) * 32 + col * 2) * row_b + ((co0 / 32) * 2 + (jj & 1)) * 64 + ((g & 1) * 16 + (g >> 1) * 8) * 2);
#endif
    // statistics: lanes col == 0 store 4 channels (16 bytes) of each of the two channel fragments
    const unsigned voffT = col == 0 ? (unsigned)((co0 + g * 4) * 4) : P1_OOB;

    f32x4_t accA[FI][JG], accB[FI][JG];              // group n / group n - 1 (roles alternate)
    p1_u32x4 adA[ADD ? JG : 1], adB[ADD ? JG : 1];
    unsigned mkA[MASK ? JG : 1], mkB[MASK ? JG : 1];         // MASK: the mask byte of every addend vector
    float ssum[FI][4], ssq[FI][4];

    // Forward (STAT): outputs whose epilogue has run wait for the NEXT barrier before they are stored (P1_LATE) -- right behind a barrier all waves of
    // the workgroup issue their stores together, so the two 64-byte halves of a 128-byte line (neighbouring waves) reach the L2 side by side.
    // Measured (1024 images, us; stores at the section's end / behind the barrier): forward 64 -> 256 @56x56 601 / 584, 256 -> 1024 @14x14 111 / 111;
    // input gradient 256 -> 1024 98 / 107, with addend 202 / 221: kept for the forward only.  NSUB == 2: one pending group, NSUB == 1: two.
    constexpr int NSLOT = NSUB == 2 ? 1 : 2;
    p1_u32x4 outP[NSLOT][JG];
    long long m0_out[NSLOT];
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) {
        m0_out[q] = -1;
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) outP[q][jj] = (p1_u32x4){0u, 0u, 0u, 0u};
    }
    auto store_group = [&](const p1_u32x4 (&o)[JG], const long long m0) {
        // (no group: m0 < 0 -> an empty descriptor at the tensor's base: nothing is stored)
        const __amdgpu_buffer_rsrc_t rsD = p1_rsrc(p.dst + (m0 < 0 ? 0 : ((P1_EXP(p) & 1) ? (m0 & 255) : m0)) * row_b, m0 < 0 ? 0 : p.M - m0, 64, row_b);
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) {
            __builtin_amdgcn_raw_buffer_store_b128(o[jj], rsD, voffS[jj], 0, 2);
            store_b128_guard(o[jj]);
        }
    };

    // ---- one group section: the MFMAs of group (m0u, POS) into `acc`, the epilogue of the previous group (`accp`, `adp`, first pixel m0p,
    // position 1 - POS) threaded through them.  TOP: the section opens a sub-tile (wait, barrier, pending stores, next LDS-DMA round).
    // Issue order of a section's memory operations (every count below follows from it):
    //   [TOP, LATE: stores of the pending groups] [TOP, !SPREAD: LDS-DMA round] [addend loads] [TOP, SPREAD: LDS-DMA pieces, in the slices]
    //   [block end: statistics stores] [!LATE: stores of the previous group]
    auto section = [&](auto posc, f32x4_t (&acc)[FI][JG], f32x4_t (&accp)[FI][JG], p1_u32x4 (&ad)[ADD ? JG : 1], p1_u32x4 (&adp)[ADD ? JG : 1],
                       unsigned (&mk)[MASK ? JG : 1], unsigned (&mkp)[MASK ? JG : 1],
                       const long long m0u, const long long m0p, const long long m0_next_dma, const int stage) {
        constexpr int POS = decltype(posc)::value;   // 0 / 1: first / second 64 pixels of the wave's block
        constexpr bool TOP = NSUB == 2 || POS == 0;
        constexpr int PPOS = 1 - POS;                // position of the previous group (of the previous unit if POS == 0)
        constexpr bool BLOCK_END = STAT && PPOS == 1;
        constexpr int sb = NSUB == 2 ? POS : 0, jg = NSUB == 2 ? 0 : POS;
        constexpr bool LATE = (P1_LATE != 0) && STAT, SPREAD = P1_SPREAD != 0;
        constexpr int ST_EARLY = LATE ? 0 : ST, ST_LATE = LATE ? NSLOT * ST : 0;
        constexpr int SLOT = NSUB == 2 ? 0 : PPOS;   // the pending-output slot this section's epilogue fills
        const long long m0g = m0u + sb * PXT + px0 + jg * 64;          // first pixel of this wave's group
        __amdgpu_buffer_rsrc_t rs_dma;
        if constexpr (TOP) {
            // operations younger than the last LDS-DMA piece of the round this sub-tile waits for: the rest of the section that issued it (and, NSUB == 1,
            // the whole section between)
            constexpr int ss_of_issuer = (STAT && (NSUB == 2 ? PPOS : 0) == 0) ? SS : 0;
            constexpr int younger = (SPREAD ? 0 : AD) + ss_of_issuer + ST_EARLY + (NSUB == 1 ? AD + ST_EARLY : 0);
            p1_wait_vmcnt<younger>();
            __builtin_amdgcn_s_barrier();            // everybody's share has landed; everybody has finished reading the other buffer
            if constexpr (LATE) {
#pragma unroll
                for (int q = 0; q < NSLOT; ++q) store_group(outP[q], m0_out[q]);
            }
            rs_dma = p1_rsrc(p.src + ((P1_EXP(p) & 2) ? (m0_next_dma & 255) : m0_next_dma) * (K * 2), p.M - m0_next_dma, PXT, K * 2);
            if constexpr (!SPREAD) {
#pragma unroll
                for (int i = 0; i < NDMA; ++i) issue_piece(rs_dma, i, stage ^ 1);
            }
        }
        if constexpr (ADD) {
            const p1_u32x4 rs = p1_desc(p.addend + m0g * row_b, p.M - m0g, 64, row_b);
            asm volatile("s_nop 4" ::: "memory");    // (the descriptor's words may come straight from v_readfirstlane: 5 wait states)
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) p1_load16(ad[jj], voffS[jj], rs);
            if constexpr (MASK) {                    // (a row's mask bytes sit at its byte offsets / 16; rows past the end read 0: nothing is added, nothing stored)
                const long long rows = p.M - m0g;
                const p1_u32x4 rsm = p1_desc((const char*)p.addend_mask + ((m0g * row_b) >> 4), 1, 1, (int)(((rows < 0 ? 0 : (rows > 64 ? 64 : rows)) * row_b) >> 4));
                asm volatile("s_nop 4" ::: "memory");
#pragma unroll
                for (int jj = 0; jj < JG; ++jj) p1_load_byte(mk[jj], voffS[jj] >> 4, rsm);
            }
        }
        const unsigned r0 = rd0 + stage * TILE, r1 = rd1 + stage * TILE;
        unsigned pk0[JG][2];                          // packed outputs of the previous group's first channel fragment
        p1_u32x4 out[JG];
        uint4 bf[2][JG];
        p1_static_for<0, JG>([&](auto jc) { constexpr int j = decltype(jc)::value; bf[0][j] = p1_lds_read16<(jg * JG + j) * 2048>(r0); });
        p1_static_for<0, KK>([&](auto kkc) {
            constexpr int kk = decltype(kkc)::value;
            if constexpr (kk + 1 < KK) {              // fragments of the next K-step are requested before the MFMAs of this one
                p1_static_for<0, JG>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    bf[(kk + 1) & 1][j] = p1_lds_read16<((kk + 1) >> 1) * PXT * 128 + (jg * JG + j) * 2048>(((kk + 1) & 1) ? r1 : r0);
                });
                p1_wait_lgkmcnt<JG>();
            } else {
                p1_wait_lgkmcnt<0>();
            }
#pragma unroll
            for (int f = 0; f < FI; ++f)
#pragma unroll
                for (int j = 0; j < JG; ++j) {
                    const f32x4_t c = kk == 0 ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : acc[f][j];
                    acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[f][kk]), __builtin_bit_cast(bf16x8_t, bf[kk & 1][j]), c, 0, 0, 0);
                }
            // ---- epilogue slice kk of the previous group: tasks t = (f, jj), f-major (the first fragment's statistics are final half way) ----
            constexpr int TPS = 8 / KK;               // tasks per slice
            if constexpr (ADD && kk == 0) {
                // the previous group's addend has landed.  Younger than its loads: the rest of ITS section (Q: position PPOS) and what this one has issued so far
                constexpr bool TOPQ = NSUB == 2 || PPOS == 0;
                constexpr int younger = ((TOPQ && SPREAD) ? NDMA : 0) + ((STAT && PPOS == 0) ? SS : 0) + ST_EARLY
                                        + (TOP ? ST_LATE : 0) + ((TOP && !SPREAD) ? NDMA : 0) + AD;
                p1_wait_loads4<younger>(adp[0], adp[ADD ? 1 : 0], adp[ADD ? 2 : 0], adp[ADD ? 3 : 0]);
                if constexpr (MASK) {                // (the mask bytes were requested right behind the addend: the same wait covers them)
                    asm volatile("" : "+v"(mkp[0]), "+v"(mkp[MASK ? 1 : 0]), "+v"(mkp[MASK ? 2 : 0]), "+v"(mkp[MASK ? 3 : 0]));
#pragma unroll
                    for (int jj = 0; jj < JG; ++jj) adp[jj] = p1_masked(adp[jj], mkp[MASK ? jj : 0]);
                }
            }
#pragma unroll
            for (int t = kk * TPS; t < (kk + 1) * TPS; ++t) {
                const int f = t >> 2, jj = t & 3;
                float v[4] = {accp[f][jj][0], accp[f][jj][1], accp[f][jj][2], accp[f][jj][3]};
                if constexpr (ADD) {
                    // the addend in the layout of the stores; the lane-row exchange (an involution) brings it to the accumulators' layout:
                    // word pair (a[0], a[2]) -> channels 4g, 4g+1 of fragments 0 / 1; (a[1], a[3]) -> channels 4g+2, 4g+3
                    const p1_u32x4 a = adp[ADD ? jj : 0];
                    const p1_u32x2 lo = __builtin_amdgcn_permlane16_swap(a[0], a[2], false, false);
                    const p1_u32x2 hi = __builtin_amdgcn_permlane16_swap(a[1], a[3], false, false);
                    const unsigned w0 = f ? lo[1] : lo[0], w1 = f ? hi[1] : hi[0];
                    v[0] += __uint_as_float(w0 << 16); v[1] += __uint_as_float(w0 & 0xffff0000u);
                    v[2] += __uint_as_float(w1 << 16); v[3] += __uint_as_float(w1 & 0xffff0000u);
                }
                const unsigned q0 = pack_bf16x2(v[0], v[1]), q1 = pack_bf16x2(v[2], v[3]);
                if constexpr (STAT) {
                    if (PPOS == 0 && jj == 0) {       // first fragment of a block: the sums start here
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ssum[f][r] = v[r]; ssq[f][r] = v[r] * v[r]; }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ssum[f][r] += v[r]; ssq[f][r] = fmaf(v[r], v[r], ssq[f][r]); }
                    }
                }
                if (f == 0) { pk0[jj][0] = q0; pk0[jj][1] = q1; }
                else {
                    const p1_u32x2 lo = __builtin_amdgcn_permlane16_swap(pk0[jj][0], q0, false, false);
                    const p1_u32x2 hi = __builtin_amdgcn_permlane16_swap(pk0[jj][1], q1, false, false);
                    out[jj] = (p1_u32x4){lo[0], hi[0], lo[1], hi[1]};
                }
                if constexpr (BLOCK_END) {
                    // the block's sums over the 16 pixel lanes of a row: fragment 0 is final after task 3 (one channel per later task), fragment 1 after task 7
                    if (t >= 4) { ssum[0][t - 4] = row16_sum(ssum[0][t - 4]); ssq[0][t - 4] = row16_sum(ssq[0][t - 4]); }
                    if (t == 7) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ssum[1][r] = row16_sum(ssum[1][r]); ssq[1][r] = row16_sum(ssq[1][r]); }
                    }
                }
            }
            if constexpr (TOP && SPREAD) {            // this slice's share of the next LDS-DMA round (the issue cost of a piece runs under the MFMAs)
#pragma unroll
                for (int i = 0; i < NDMA; ++i)
                    if (i * KK / NDMA == kk) issue_piece(rs_dma, i, stage ^ 1);
            }
            // thread the slice's vector instructions through the MFMAs (in program order they would run behind them, with the matrix pipe idle)
            {
                constexpr int VPM = STAT ? (TPS * 3 + 1) : (TPS * 2 + 1);
#pragma unroll
                for (int k = 0; k < FI * JG; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                }
            }
        });
        if constexpr (BLOCK_END) {                    // 4 stores: sums and sums of squares of both fragments (lanes col == 0, 16 bytes each)
            const long long blk = m0p >> 7;           // (m0p is the first pixel of the block's SECOND group: block = (m0p - 64) / 128 = m0p >> 7)
            const bool ok = m0p >= 0 && blk < p.n_mblocks;
            const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc((void*)(p.stat + (ok ? blk : 0) * p.Cd), 0, ok ? (int)((p.n_mblocks + 1LL) * p.Cd * 4) : 0, 0x00020000);
            const int plane = p.n_mblocks * p.Cd * 4;
#pragma unroll
            for (int f = 0; f < FI; ++f) {
                { const p1_u32x4 sv = __builtin_bit_cast(p1_u32x4, (p1_f32x4){ssum[f][0], ssum[f][1], ssum[f][2], ssum[f][3]}); __builtin_amdgcn_raw_buffer_store_b128(sv, rsT, voffT + f * 64, 0, 0); store_b128_guard(sv); }
                { const p1_u32x4 sv = __builtin_bit_cast(p1_u32x4, (p1_f32x4){ssq[f][0], ssq[f][1], ssq[f][2], ssq[f][3]}); __builtin_amdgcn_raw_buffer_store_b128(sv, rsT, voffT + f * 64, plane, 0); store_b128_guard(sv); }
            }
        }
        if constexpr (LATE) {
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) outP[SLOT][jj] = out[jj];
            m0_out[SLOT] = m0p;
        } else {
            store_group(out, m0p);
        }
    };

    const long long n_units = (p.M + UNIT - 1) / UNIT;
    long long unit = worker;
    if (unit >= n_units) return;
    // accumulators / addend of the group "before the first": an epilogue of zeros whose stores and statistics go nowhere (m0p < 0: empty descriptors)
#pragma unroll
    for (int f = 0; f < FI; ++f)
#pragma unroll
        for (int j = 0; j < JG; ++j) accB[f][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < (ADD ? JG : 1); ++j) adB[j] = (p1_u32x4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < (MASK ? JG : 1); ++j) mkB[j] = 0u;
#pragma unroll
    for (int f = 0; f < FI; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[f][r] = 0.f; ssq[f][r] = 0.f; }
    {
        const __amdgpu_buffer_rsrc_t rs0 = p1_rsrc(p.src + unit * UNIT * (K * 2), p.M - unit * UNIT, PXT, K * 2);
#pragma unroll
        for (int i = 0; i < NDMA; ++i) issue_piece(rs0, i, 0);
    }
    p1_wait_vmcnt<0>();                              // (the counted waits below assume the steady-state queue: the first round is waited for here)
    long long m0p = -(1LL << 40);                    // first pixel of the previous group of this wave (none yet)
    int stage = 0;
    for (; unit < n_units; unit += p.n_workers) {
        const long long m0u = unit * UNIT, m0n = (unit + p.n_workers) * UNIT;       // (a unit past the end: its LDS-DMA reads zeros, nobody multiplies them)
        if constexpr (NSUB == 2) {
            section(std::integral_constant<int, 0>{}, accA, accB, adA, adB, mkA, mkB, m0u, m0p, m0u + PXT, stage);
            stage ^= 1;
            section(std::integral_constant<int, 1>{}, accB, accA, adB, adA, mkB, mkA, m0u, m0u + px0, m0n, stage);
            stage ^= 1;
            m0p = m0u + PXT + px0;
        } else {
            section(std::integral_constant<int, 0>{}, accA, accB, adA, adB, mkA, mkB, m0u, m0p, m0n, stage);
            section(std::integral_constant<int, 1>{}, accB, accA, adB, adA, mkB, mkA, m0u, m0u + px0, m0n, stage);
            stage ^= 1;
            m0p = m0u + px0 + 64;
        }
    }
    // ---- drain: the epilogue of the last group, alone ----
    p1_wait_vmcnt<0>();                              // (also the last LDS-DMA round, a unit past the end: nothing may land in LDS after the workgroup has left)
    if constexpr ((P1_LATE != 0) && STAT) {
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) store_group(outP[q], m0_out[q]);
    }
    {
        f32x4_t (&accp)[FI][JG] = accB;
        p1_u32x4 (&adp)[ADD ? JG : 1] = adB;
        if constexpr (ADD) p1_wait_loads4<0>(adp[0], adp[ADD ? 1 : 0], adp[ADD ? 2 : 0], adp[ADD ? 3 : 0]);
        if constexpr (MASK) {
            unsigned (&mkp)[MASK ? JG : 1] = mkB;
            asm volatile("" : "+v"(mkp[0]), "+v"(mkp[MASK ? 1 : 0]), "+v"(mkp[MASK ? 2 : 0]), "+v"(mkp[MASK ? 3 : 0]));
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) adp[jj] = p1_masked(adp[jj], mkp[MASK ? jj : 0]);
        }
        const __amdgpu_buffer_rsrc_t rsD = p1_rsrc(p.dst + m0p * row_b, p.M - m0p, 64, row_b);
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) {
            unsigned q[FI][2];
            p1_u32x2 alo = {0u, 0u}, ahi = {0u, 0u};
            if constexpr (ADD) {
                alo = __builtin_amdgcn_permlane16_swap(adp[ADD ? jj : 0][0], adp[ADD ? jj : 0][2], false, false);
                ahi = __builtin_amdgcn_permlane16_swap(adp[ADD ? jj : 0][1], adp[ADD ? jj : 0][3], false, false);
            }
#pragma unroll
            for (int f = 0; f < FI; ++f) {
                float v[4] = {accp[f][jj][0], accp[f][jj][1], accp[f][jj][2], accp[f][jj][3]};
                if constexpr (ADD) {
                    const unsigned w0 = f ? alo[1] : alo[0], w1 = f ? ahi[1] : ahi[0];
                    v[0] += __uint_as_float(w0 << 16); v[1] += __uint_as_float(w0 & 0xffff0000u);
                    v[2] += __uint_as_float(w1 << 16); v[3] += __uint_as_float(w1 & 0xffff0000u);
                }
                q[f][0] = pack_bf16x2(v[0], v[1]); q[f][1] = pack_bf16x2(v[2], v[3]);
                if constexpr (STAT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ssum[f][r] += v[r]; ssq[f][r] = fmaf(v[r], v[r], ssq[f][r]); }
                }
            }
            const p1_u32x2 lo = __builtin_amdgcn_permlane16_swap(q[0][0], q[1][0], false, false);
            const p1_u32x2 hi = __builtin_amdgcn_permlane16_swap(q[0][1], q[1][1], false, false);
            const p1_u32x4 o = {lo[0], hi[0], lo[1], hi[1]};
            __builtin_amdgcn_raw_buffer_store_b128(o, rsD, voffS[jj], 0, 2);
            store_b128_guard(o);
        }
        if constexpr (STAT) {
            const long long blk = m0p >> 7;
            const bool ok = blk < p.n_mblocks;
            const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc((void*)(p.stat + (ok ? blk : 0) * p.Cd), 0, ok ? (int)((p.n_mblocks + 1LL) * p.Cd * 4) : 0, 0x00020000);
            const int plane = p.n_mblocks * p.Cd * 4;
#pragma unroll
            for (int f = 0; f < FI; ++f) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { ssum[f][r] = row16_sum(ssum[f][r]); ssq[f][r] = row16_sum(ssq[f][r]); }
                { const p1_u32x4 sv = __builtin_bit_cast(p1_u32x4, (p1_f32x4){ssum[f][0], ssum[f][1], ssum[f][2], ssum[f][3]}); __builtin_amdgcn_raw_buffer_store_b128(sv, rsT, voffT + f * 64, 0, 0); store_b128_guard(sv); }
                { const p1_u32x4 sv = __builtin_bit_cast(p1_u32x4, (p1_f32x4){ssq[f][0], ssq[f][1], ssq[f][2], ssq[f][3]}); __builtin_amdgcn_raw_buffer_store_b128(sv, rsT, voffT + f * 64, plane, 0); store_b128_guard(sv); }
            }
        }
    }
#endif
}

template <int K, int NWC, int NW, bool STAT, bool ADD, bool MASK = false> static void p1_launch(const P1Params& p, int grid, hipStream_t st) {
    hipLaunchKernelGGL((conv1x1_pipe_kernel<K, NWC, NW, STAT, ADD, MASK>), dim3(grid), dim3(NW * 64), 0, st, p);
}
template <int K, int NWC, int NW> static void p1_pick(const P1Params& p, int grid, bool stat, bool add, hipStream_t st) {
    if (stat) p1_launch<K, NWC, NW, true, false>(p, grid, st);
    else if (add && p.addend_mask) p1_launch<K, NWC, NW, false, true, true>(p, grid, st);
    else if (add) p1_launch<K, NWC, NW, false, true>(p, grid, st);
    else p1_launch<K, NWC, NW, false, false>(p, grid, st);
}

// returns 1 if the kernel handled the call: bf16 1x1 convolution (forward or input gradient), 64 / 128 / 256 input channels, output channels a
// multiple of 128, one shared weight set, optional same-shape addend (mode 1), optional BatchNorm partial sums (mode 0).  FB_C1S_PIPE=0: the
// round-3 form (conv1x1_stream.hip) takes these calls (same results up to the order of the statistics' fp32 additions)
// the shapes this kernel takes (fb_conv_masked_addend_supported asks for the masked-addend form: conv_igemm.hip)
int fb_conv1x1_pipe_takes(const fb_conv_args* a) {
    const char* sw = getenv("FB_C1S_PIPE");           // (read per call: the tests compare the two forms inside one process)
    if (sw != nullptr && atoi(sw) == 0) return 0;
    if (a->R != 1 || a->S != 1 || a->stride != 1 || a->pad != 0 || a->dtype != FB_BF16) return 0;
    if (a->Hs != a->Hd || a->Ws != a->Wd) return 0;
    if (a->Cs != 64 && a->Cs != 128 && a->Cs != 256) return 0;
    // K = 128 without addend stays on the round-3 kernel: measured SLOWER here (forward 128 -> 512 @28x28, 1024 images: 257 us against 203; input gradient
    // 261 against 214) although it is faster with the stores kept in L2 (159 us) -- its write stream meets HBM worse, and neither the stores behind the
    // barrier nor 4-wave workgroups changed that; with the addend it wins (375 against 399)
    if (a->Cs == 128 && !a->addend && !(sw != nullptr && atoi(sw) == 2)) return 0;
    if (a->Cd % 128 != 0) return 0;
    if (a->wset_stride != 0 && a->imgs_per_wset > 0 && a->imgs_per_wset < a->n_img) return 0;
    if (a->bst_x) return 0;
    if (a->addend_mask && (!a->addend || a->mode != 1 || a->addend_mode != 1 || getenv("FB_C1P_NO_MASK") != nullptr)) return 0;
    if (a->mode == 0 && a->addend) return 0;
    if (a->mode == 1 && (a->stat_partial || (a->addend && a->addend_mode != 1))) return 0;
    const long long M = (long long)a->n_img * a->Hd * a->Wd;
    if (M * a->Cs * 2 >= (1LL << 40) || M * a->Cd * 2 >= (1LL << 40) || (M / 128 + 2) * a->Cd * 8 >= (1LL << 31)) return 0;
    return 1;
}

int fb_try_conv1x1_pipe(const fb_conv_args* a, hipStream_t st) {
    if (!fb_conv1x1_pipe_takes(a)) return 0;
    const long long M = (long long)a->n_img * a->Hd * a->Wd;
    // 8-wave workgroups (one per CU, 256 channels) where the layer has them, else 4-wave ones (two per CU, 128 channels); FB_C1P_NW overrides
    static const int nw_env = fb_getenv_experimental("FB_C1P_NW") ? atoi(fb_getenv_experimental("FB_C1P_NW")) : 0;
    int nw = nw_env ? nw_env : 8;
    const int ch_per_wg8 = a->Cs == 64 ? 128 : 256;
    if (nw == 8 && a->Cd % ch_per_wg8 != 0) nw = 4;
    const int NWC = nw == 8 ? (a->Cs == 64 ? 4 : 8) : (a->Cs == 64 ? 2 : 4);
    const int pxt = 16384 / a->Cs, unit = pxt > 128 ? pxt : 128;
    P1Params p;
    p.src = (const char*)a->src; p.wgt = (const char*)a->wgt; p.dst = (char*)a->dst; p.addend = (const char*)a->addend; p.stat = a->stat_partial;
    p.addend_mask = (const unsigned char*)a->addend_mask;
    p.M = M; p.Cd = a->Cd;
    p.n_co = a->Cd / (NWC * 32);
    p.n_mblocks = (int)((M + 127) / 128);
    const long long n_units = (M + unit - 1) / unit;
    const int n_cu = fb_persistent_cus();
    long long workers = ((nw == 8 ? 1LL : 2LL) * n_cu) / p.n_co;
    if (workers < 1) workers = 1;
    if (workers > n_units) workers = n_units;
    p.n_workers = (int)workers;
#ifdef FB_C1P_EXPERIMENTS
    p.exp = getenv("FB_C1P_EXP") ? atoi(getenv("FB_C1P_EXP")) : 0;
#else
    p.exp = 0;
#endif
    const int grid = p.n_workers * p.n_co;
    const bool stat = a->stat_partial != nullptr, add = a->addend != nullptr;
    if (nw == 8) {
        if (a->Cs == 64) p1_pick<64, 4, 8>(p, grid, stat, add, st);
        else if (a->Cs == 128) p1_pick<128, 8, 8>(p, grid, stat, add, st);
        else p1_pick<256, 8, 8>(p, grid, stat, add, st);
    } else {
        if (a->Cs == 64) p1_pick<64, 2, 4>(p, grid, stat, add, st);
        else if (a->Cs == 128) p1_pick<128, 4, 4>(p, grid, stat, add, st);
        else p1_pick<256, 4, 4>(p, grid, stat, add, st);
    }
    return 1;
}
