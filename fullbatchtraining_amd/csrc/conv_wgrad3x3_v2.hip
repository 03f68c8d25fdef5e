// Weight gradient of 3x3 / pad 1 convolutions of stride 1 and 2, bf16, all nine taps from one LDS-resident input halo.
//
//   dW[co][tap][ci] = sum_p dY[p][co] * X[stride*p + tap - 1][ci]     (output maps 32x32 .. 4x4 and 56 / 28 / 14; stride 2: 16x16 .. 4x4)
//
// One workgroup owns a 64(co) x 64(ci) x 9(tap) output block and a slice of whole images of one chunk (split-K over images).
// Wave w accumulates ci block [16w,16w+16) x 9 taps x 64 co (36 fragments = 144 accumulator registers).  Per K-step
// (64 output pixels; 32 for 4x4 maps) dY[px][64 co] and the X halo are staged once and feed all nine taps:
//   * dY rows and the halo go HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds` (zero padding = out-of-range voffset),
//     double-buffered: the loads of step s+1 are in flight while step s is multiplied
//   * 128-byte pixel rows, 32-byte-slot XOR swizzle  slot' = slot ^ ((row>>1)&1 | ((row>>3)&1)<<1)  applied on the source
//     side: the four rows of a transposed 4x16 read hit four distinct bank octets
//   * the halo row pitch (48 / 32 / 16 / 16 rows for W = 32 / 16 / 8 / 4) is a multiple of 16, which keeps bits 1 and 3 of the
//     row index independent of the vertical tap and of the 32-pixel block: every ds_read_b64_tr_b16 address is one of a few
//     precomputed lane registers + an immediate -- no address arithmetic in the loop
//   * stride 2: the K-step is 32 output pixels; the halo holds the 2 RS + 1 input rows they touch, a transposed read walks
//     every second halo row (slot swizzle (row>>1)&3, pitch a multiple of 8 rows; same-parity rows share a 128-byte bank
//     half, so these reads are 2-way conflicted -- LDS is far from being the limit here).  Before this path the stride-2
//     layers ran one workgroup per tap and re-read dY nine times (156 TF/s on the 64->128 layer)
// Output: fp32 slabs [group][split][co][tap][ci] (same layout as conv_wgrad.hip), reduced in fixed order by fb_wgrad_reduce.
#include "common.h"

#include <type_traits>

struct Wgrad3V2Params {
    const char* x; const char* dy; float* out;
    int n_img, H, Cs, Cd;
    int imgs_per_group, imgs_per_block, split_k;
    long long group_stride;
    int n_chains; float* sq_part;                            // CHAIN: see conv_wgrad3x3_v2_kernel
};

namespace {
typedef __attribute__((ext_vector_type(2))) unsigned w3_u32x2;
template <int OFF> __device__ __forceinline__ w3_u32x2 w3_read_tr(unsigned byte_addr) {
    w3_u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
template <int N> __device__ __forceinline__ void w3_wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> __device__ __forceinline__ void w3_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int I, int N, typename F> __device__ __forceinline__ void w3_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); w3_static_for<I + 1, N>(f); }
}
__device__ __forceinline__ uint4 w3_join(w3_u32x2 lo, w3_u32x2 hi) { return make_uint4(lo[0], lo[1], hi[0], hi[1]); }
__device__ __forceinline__ int w3_f(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }
constexpr unsigned W3_OOB = 0x80000000u;

// Geometry of one K-step for OUTPUT feature-map width W and stride SD.
//   stride 1: W = 32: 2 image rows per step, W = 16: 4 image rows, W = 8: one whole image, W = 4: two whole images (32 pixels)
//   stride 2: 32 output pixels per step -- W = 16: 2 output rows (5 input rows), W = 8: 4 output rows (9 input rows, half an
//             image), W = 4: two whole images (9 input rows each)
// Feature maps whose width is not a power of two (ImageNet-shaped models: 56 / 28 / 14): a 32-pixel block no longer covers whole image
// rows, so a lane's halo row depends on the block (PER_BLOCK: one address register per block, tap column and 4-pixel half instead of an
// immediate) and a step covers RS whole rows padded to a multiple of 32 pixels (the padding pixels carry dY = 0 through out-of-range DMA
// offsets): W = 56: 1 row (56 of 64 pixels), W = 28: 4 rows (112 of 128), W = 14: 7 rows (98 of 128).  Stages of 32-40 KiB: two workgroups per CU.
template <int W> struct W3GeoGen {
    static constexpr bool PER_BLOCK = true;
    static constexpr int RS = W == 56 ? 1 : (W == 28 ? 4 : 7);
    static constexpr int VALID = RS * W;                                       // real output pixels of a step
    static constexpr int KPX = (VALID + 31) / 32 * 32;
    static constexpr int PITCH = W == 56 ? 64 : (W == 28 ? 32 : 16);
    static constexpr int IMGS = 1;
    static constexpr int IMG_ROWS = (RS + 2) * PITCH;
    static constexpr int HROWS = IMG_ROWS;                                     // 192, 192, 144
    static constexpr bool WHOLE = false;
    // halo row (vertical tap index 0) of pixel p of the step for tap column s; padding pixels read row 0 (their dY is zero)
    __device__ static __forceinline__ int pix_row(int p, int s) { return p < VALID ? (p / W) * PITCH + (p % W) + s : 0; }
    static constexpr int blk_rows(int, int R) { return R * PITCH; }           // (the block is in the lane registers)
    __device__ static __forceinline__ int fB(int row) { return w3_f(row); }
};
template <int W, int SD> struct W3Geo;
template <> struct W3Geo<56, 1> : W3GeoGen<56> {};
template <> struct W3Geo<28, 1> : W3GeoGen<28> {};
template <> struct W3Geo<14, 1> : W3GeoGen<14> {};
template <int W> struct W3Geo<W, 1> {
    static constexpr bool PER_BLOCK = false;
    static constexpr int VALID = W == 4 ? 32 : 64;
    static constexpr int KPX = W == 4 ? 32 : 64;                               // output pixels per step
    static constexpr int PITCH = W == 32 ? 48 : (W == 16 ? 32 : 16);           // halo row pitch (multiple of 16)
    static constexpr int RS = W >= 16 ? 64 / W : W;                            // output rows covered by a step (per image)
    static constexpr int IMGS = W == 4 ? 2 : 1;                                // whole images per step (W <= 8)
    static constexpr int IMG_ROWS = (RS + 2) * PITCH;                          // halo rows of one image part
    static constexpr int HROWS = IMGS * IMG_ROWS;                              // 192, 192, 160, 192
    static constexpr int HI_DELTA = W == 4 ? 16 : 4;                           // halo rows between pixel p and p+4
    static constexpr bool WHOLE = W <= 8;                                      // a step holds whole images: static y validity
    // halo row (vertical tap index 0, 32-pixel block 0) of pixel pl in [0,32) for tap column s
    __device__ static __forceinline__ int lane_row(int pl, int s) {
        if constexpr (W == 4) return (pl / 16) * IMG_ROWS + ((pl % 16) / 4) * PITCH + (pl % 4) + s;
        else return (pl / W) * PITCH + (pl % W) + s;
    }
    // rows added by 32-pixel block BLK and vertical tap index R: the immediate part, a multiple of 16 rows
    static constexpr int blk_rows(int BLK, int R) { return ((BLK * 32) / W + R) * PITCH; }
    __device__ static __forceinline__ int fB(int row) { return w3_f(row); }
};
template <int W> struct W3Geo<W, 2> {
    static constexpr bool PER_BLOCK = false;
    static constexpr int VALID = 32;
    static constexpr int KPX = 32;
    static constexpr int PITCH = W == 16 ? 40 : (W == 8 ? 24 : 16);            // >= 2W + 1 columns, multiple of 8
    static constexpr int RS = W == 16 ? 2 : 4;
    static constexpr int IMGS = W == 4 ? 2 : 1;
    static constexpr int IMG_ROWS = (2 * RS + 1) * PITCH;
    static constexpr int HROWS = IMGS * IMG_ROWS;                              // 200, 216, 288
    static constexpr int HI_DELTA = W == 4 ? 2 * PITCH : 8;                    // output pixel +4 = next output row (W = 4) / 8 columns
    static constexpr bool WHOLE = W == 4;
    __device__ static __forceinline__ int lane_row(int pl, int s) {
        if constexpr (W == 4) return (pl / 16) * IMG_ROWS + 2 * ((pl % 16) / 4) * PITCH + 2 * (pl % 4) + s;
        else return 2 * (pl / W) * PITCH + 2 * (pl % W) + s;
    }
    static constexpr int blk_rows(int, int R) { return R * PITCH; }
    __device__ static __forceinline__ int fB(int row) { return (row >> 1) & 3; }
};
// 4x4 maps, stride 1, COMPACT: an image is its 16 pixels + ONE zero row (17 halo rows instead of 6 x 16), a K-step is four whole images
// (64 pixels, 11 KiB of halo + 8 KiB of dY instead of 24 + 4 KiB for 32 pixels): a third of the LDS-DMA bytes and half the barriers per
// MFMA.  A lane's tap neighbour (or the image's zero row) is one of nine address registers per 4-pixel half instead of three.
struct W3GeoCP4 {
    static constexpr bool PER_BLOCK = false;
    static constexpr int VALID = 64, KPX = 64, PITCH = 4, RS = 4, IMGS = 4;
    static constexpr int IMG_ROWS = 17;                                        // two images per 32-pixel block: rows 0..33 of the block
    static constexpr int BLK_ROWS = 48;                                        // block pitch: a multiple of 16 rows (the swizzle bits of a row must not depend on the block)
    static constexpr int HROWS = 88;                                           // 48 + 34 rows, padded to whole 1 KiB groups
    static constexpr int HI_DELTA = 4;
    static constexpr bool WHOLE = true;
    // halo row of pixel pl in [0,32) (block 0) for tap u: the neighbour pixel, or the zero row of the image
    __device__ static __forceinline__ int tap_row(int pl, int u) {
        const int img = pl / 16, y = (pl % 16) / 4 + u / 3 - 1, x = pl % 4 + u % 3 - 1;
        return img * IMG_ROWS + ((y >= 0 && y < 4 && x >= 0 && x < 4) ? y * 4 + x : 16);
    }
    __device__ static __forceinline__ int lane_row(int, int) { return 0; }
    static constexpr int blk_rows(int BLK, int) { return BLK * BLK_ROWS; }
    __device__ static __forceinline__ int fB(int row) { return w3_f(row); }
};
}  // namespace

// CHAIN (compact 4x4 only): the SUM of the chunk gradients instead of one gradient per chunk.  The reference's statistics need every chunk's
// gradient norm, so the per-chunk form writes 147 KiB of fp32 per workgroup and chunk (7 of the 11 M parameters of ResNet-18 sit in the three
// 512 -> 512 layers: 2.8 GB per chunk group written here and read again by fb_mt_accumulate).  Here workgroup (tile, s) of n_chains walks chunks
// s, s + n_chains, ... : per chunk it accumulates the chunk's tile, adds its sum of squares to sq_part[chunk][tile][wave], folds it into a second
// register set and starts over; at the end it writes ONE tile, slab s of n_chains (fb_wgrad_reduce adds the slabs in fixed order).  Two
// accumulator sets do not fit next to 64 output channels per wave: EIGHT waves, a wave owns 16 ci x 9 taps x 32 co (72 + 72 registers).
template <int W, int SD, bool CP = false, bool CHAIN = false>
__global__ __launch_bounds__(CHAIN ? 512 : 256) void conv_wgrad3x3_v2_kernel(const Wgrad3V2Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(!CP || (W == 4 && SD == 1), "compact layout: 4x4 maps, stride 1");
    static_assert(!CHAIN || CP, "chunk chains: compact 4x4 layout");
    constexpr int NWV = CHAIN ? 8 : 4, FIW = CHAIN ? 2 : 4;     // waves per workgroup; 16-channel dY fragments per wave
    using G = typename std::conditional<CP, W3GeoCP4, W3Geo<W, SD>>::type;
    constexpr int WI = SD * W;                          // input width; p.H is the OUTPUT height
    const int Hi = SD * p.H;
    constexpr int KPX = G::KPX, PITCH = G::PITCH, HROWS = G::HROWS;
    constexpr int A_BYTES = KPX * 128, B_BYTES = HROWS * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int NGA = KPX / 8, NGB = HROWS / 8;      // 1 KiB row groups of the dY tile / of the halo
    constexpr int KA = (NGA + NWV - 1) / NWV, KB = (NGB + NWV - 1) / NWV;
    // (CHAIN: three stages with counted waits were measured -- 897 us against 807 us with two at 512->512, 98 chunks, 8 chains)
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cib = CHAIN ? (wave & 3) : wave, coh = CHAIN ? (wave >> 2) : 0;      // the wave's 16-channel block of X; (CHAIN) its 32-channel half of dY
    // 1-D grid, XCD-aware order: workgroups are dealt to the 8 XCDs round-robin, so the remap makes consecutive work items
    // live on ONE XCD -- the (Cd/64)*(Cs/64) tiles of a (chunk, K slice) read the same dY rows and X halos and now share them
    // in that XCD's L2 (PMC: every launch fetched ~1.3 GB before, dY and X once per tile row / column)
    const int tiles_n = p.Cs / 64, tiles = tiles_n * (p.Cd / 64);
    const int n_items = gridDim.x;
    int item;
    {
        const int b = blockIdx.x, q = n_items >> 3, r = n_items & 7, xcd = b & 7, slot = b >> 3;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tile = item % tiles, gs = item / tiles;
    const int tile_m = __builtin_amdgcn_readfirstlane(tile / tiles_n), tile_n = __builtin_amdgcn_readfirstlane(tile % tiles_n);
    const int group = __builtin_amdgcn_readfirstlane(CHAIN ? gs : gs / p.split_k), split = __builtin_amdgcn_readfirstlane(CHAIN ? 0 : gs % p.split_k);
    // CHAIN: `group` is the chain index s; the descriptors cover the whole tensors and a step addresses its images from there
    const int n_chunks = p.n_img / p.imgs_per_group;
    const int my_chunks = CHAIN ? (n_chunks - group + p.n_chains - 1) / p.n_chains : 0;
    const int spc = p.imgs_per_group / G::IMGS;             // (CHAIN) K-steps per chunk
    const int img0 = CHAIN ? 0 : group * p.imgs_per_group + split * p.imgs_per_block;
    const int img_end = CHAIN ? p.n_img : min(img0 + p.imgs_per_block, (group + 1) * p.imgs_per_group);
    const int steps_per_img = G::WHOLE ? 1 : p.H / G::RS;
    const int n_steps = CHAIN ? my_chunks * spc : (G::WHOLE ? (img_end - img0) / G::IMGS : (img_end - img0) * steps_per_img);
    const int rowA_b = p.Cd * 2, rowB_b = p.Cs * 2;    // bytes per pixel of dY / X
    const int lrow8 = lane >> 3;

    // descriptors based at the workgroup's first image (a K slice is at most a chunk: the 32-bit offsets below stay small whatever the tensors'
    // sizes -- chunk groups beyond 2^31 bytes per tensor)
    const int n_own = max(img_end - img0, 0);
    const __amdgpu_buffer_rsrc_t rsrcA =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.dy + (long long)img0 * p.H * W * rowA_b), 0, n_own * p.H * W * rowA_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (long long)img0 * Hi * WI * rowB_b), 0, n_own * Hi * WI * rowB_b, 0x00020000);
    // per-lane source offsets relative to the first pixel of the step; source-side slot swizzle
    unsigned voffA[KA];
#pragma unroll
    for (int k = 0; k < KA; ++k) {
        const int row = (wave + NWV * k) * 8 + lrow8;                        // pixel of the step
        const int lslot = ((lane & 7) >> 1) ^ w3_f(row);
        voffA[k] = row < G::VALID ? (unsigned)(row * rowA_b + tile_m * 128 + lslot * 32 + (lane & 1) * 16) : W3_OOB;
    }
    int hyB[KB]; unsigned voffB[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const int row = (wave + NWV * k) * 8 + lrow8;                        // halo row index
        const int img_l = row / G::IMG_ROWS, rr = row - img_l * G::IMG_ROWS;
        const int hy = rr / PITCH, hx = rr - hy * PITCH;
        const int lslot = ((lane & 7) >> 1) ^ G::fB(row);
        const bool xok = hx >= 1 && hx <= WI && (wave + NWV * k) < NGB;
        const unsigned base = (unsigned)((img_l * Hi * WI + (hx - 1)) * rowB_b + tile_n * 128 + lslot * 32 + (lane & 1) * 16);
        if constexpr (CP) {                                                    // block b = rows 48 b ..: image i at + 17 i, its pixel q at + q, row 16 = zeros
            const int blk = row / W3GeoCP4::BLK_ROWS, rb = row % W3GeoCP4::BLK_ROWS, im = rb / 17, q = rb % 17;
            hyB[k] = 0;
            voffB[k] = (blk < 2 && rb < 34 && q < 16) ? (unsigned)(((blk * 2 + im) * 16 + q) * rowB_b + tile_n * 128 + lslot * 32 + (lane & 1) * 16) : W3_OOB;
        } else if constexpr (G::WHOLE) {
            const int sy = hy - 1;
            hyB[k] = 0;
            voffB[k] = (xok && sy >= 0 && sy < WI) ? base + (unsigned)(sy * WI * rowB_b) : W3_OOB;
        } else {
            hyB[k] = xok ? hy - 1 : -(1 << 20);
            voffB[k] = base;                                                   // + sy*W*rowB_b when the row is inside the image
        }
    }
    auto issue = [&](int stage, int step) {
        char* base = lds + stage * STAGE;
        int img, y0;
        if constexpr (CHAIN) { img = (group + (step / spc) * p.n_chains) * p.imgs_per_group + (step % spc) * G::IMGS; y0 = 0; }
        else if constexpr (G::WHOLE) { img = step * G::IMGS; y0 = 0; }                   // (image index relative to img0)
        else { img = step / steps_per_img; y0 = (step % steps_per_img) * G::RS; }
        const int soffA = (img * p.H + y0) * W * rowA_b;
#pragma unroll
        for (int k = 0; k < KA; ++k)
            if (wave + NWV * k < NGA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(base + (wave + NWV * k) * 1024), 16,
                                                         voffA[k], soffA, 0, 0);
        const int soffB = img * Hi * WI * rowB_b;
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            if (wave + NWV * k < NGB) {
                unsigned v = voffB[k];
                if constexpr (!G::WHOLE) {
                    const int sy = SD * y0 + hyB[k];
                    v = (unsigned)sy < (unsigned)Hi ? voffB[k] + (unsigned)(sy * WI * rowB_b) : W3_OOB;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (__attribute__((address_space(3))) void*)(base + A_BYTES + (wave + NWV * k) * 1024),
                                                         16, v, soffB, 0, 0);
            }
        }
    };

    // ---- precomputed fragment read addresses -------------------------------------------------------------------------------
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int t = lane & 15, g = lane >> 4;
    const int pl = g * 8 + (t >> 2);                     // pixel within the 32-pixel block (low half of the 8-pixel run)
    // A: row = pb + pl (+4); slot = i ^ f(row)   (f is independent of pb (multiple of 32) and of the +4: pl & 4 == 0)
    unsigned la[FIW];                                    // (CHAIN: fragments 2 coh, 2 coh + 1 of the tile's four)
#pragma unroll
    for (int i = 0; i < FIW; ++i) la[i] = lds0 + pl * 128 + (((i + coh * FIW) ^ w3_f(pl)) * 32) + (t & 3) * 8;
    // B: halo row = lane_row(pl, s) + immediate; slot = wave ^ f(row); the read of pixels +4 has its own registers
    // (compact 4x4: one register per TAP -- index s runs over the nine taps)
    constexpr int NBLK = G::PER_BLOCK ? KPX / 32 : 1, NS = CP ? 9 : 3;
    unsigned lb[NBLK][NS], lbh[NBLK][NS];
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            int row, rowh;
            if constexpr (CP) { row = W3GeoCP4::tap_row(pl, s); rowh = W3GeoCP4::tap_row(pl + 4, s); }
            else if constexpr (G::PER_BLOCK) { row = G::pix_row(blk * 32 + pl, s); rowh = G::pix_row(blk * 32 + pl + 4, s); }
            else { row = G::lane_row(pl, s); rowh = row + G::HI_DELTA; }
            lb[blk][s] = lds0 + A_BYTES + row * 128 + ((cib ^ G::fB(row)) * 32) + (t & 3) * 8;
            lbh[blk][s] = lds0 + A_BYTES + rowh * 128 + ((cib ^ G::fB(rowh)) * 32) + (t & 3) * 8;
        }

    f32x4_t acc[9][FIW], tot[CHAIN ? 9 : 1][CHAIN ? FIW : 1];
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
        for (int i = 0; i < FIW; ++i) {
            acc[u][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            if constexpr (CHAIN) tot[u][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }

    if (n_steps > 0) {
        issue(0, 0);
        w3_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }
    for (int step = 0; step < n_steps; ++step) {
        const int cur = step & 1;
        if (step + 1 < n_steps) issue(cur ^ 1, step + 1);
        const unsigned so = cur * STAGE;
        unsigned a0[FIW];
#pragma unroll
        for (int i = 0; i < FIW; ++i) a0[i] = la[i] + so;
        w3_static_for<0, KPX / 32>([&](auto blkc) {
            constexpr int BLK = decltype(blkc)::value;
            constexpr int PB = BLK * 32;                                  // first pixel of the 32-pixel block
            constexpr int BI = G::PER_BLOCK ? BLK : 0;
            unsigned b0[NS], b1[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) { b0[s] = lb[BI][s] + so; b1[s] = lbh[BI][s] + so; }
            uint4 af[FIW];
            w3_static_for<0, FIW>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                af[I] = w3_join(w3_read_tr<PB * 128>(a0[I]), w3_read_tr<PB * 128 + 512>(a0[I]));
            });
            // the LGKM counter holds 15 outstanding operations: read in three batches of <= 14
            uint4 bf[9];
            w3_static_for<0, 3>([&](auto bc) {
                constexpr int B0 = decltype(bc)::value * 3;
                w3_static_for<B0, B0 + 3>([&](auto uc) {
                    constexpr int U = decltype(uc)::value, R = U / 3, S = U % 3;
                    constexpr int OFF = G::blk_rows(BLK, R) * 128, BS = CP ? U : S;
                    bf[U] = w3_join(w3_read_tr<OFF>(b0[BS]), w3_read_tr<OFF>(b1[BS]));
                });
                w3_wait_lgkmcnt<0>();
#pragma unroll
                for (int u = B0; u < B0 + 3; ++u)
#pragma unroll
                    for (int i = 0; i < FIW; ++i) acc[u][i] = mma_chunk<bf16_tag>(af[i], bf[u], acc[u][i]);
            });
        });
        if constexpr (CHAIN) {
            if (step % spc == spc - 1) {                 // the chunk's tile is complete: its sum of squares, fold it into the total, start over
                float sq = 0.f;
#pragma unroll
                for (int u = 0; u < 9; ++u)
#pragma unroll
                    for (int i = 0; i < FIW; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            sq = fmaf(acc[u][i][q], acc[u][i][q], sq);
                            tot[u][i][q] += acc[u][i][q];
                            acc[u][i][q] = 0.f;
                        }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
                if (lane == 0) p.sq_part[((long long)(group + (step / spc) * p.n_chains) * tiles + tile) * 8 + wave] = sq;
            }
        }
        w3_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }

    // fp32 slab [co][tap][ci].  (Staging the tiles through LDS for 16-byte, 256-byte-contiguous stores was measured: no gain --
    // the epilogue costs what writing the per-chunk fp32 gradients to HBM costs, e.g. 283 MB per launch for a 512x512 layer.)
    // One buffer descriptor per slab, one per-lane offset, the (tap, fragment, row) part as a SCALAR offset: a store costs one s_mul instead of the
    // seven vector instructions (two 64-bit multiply-adds among them) of the pointer form (~1000 VALU instructions per workgroup before)
    float* out = CHAIN ? p.out + ((long long)group * p.Cd) * 9 * p.Cs : p.out + group * p.group_stride + ((long long)split * p.Cd) * 9 * p.Cs;
    const __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, p.Cd * 9 * p.Cs * 4, 0x00020000);
    const int voffO = (((tile_m * 64 + coh * 32 + (lane >> 4) * 4) * 9) * p.Cs + tile_n * 64 + cib * 16 + (lane & 15)) * 4;
    const int row4 = p.Cs * 4;
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
        for (int i = 0; i < FIW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(CHAIN ? tot[u][i][q] : acc[u][i][q]), rsrcO, voffO, ((i * 16 + q) * 9 + u) * row4, 0);
#endif
}

// returns 1 if handled (bf16 only; f32 stays on conv_wgrad3x3.hip / conv_wgrad.hip)
int fb_try_wgrad3x3_v2(const fb_wgrad_args* a, hipStream_t st) {
    static const bool disabled = getenv("FB_DISABLE_WGRAD3_V2") != nullptr;
    if (disabled || a->dtype != FB_BF16) return 0;
    if (a->R != 3 || a->S != 3 || a->pad != 1 || (a->stride != 1 && a->stride != 2)) return 0;
    const int SD = a->stride;
    if (a->Hs != SD * a->Hd || a->Ws != SD * a->Wd || a->Hd != a->Wd) return 0;
    const int W = a->Wd;
    const bool general = SD == 1 && (W == 56 || W == 28 || W == 14);     // ImageNet-shaped maps: rows padded to 32-pixel blocks
    if (W != 32 && W != 16 && W != 8 && W != 4 && !general) return 0;
    if (SD == 2 && W == 32) return 0;
    if (a->Cs % 64 != 0 || a->Cd % 64 != 0) return 0;
    // K slices are whole images; the last slice of a chunk may be shorter (or empty: it then contributes zeros)
    int imgs_per_block = (a->imgs_per_group + a->split_k - 1) / a->split_k;
    // 4x4 maps: image pairs per K-step; stride 1 with chunks of a multiple of four images: the compact layout, four images per K-step
    // (FB_WGRAD3_COMPACT=0: the padded layout)
    const char* cp_env = getenv("FB_WGRAD3_COMPACT");
    const bool compact = W == 4 && SD == 1 && a->imgs_per_group % 4 == 0 && !(cp_env && atoi(cp_env) == 0);
    if (compact) imgs_per_block = (imgs_per_block + 3) / 4 * 4;
    else if (W == 4) { if (a->imgs_per_group & 1) return 0; imgs_per_block += imgs_per_block & 1; }
    // (a workgroup addresses its own K slice only: at most a chunk)
    const long long bytes_x = (long long)(imgs_per_block + 2) * a->Hs * a->Ws * a->Cs * 2, bytes_dy = (long long)(imgs_per_block + 2) * a->Hd * a->Wd * a->Cd * 2;
    if (bytes_x >= (1LL << 31) || bytes_dy >= (1LL << 31)) return 0;
    Wgrad3V2Params p;
    p.x = (const char*)a->x; p.dy = (const char*)a->dy; p.out = a->dw_partial;
    p.n_img = a->n_img; p.H = a->Hd; p.Cs = a->Cs; p.Cd = a->Cd;
    p.imgs_per_group = a->imgs_per_group; p.split_k = a->split_k; p.imgs_per_block = imgs_per_block;
    p.group_stride = a->group_stride ? a->group_stride : (long long)a->split_k * a->Cd * 9 * a->Cs;
    p.n_chains = 0; p.sq_part = nullptr;
    const int n_groups = a->n_img / a->imgs_per_group;
    dim3 grid((a->Cd / 64) * (a->Cs / 64) * n_groups * a->split_k);
    if (SD == 1) {
        if (W == 56) hipLaunchKernelGGL((conv_wgrad3x3_v2_kernel<56, 1>), grid, dim3(256), 0, st, p);
        else if (W == 28) hipLaunchKernelGGL((conv_wgrad3x3_v2_kernel<28, 1>), grid, dim3(256), 0, st, p);
        else if (W == 14) hipLaunchKernelGGL((conv_wgrad3x3_v2_kernel<14, 1>), grid, dim3(256), 0, st, p);
        else if (W == 32) hipLaunchKernelGGL((conv_wgrad3x3_v2_kernel<32, 1>), grid, dim3(256), 0, st, p);
        else if (W == 16) hipLaunchKernelGGL((conv_wgrad3x3_v2_kernel<16, 1>), grid, dim3(256), 0, st, p);
        else if (W == 8) hipLaunchKernelGGL((conv_wgrad3x3_v2_kernel<8, 1>), grid, dim3(256), 0, st, p);
        else if (compact) hipLaunchKernelGGL((conv_wgrad3x3_v2_kernel<4, 1, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad3x3_v2_kernel<4, 1>), grid, dim3(256), 0, st, p);
    } else {
        if (W == 16) hipLaunchKernelGGL((conv_wgrad3x3_v2_kernel<16, 2>), grid, dim3(256), 0, st, p);
        else if (W == 8) hipLaunchKernelGGL((conv_wgrad3x3_v2_kernel<8, 2>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad3x3_v2_kernel<4, 2>), grid, dim3(256), 0, st, p);
    }
    return 1;
}

// ---- chunk-chained form (bf16, 3x3 / stride 1 / pad 1 on 4x4 maps): the SUM over the chunks + every chunk's sum of squares ----------------------
static bool w3_chain_ok(const fb_wgrad_args* a) {
    static const bool disabled = getenv("FB_DISABLE_WGRAD3_CHAIN") != nullptr;
    if (disabled || !a || a->dtype != FB_BF16) return false;
    if (a->R != 3 || a->S != 3 || a->pad != 1 || a->stride != 1) return false;
    if (a->Hd != 4 || a->Wd != 4 || a->Hs != 4 || a->Ws != 4) return false;
    if (a->Cs % 64 != 0 || a->Cd % 64 != 0 || a->imgs_per_group < 4 || a->imgs_per_group % 4 != 0 || a->n_img % a->imgs_per_group != 0) return false;
    if (a->bn_x || a->amax_x || a->amax_dy) return false;
    // a workgroup addresses the whole tensors from one descriptor
    if ((long long)a->n_img * 16 * a->Cs * 2 >= (1LL << 31) || (long long)a->n_img * 16 * a->Cd * 2 >= (1LL << 31)) return false;
    return true;
}

extern "C" int32_t fb_wgrad_chain_supported(const fb_wgrad_args* a) { return w3_chain_ok(a) ? 1 : 0; }

extern "C" int fb_conv2d_wgrad_chain(const fb_wgrad_args* a, int32_t n_chains, float* slabs, float* sq_part, void* stream) {
    if (!a || !a->x || !a->dy || !slabs || !sq_part) FB_FAIL(FB_ERR_ARG, "fb_conv2d_wgrad_chain: null pointer");
    if (!w3_chain_ok(a)) FB_FAIL(FB_ERR_UNSUPPORTED, "fb_conv2d_wgrad_chain: bf16 3x3 / stride 1 / pad 1 weight gradients on 4x4 maps, channels in multiples of 64, chunks of 4k images");
    const int n_chunks = a->n_img / a->imgs_per_group;
    if (n_chains < 1 || n_chains > n_chunks) FB_FAIL(FB_ERR_ARG, "fb_conv2d_wgrad_chain: n_chains=%d for %d chunks", n_chains, n_chunks);
    Wgrad3V2Params p;
    p.x = (const char*)a->x; p.dy = (const char*)a->dy; p.out = slabs;
    p.n_img = a->n_img; p.H = a->Hd; p.Cs = a->Cs; p.Cd = a->Cd;
    p.imgs_per_group = a->imgs_per_group; p.split_k = 1; p.imgs_per_block = a->imgs_per_group;
    p.group_stride = 0; p.n_chains = n_chains; p.sq_part = sq_part;
    dim3 grid((a->Cd / 64) * (a->Cs / 64) * n_chains);
    hipLaunchKernelGGL((conv_wgrad3x3_v2_kernel<4, 1, true, true>), grid, dim3(512), 0, (hipStream_t)stream, p);
    FB_CHECK_LAUNCH("fb_conv2d_wgrad_chain");
    return FB_OK;
}
