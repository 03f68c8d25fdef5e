/* fb_engine.h -- C ABI of libfbengine.so, the MI355X (gfx950) device side of the full-batch GD hot path.
 *
 * The reference (JonasGeiping/fullbatchtraining) has no FFI: its device work is whatever PyTorch dispatches from
 *   fullbatch/training/training.py:76-83   (_compute_batched_gradient: model fwd, CE, autograd.grad)
 *   fullbatch/models/resnets.py:179-230,296-316 (conv / BN / ReLU / pool / fc graph)
 *   fullbatch/models/modules.py:211-300    (GradRegularizer finite differences: foreach mul/add/sub/div, norms)
 *   fullbatch/training/training.py:45-47   (_stable_mean_accumulation), :162 (per-chunk squared norm)
 *   fullbatch/training/training.py:198-211 (global-norm clip)  +  torch.optim.SGD.step (optimizers.py:28)
 * Each entry point below names the reference call it replaces.  A maintainer binds these with ctypes
 * (see INTEGRATION.md); no torch types cross this boundary.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch caching allocator); nothing here allocates
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never synchronises
 *   - return 0 on success, a negative fb_status otherwise; fb_last_error_string() describes the last failure of the
 *     calling thread; no other global mutable state, re-entrant, one host thread per GPU
 *   - activations are NHWC ("pixel-major"): [image][y][x][channel], channel counts multiples of 32
 *   - `dtype`: FB_F32 (exact f32 MFMA 16x16x4) or FB_BF16 (bf16 MFMA 16x16x32, fp32 accumulate)
 *   - a launch processes `n_groups` chunks at once ("chunk group"); a chunk is the reference's unit of
 *     BN statistics / loss mean / gradient (training.py:150-168).  Per-chunk quantities are indexed [group][...]
 *   - conv weights live in K-contiguous "KRSC" form: [Cout][R*S][Cin]; per-chunk weight sets (finite-difference
 *     second pass, modules.py:226) are selected by image index / imgs_per_wset
 */
#ifndef FB_ENGINE_H
#define FB_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { FB_OK = 0, FB_ERR_ARG = -1, FB_ERR_SHAPE = -2, FB_ERR_LAUNCH = -3, FB_ERR_UNSUPPORTED = -4 } fb_status;
typedef enum { FB_F32 = 0, FB_BF16 = 1 } fb_dtype;

const char* fb_last_error_string(void);
int fb_abi_version(void);

/* Optional measurement aid (the only process-global state in the library, off by default): while enabled every
 * fb_conv2d / fb_conv2d_wgrad launch is bracketed by HIP events recorded on the launch stream.  fb_profile_read waits
 * for the recorded launches and returns, per kernel class {0: igemm fwd, 1: igemm dgrad, 2: wgrad}, the summed
 * elapsed milliseconds, the number of launches and the number of launches that were not recorded (pool exhausted). */
int fb_profile_enable(int on, int capacity);
int fb_profile_read(double* ms, int64_t* launches, int64_t* dropped);   /* arrays of 6: + {3: BN apply, 4: BN backward reduce, 5: BN backward apply} */
/* Per-launch records of everything recorded since the last fb_profile_read (which resets: call this one first).  Row i of `info`
 * ([cap][12] int32): {class, 11 shape words}: convolutions {n_img, Hs, Ws, Cs, Hd, Wd, Cd, R, stride, flags, kernel id}, BatchNorm passes
 * {pixels/128, C, pixels_per_group/128, dtype, residual?, mask?, dy_out?, pooled?, 0, 0, 0}; ms[i] = elapsed milliseconds.
 * Returns the number of rows written (<= cap) or a negative fb_status. */
int64_t fb_profile_read_launches(int32_t* info, float* ms, int64_t cap);

/* ---------------------------------------------------------------- native launch executor -------------------------- */
/* The reference drives its hot loop from the Python interpreter, one dispatch per operator (training.py:144-174).  A chunk group's
 * forward + backward here is a STATIC sequence of 430 (ResNet-18) to 6000 (ResNet-152) launches, so a host binding records it once and
 * replays it with one call: fb_cmdlist_add_call files an entry point (fb_cmd_fn_id(name)) with one 64-bit word per argument except the
 * trailing stream (integers / pointers extended to 64 bits, float / double as bit patterns; `blob`: the argument struct word 0 points
 * to -- copied into the list) and the INDEX of the stream it goes to; fb_cmdlist_add_event files a record (kind 1) or wait (kind 2) of a
 * library event (fb_event_new; fb_event_record / fb_event_wait are the eager forms on the same events, so eager and replayed regions
 * order against each other).  fb_cmdlist_replay issues the list in order on streams[index].  Lists are host objects: created /
 * destroyed by the caller, never shared between threads. */
int32_t fb_cmd_fn_id(const char* name);          /* -1: this entry point cannot be recorded */
int32_t fb_cmd_fn_nargs(int32_t fn);             /* arguments including the stream */
int32_t fb_event_new(void);                      /* -1 on failure */
int32_t fb_event_count(void);                    /* events created so far (they live as long as the process: hosts REUSE the ids of lists they drop) */
int fb_event_record(int32_t ev, void* stream);
int fb_event_wait(int32_t ev, void* stream);
void* fb_cmdlist_create(void);
void fb_cmdlist_destroy(void* list);
int64_t fb_cmdlist_size(const void* list);
int fb_cmdlist_add_call(void* list, int32_t fn, const uint64_t* words, int32_t n_words, int32_t stream_idx, const void* blob, int32_t blob_bytes);
int fb_cmdlist_add_event(void* list, int32_t kind, int32_t ev, int32_t stream_idx);
int fb_cmdlist_replay(const void* list, void* const* streams, int32_t n_streams);

/* ---------------------------------------------------------------- convolution ------------------------------------ */
/* Implicit-GEMM convolution on MFMA.  mode 0: forward  y = conv(x, w)        (ATen convolution, resnets.py:69,206,209,150)
 *                                      mode 1: dgrad    dx = conv_input_grad   (ATen convolution_backward, input part)
 * src  [n_img][Hs][Ws][Cs]   dst [n_img][Hd][Wd][Cd]
 * wgt  [n_wsets][Cd][R*S][Cs] (for mode 1 the caller passes the transposed set produced by fb_weight_prep)
 * addend (optional, mode 1): dst += addend            (addend_mode 1: same shape)
 *                            dst += 0.25*addend[y/2][x/2] (addend_mode 2: gradient of AvgPool2d(2,2), resnets.py:149)
 * stat_partial (optional, mode 0): [2][ceil(M/128)][Cd] per-128-pixel-block channel sums / sums of squares of the fp32
 *   accumulators, consumed by fb_bn_fwd_finalize (training-mode BatchNorm statistics, resnets.py:71).
 * bst_x + bst_mask + stat_partial (optional, mode 1): the reduction pass of the BatchNorm backward that CONSUMES dst (autograd's
 *   native_batch_norm_backward sums behind resnets.py:71,118-121), fused into the epilogue: with g = dst * relu_mask (bst_mask: the bitmask
 *   fb_bn_apply wrote for that BatchNorm's output, 1 byte per 8 channels) and x = bst_x (that BatchNorm's input, shape of dst),
 *   stat_partial receives [2][M/128][Cd] per-128-pixel-block sums of g and of g*x (RAW x: fb_bn_bwd_finalize with raw_x = 1
 *   turns them into dbeta / dgamma).  Ask fb_conv_bwd_stat_supported first; otherwise call fb_bn_bwd_reduce. */
typedef struct {
    const void* src; const void* wgt; void* dst; const void* addend; float* stat_partial;
    int32_t n_img, Hs, Ws, Cs, Hd, Wd, Cd;
    int32_t R, S, stride, pad, mode;
    int32_t imgs_per_wset; int64_t wset_stride;   /* elements between weight sets; 0 = shared */
    int32_t addend_mode; int32_t dtype;
    const void* addend_mask;   /* optional (addend_mode 1): ReLU bitmask of the addend as written by fb_bn_apply (1 byte per 16-byte
                                * vector): dst += addend only where the bit is set, i.e. the masked residual gradient d * (out > 0) of
                                * reference resnets.py:118-121 / autograd's threshold_backward without a materialised copy */
    const void* bst_x; const void* bst_mask;   /* optional (mode 1): fused BatchNorm-backward reduction, see above */
    const float* amax_src; const float* amax_wgt;   /* optional (FB_F32): device arrays holding the largest magnitudes of src and of wgt
                                * (fb_absmax).  Both set: every fp32 operand enters as two scaled fp16 pieces and a product takes three fp16
                                * MFMAs (22 significand bits per operand, fp32 accumulation); src is split inside the kernel, wgt must be the
                                * fp16x2 planes fb_weight_prep(amax) wrote.  amax_src[k] belongs to images [k*amax_imgs, (k+1)*amax_imgs) (one
                                * scale per chunk, so that a chunk's result does not depend on how chunks are batched), amax_wgt[w] to weight
                                * set w.  Unset: three bf16 pieces, six MFMAs (exact fp32 operands), or the exact-f32 MFMA with FB_F32_EXACT=1 */
    int32_t amax_imgs;
} fb_conv_args;
int fb_conv2d(const fb_conv_args* a, void* stream);
/* 1 if fb_conv2d implements addend_mask for these arguments (otherwise it fails with FB_ERR_UNSUPPORTED and the caller masks the
 * addend itself: fb_bn_bwd_apply's dy_out) */
int32_t fb_conv_masked_addend_supported(const fb_conv_args* a);
/* 1 if fb_conv2d implements the fused BatchNorm-backward reduction (bst_x / bst_mask / stat_partial in mode 1) for these arguments */
int32_t fb_conv_bwd_stat_supported(const fb_conv_args* a);
/* Largest magnitudes of n_sets slices of n fp32 values, set_stride floats apart (one streaming pass, atomic maximum of the bit patterns):
 * per_set = 1: out[s] = max |x[s*set_stride + i]| (one scale per chunk: the amax_* arrays above); per_set = 0: out[0] = the maximum
 * over all slices */
int fb_absmax(const float* x, int64_t n, int32_t n_sets, int64_t set_stride, int32_t per_set, float* out, void* stream);

/* wgrad: dw[g][split][Cd][R*S][Cs] (fp32 partial slabs) = sum over the pixels of chunk g (split-K slice `split`) of
 * dy[p][Cd] (x) x[src(p,tap)][Cs]        (ATen convolution_backward, weight part).  Deterministic: no atomics.
 * group_stride (floats): 0 = dense slabs as above; otherwise the slab of (g, split) starts at
 * dw_partial + g*group_stride + split*Cd*R*S*Cs -- with split_k == 1 and unpadded channels that writes the per-chunk
 * gradient straight into the [g][P] gradient arena (group_stride = P) and no fb_wgrad_reduce pass is needed. */
typedef struct {
    const void* x; const void* dy; float* dw_partial;
    int32_t n_img, Hs, Ws, Cs, Hd, Wd, Cd;
    int32_t R, S, stride, pad;
    int32_t imgs_per_group; int32_t split_k; int32_t dtype;
    int64_t group_stride;
    const float* amax_x; const float* amax_dy;   /* optional (FB_F32): largest magnitudes of x and dy PER GROUP of imgs_per_group images
                                                  * (fb_absmax, per_set): fp16x2 split as in fb_conv_args */
    /* optional (1x1, Cs < 64: the stem on its patches; bf16 and fp32 without amax_*; fb_wgrad_bn_fused_supported): the BatchNorm backward apply step INSIDE the
     * operand loader -- `dy` then holds the gradient w.r.t. the BatchNorm + ReLU OUTPUT, and the kernel multiplies by
     *   dy'[p][c] = coef[g][c][0] * (dy[p][c] masked by bn_mask) + coef[g][c][1] * bn_x[p][c] + coef[g][c][2]      (rounded to the storage type)
     * i.e. exactly what fb_bn_bwd_apply would have written to dx with the coefficients of fb_bn_bwd_finalize: a layer whose input gradient
     * nobody needs (the stem) saves writing dx and reading it back.  bn_mask: ReLU bitmask (1 byte per 16-byte vector) or NULL. */
    const void* bn_x; const void* bn_mask; const float* bn_coef;
} fb_wgrad_args;
int32_t fb_wgrad_bn_fused_supported(const fb_wgrad_args* a);
int fb_conv2d_wgrad(const fb_wgrad_args* a, void* stream);
/* Chunk-chained weight gradient (ABI v12; bf16, 3x3 / stride 1 / pad 1 on 4x4 maps, chunks = imgs_per_group of 4k images;
 * fb_wgrad_chain_supported): the SUM over the chunks of what fb_conv2d_wgrad writes per chunk, plus every chunk's sum of squares -- the two
 * things the accumulate-over-all-chunks loop keeps of a chunk gradient (running mean: reference training.py:45-47,163-165; grad_norms:
 * training.py:162).  Workgroup (tile, s) walks chunks s, s + n_chains, ... and writes ONE fp32 tile:
 *   slabs   [n_chains][Cd][9][Cs]        partial sums; fb_wgrad_reduce(slabs, out, 0, 1, n_chains, Cd, 9, Cs, Cs) adds them in fixed order
 *   sq_part [n_img/imgs_per_group][(Cd/64)*(Cs/64)][8]   per chunk: partial sums of squares of that chunk's gradient (sum them per chunk)
 * a->dw_partial, split_k and group_stride are ignored.  The sum differs from adding the per-chunk results one by one only in rounding. */
int32_t fb_wgrad_chain_supported(const fb_wgrad_args* a);
int fb_conv2d_wgrad_chain(const fb_wgrad_args* a, int32_t n_chains, float* slabs, float* sq_part, void* stream);

/* Workspace sizes (in floats) of the caller-owned scratch buffers the entry points below take; host-side arithmetic only,
 * no launch.  The library never allocates device memory.
 *   fb_ws_conv_stat_floats   : stat_partial of fb_conv2d(mode 0)  = 2 * ceil(n_img*Hd*Wd / 128) * Cd
 *   fb_ws_wgrad_slab_floats  : dw_partial of fb_conv2d_wgrad      = (n_img/imgs_per_group) * split_k * Cd * R*S * Cs
 *   fb_ws_bn_partial_floats  : partial of fb_bn_bwd_reduce        = 2 * ceil(n_pixels / 128) * C   (upper bound)
 *   fb_ws_mt_floats          : ws of the fb_mt_* reductions       = max(n_groups, 2) * FB_MT_BLOCKS */
int64_t fb_ws_conv_stat_floats(const fb_conv_args* a);
int64_t fb_ws_wgrad_slab_floats(const fb_wgrad_args* a);
int64_t fb_ws_bn_partial_floats(int64_t n_pixels, int32_t C);
int64_t fb_ws_mt_floats(int32_t n_groups);
/* sums the split_k slabs in fixed order, drops channel padding (Cs_pad -> Cs_real), writes [g][Cd][R*S][Cs_real]
 * at out + g*out_group_stride */
int fb_wgrad_reduce(const float* dw_partial, float* out, int64_t out_group_stride, int32_t n_groups, int32_t split_k,
                    int32_t Cd, int32_t taps, int32_t Cs_pad, int32_t Cs_real, void* stream);
/* master fp32 KRSC weights [Cout][taps][Cin_real] of n_wsets sets (wset_stride_in floats apart) ->
 * w_fwd [Cout][taps][Cin_pad] and (optional) w_dgrad [Cin_pad][taps][Cout] in `dtype`, sets wset_stride_out elements apart.
 * amax (optional, FB_F32): device array, amax[s] = largest magnitude of the master weights of set s (fb_absmax, per_set): the copies are
 * written as fp16x2 planes (per 32 values: 32 scaled fp16 high pieces, then 32 low pieces; same size as fp32) -- the weight format fb_conv2d
 * expects whenever fb_conv_args.amax_wgt is set (to this same scalar) */
int fb_weight_prep(const float* master, int64_t wset_stride_in, int64_t wset_stride_out, int32_t n_wsets, int32_t Cout, int32_t taps,
                   int32_t Cin_real, int32_t Cin_pad, void* w_fwd, void* w_dgrad, int32_t dtype, const float* amax, void* stream);

/* ---------------------------------------------------------------- batch norm ------------------------------------- */
/* Per (group, channel): mean, biased var from the partial sums; writes mean/var rows into the [n_groups][ch_total]
 * statistics tables at column ch_off, and scale = gamma*invstd, shift = beta - mean*scale  ([n_groups][C]). */
int fb_bn_fwd_finalize(const float* stat_partial, int32_t n_mblocks, int32_t n_groups, int32_t C, double count,
                       const float* gamma, const float* beta, int64_t param_group_stride, float eps,
                       float* mean_tab, float* var_tab, int32_t ch_total, int32_t ch_off,
                       float* scale, float* shift, float* invstd, void* stream);
/* y = relu?(x*scale[g][c] + shift[g][c] + residual)  residual: none | res | res*rscale[g][c]+rshift[g][c]
 * (BatchNorm2d + ReLU(inplace) + `out += identity`, resnets.py:217-228).
 * amax_out (optional, FB_F32): device array that receives max |y| of every statistics group (fb_absmax per_set semantics, tracked by the
 * same pass): the scale source of the fp16x2 convolutions that read y */
int fb_bn_apply(const void* x, void* y, const float* scale, const float* shift, const void* res, const float* rscale,
                const float* rshift, int64_t n_pixels, int32_t C, int64_t pixels_per_group, int64_t valid_pixels_per_group,
                int32_t relu, void* mask_out, void* pool_out, int32_t pool_W, int32_t dtype, float* amax_out, float* amax_ws, void* stream);
/* floats of scratch (amax_ws) that fb_bn_apply / fb_bn_bwd_apply need when amax_out is set: one per 512-vector run */
int64_t fb_ws_bn_amax_floats(int64_t n_pixels, int32_t C, int64_t pixels_per_group);
/* pool_out (optional): AvgPool2d(2,2) of y, [n_img][H/2][W/2][C] (the 'C' downsample shortcut of the NEXT block, resnets.py:149), written
 * by the same pass -- bit-identical to fb_avgpool2_fwd on y, one read of the activation less.  pool_W = image width of y; only where
 * fb_bn_apply_can_pool() says so (bf16, W*C == 2048: the 64@32, 128@16, 256@8 maps of ResNet-18/34). */
int32_t fb_bn_apply_can_pool(int32_t C, int32_t W, int64_t pixels_per_group, int32_t dtype);
/* valid_pixels_per_group (0 = all): chunk sizes that do not fill whole 128-pixel statistics blocks (data.batch_size=125, reference
 * data_preparation.py:64-72) are stored padded with zero images; the pixels [valid, pixels_per_group) of every group are written as
 * exact zeros with a clear mask, so they stay out of every later sum (statistics are divided by the REAL count). */
/* mask_out (optional): ReLU mask, one byte per 16-byte vector of y (bit k = element k > 0).  The backward kernels accept it
 * in place of y (1/16 of the bytes).  When both `mask` and `y` are NULL no mask is applied (BN without ReLU). */
/* sequential running-stat EMA for every BN channel of the network in one launch (momentum 0.1, unbiased var,
 * torch BatchNorm2d).  mean_tab/var_tab: [n_passes][n_groups][ch_total] (passes pass_stride floats apart).
 * Update order: for g in groups: for p in passes  -- i.e. chunk g's base pass, then its finite-difference passes,
 * exactly the order in which the reference's sequential loop touches the buffers. */
int fb_bn_running_update(float* running_mean, float* running_var, const float* mean_tab, const float* var_tab, int32_t n_passes,
                         int64_t pass_stride, const float* unbias, int32_t n_groups, int32_t ch_total, float momentum, void* stream);
/* backward: partial sums of dy and dy*xhat with dy = dout * (y > 0) when y != NULL (threshold_backward).  One partial row per
 * workgroup of 128..1024 pixels: fb_bn_bwd_reduce_rows() is the row count to hand to fb_bn_bwd_finalize as n_mblocks. */
int32_t fb_bn_bwd_reduce_rows(int64_t n_pixels, int64_t pixels_per_group);
int fb_bn_bwd_reduce(const void* dout, const void* y, const void* mask, const void* x, const float* mean_tab, const float* invstd,
                     int32_t ch_total, int32_t ch_off, float* partial, int64_t n_pixels, int32_t C,
                     int64_t pixels_per_group, int32_t dtype, void* stream);
/* dgamma/dbeta -> gradient arena (per group), coefficients for fb_bn_bwd_apply.  raw_x = 0: partial holds sums of dy and of dy*xhat
 * (fb_bn_bwd_reduce); raw_x = 1: sums of dy and of dy*x (the fused reduction of fb_conv2d): sum dy*xhat = invstd*(sum dy*x - mean*sum dy),
 * evaluated in double */
int fb_bn_bwd_finalize(const float* partial, int32_t n_mblocks, int32_t n_groups, int32_t C, double count,
                       const float* scale, const float* mean_tab, const float* invstd, int32_t ch_total, int32_t ch_off,
                       float* dgamma, float* dbeta, int64_t grad_group_stride, float* coef, int32_t raw_x, void* stream);
/* dx = c_dy*dy + c_x*x + c_0 ; optionally also stores dy (masked gradient, used by the shortcut branch); amax_out (optional, FB_F32):
 * receives max |dx| as in fb_bn_apply */
int fb_bn_bwd_apply(const void* dout, const void* y, const void* mask, const void* x, const float* coef, void* dx, void* dy_out,
                    int64_t n_pixels, int32_t C, int64_t pixels_per_group, int32_t dtype, float* amax_out, float* amax_ws, void* stream);

/* Two BatchNorms that take the SAME incoming gradient through the SAME ReLU mask (a downsampling block, resnets.py:222-230: out = relu(bn2(conv2) +
 * bn_s(conv_s)): both BatchNorm backward passes start from the gradient of `out`): one read of dout serves both.  fb_bn_bwd_reduce2 writes each
 * BatchNorm's partial rows (sum dy is common) exactly as fb_bn_bwd_reduce would -- finalize each with fb_bn_bwd_finalize -- and fb_bn_bwd_apply2
 * writes both dx tensors.  Bit-identical to the separate calls; the channel vectors of a pixel must divide 256. */
int fb_bn_bwd_reduce2(const void* dout, const void* mask, const void* x_a, const float* invstd_a, int32_t ch_off_a, float* partial_a,
                      const void* x_b, const float* invstd_b, int32_t ch_off_b, float* partial_b, const float* mean_tab, int32_t ch_total,
                      int64_t n_pixels, int32_t C, int64_t pixels_per_group, int32_t dtype, void* stream);
int fb_bn_bwd_apply2(const void* dout, const void* mask, const void* x_a, const float* coef_a, void* dx_a, const void* x_b, const float* coef_b,
                     void* dx_b, int64_t n_pixels, int32_t C, int64_t pixels_per_group, int32_t dtype, void* stream);
/* The three calls above in ONE pass over (dout, x) (csrc/bn_bwd_fused.hip; autograd's native_batch_norm_backward + threshold_backward behind
 * resnets.py:214-230): a cluster of workgroups that is resident as a whole keeps a statistics group's operands in registers between the
 * reduction and the apply step -- 3 tensor passes instead of 5.  Same arithmetic as reduce / finalize / apply (fp32 sums per thread, partial
 * rows added in fixed order in double, dx = c_dy*dy + c_x*x + c_0); the result does not depend on how many groups a launch holds.
 * fb_bn_bwd_fused_supported(): 0 where the shape is not for it (a group's vectors must tile 4096-vector slices and its cluster must fit the
 * device's 2 x #CU resident workgroups) -- the caller then takes the three-call form.  `partial`: fb_ws_bn_bwd_fused_floats() floats;
 * `sync`: fb_ws_bn_bwd_fused_ints(n_groups) int32, ZEROED once by the caller (the last word is a sticky error flag: a wait of ~2 s -- a
 * cluster that never became resident -- sets it instead of hanging the device). */
int32_t fb_bn_bwd_fused_supported(int64_t n_pixels, int32_t C, int64_t pixels_per_group, int32_t dtype);
int64_t fb_ws_bn_bwd_fused_floats(int64_t n_pixels, int32_t C, int64_t pixels_per_group, int32_t dtype);
int64_t fb_ws_bn_bwd_fused_ints(int64_t n_groups);
int fb_bn_bwd_fused(const void* dout, const void* mask, const void* x, const float* mean_tab, const float* invstd, const float* scale,
                    int32_t ch_total, int32_t ch_off, float* dgamma, float* dbeta, int64_t grad_group_stride, float* coef, void* dx,
                    void* dy_out, int64_t n_pixels, int32_t C, int64_t pixels_per_group, double count, int32_t dtype, float* partial,
                    int32_t* sync, void* stream);

/* ---------------------------------------------------------------- data path --------------------------------------- */
/* Stem patch gather: images [n_img][C][H][W] fp32 (device) -> patches [n_img][Ho][Wo][cin_pad] in `dtype`, element tap*C + c
 * (tap-major like the KRSC weights), zero beyond k*k*C and outside the image (the convolution's own zero padding); the stem
 * convolution (resnets.py:150 / :179) then runs as a 1x1 convolution over these rows.
 * Optional on-device augmentation (config/data/CIFAR10.yaml:11-13: RandomCrop(H, crop_pad) then RandomHorizontalFlip):
 * crop_oy/crop_ox [n_img] int8 in [0, 2*crop_pad] (device; both or neither), flip [n_img] int8 0/1 (device, optional);
 * pad_value: HOST array of C floats = value of a black pixel after normalisation (-mean/std), NULL = 0. */
int fb_stem_patches(const float* images, void* patches, int64_t n_img, int32_t C, int32_t H, int32_t W, int32_t k, int32_t stride,
                    int32_t pad, int32_t cin_pad, const int8_t* crop_oy, const int8_t* crop_ox, const int8_t* flip, int32_t crop_pad,
                    const float* pad_value, int32_t dtype, void* stream);

/* ---------------------------------------------------------------- pooling / head --------------------------------- */
int fb_avgpool2_fwd(const void* x, void* y, int32_t n_img, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);
/* MaxPool2d(3,2,1) of the 'standard' stem (resnets.py:78); bwd scatters through recomputed argmax */
int fb_maxpool3s2_fwd(const void* x, void* y, int32_t n_img, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);
int fb_maxpool3s2_bwd(const void* x, const void* dy, void* dx, int32_t n_img, int32_t H, int32_t W, int32_t C, int32_t dtype,
                      void* stream);
/* the same pooling with the argmax remembered (what autograd's max_pool2d_with_indices keeps, resnets.py:78): idx [n][Ho][Wo][C] bytes, the window
 * position 0..8 (row-major, first maximum) every output element came from; the backward pass reads it instead of the pre-pool tensor.  Same bits
 * as fb_maxpool3s2_fwd / fb_maxpool3s2_bwd (ABI v13). */
int fb_maxpool3s2_fwd_idx(const void* x, void* y, void* idx, int32_t n_img, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);
int fb_maxpool3s2_bwd_idx(const void* idx, const void* dy, void* dx, int32_t n_img, int32_t H, int32_t W, int32_t C, int32_t dtype,
                          void* stream);
/* AdaptiveAvgPool2d(1) + flatten (resnets.py:185-186): feat[n][C] fp32 */
int fb_head_pool(const void* a, float* feat, int32_t n_img, int32_t HW, int32_t C, int32_t dtype, void* stream);
/* fc + log_softmax + nll (mean over the chunk) + argmax-correct (training.py:78-80): logits, dlogits [n][classes],
 * loss[g], correct[g].  The loss functions of get_loss_fn (training.py:391-413): label_smoothing s -> LabelSmoothCrossEntropyLoss
 * (modules.py:86-101; s = 0 is CrossEntropyLoss), only_incorrect -> IncorrectCrossEntropyLoss (modules.py:104-119). */
/* Rows with a NEGATIVE label are padding: no loss, no gradient, not counted in the mean or in `correct`. */
int fb_head_loss(const float* feat, const float* fc_w, const float* fc_b, int64_t param_group_stride, const int64_t* labels,
                 float* logits, float* dlogits, float* loss, float* correct, int32_t n_groups, int32_t imgs_per_group,
                 int32_t C, int32_t classes, float label_smoothing, int32_t only_incorrect, void* stream);
/* evaluation helpers (training.py:343-388): BN in eval mode as scale = gamma*rsqrt(running_var+eps), shift = beta - running_mean*scale
 * (torch BatchNorm2d.eval()); and the test-time-flip epilogue (training.py:370-373): outputs = softmax(z_a) + softmax(z_b),
 * loss_sum[0] = sum_n CE(outputs_n, label_n) (cross entropy applied to the summed probabilities, like the reference), correct[0] =
 * #(argmax outputs == label). */
int fb_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, float* scale,
                      float* shift, int32_t C, void* stream);
int fb_head_tta(const float* logits_a, const float* logits_b, const int64_t* labels, int32_t n, int32_t classes, float* ws /* 2n floats */,
                float* loss_sum, float* correct, void* stream);
/* dW_fc[g], db_fc[g] into the gradient arena and d_a = (dlogits @ W)/HW broadcast over the HW pixels */
int fb_head_bwd(const float* feat, const float* dlogits, const float* fc_w, int64_t param_group_stride, float* dfc_w,
                float* dfc_b, int64_t grad_group_stride, void* d_a, int32_t n_groups, int32_t imgs_per_group, int32_t HW,
                int32_t C, int32_t classes, int32_t dtype, void* stream);

/* ---------------------------------------------------------------- multi-tensor (flat fp32 arena) ------------------ */
/* out[g] = sum_i (scale*x[g][i] + add_scale*add[i])^2 (add optional, shared by the groups), deterministic two-stage.
 * ws: n_groups*FB_MT_BLOCKS floats.  (g.pow(2).sum() stack-sum, training.py:162; norm of the finite-difference direction
 * block_strength*g + acc_strength*pre_grads, modules.py:217-223) */
#define FB_MT_BLOCKS 1024
int fb_mt_sqnorm(const float* x, int64_t group_stride, int32_t n_groups, int64_t n, float scale, const float* add, float add_scale,
                 float* out, float* ws, void* stream);
/* running mean over chunks (_stable_mean_accumulation, training.py:45-47): for j: avg += (g[j]-avg)/(counter0+j+1).
 * If sq_out != NULL also writes sq_out[j] = |g[j]|^2 (fused, one pass). */
int fb_mt_accumulate(float* avg, const float* g, int64_t group_stride, int32_t n_groups, int64_t n, int32_t counter0,
                     float* sq_out, float* ws, void* stream);
/* the running mean advanced by n_groups chunks at once from their SUM (fb_conv2d_wgrad_chain + fb_wgrad_reduce):
 * avg += (gsum - n_groups*avg) / (counter0 + n_groups) -- what n_groups steps of fb_mt_accumulate's recurrence amount to (ABI v12) */
int fb_mt_accumulate_sum(float* avg, const float* gsum, int64_t n, int32_t counter0, int32_t n_groups, void* stream);
/* fb_mt_accumulate over [0, n) except up to four ranges [lo_k, hi_k) (multiples of 4 floats; lo == hi: unused), which are neither read nor
 * written and do not enter sq_out: the layers whose mean comes from fb_mt_accumulate_sum (ABI v12) */
int fb_mt_accumulate_skip(float* avg, const float* g, int64_t group_stride, int32_t n_groups, int64_t n, int32_t counter0, float* sq_out,
                          float* ws, int64_t lo0, int64_t hi0, int64_t lo1, int64_t hi1, int64_t lo2, int64_t hi2, int64_t lo3, int64_t hi3,
                          void* stream);
/* eps_n[g] = eps / sqrt(vnorm2[g]);  theta_out[g] = theta0 + (sign*eps_n[g]) * (s*g[g] + acc*pre)   (modules.py:217-226;
 * pre = the pre-computed full gradient of the acc_strength pre-pass, training.py:128-142, NULL without it) */
int fb_mt_fd_perturb(const float* theta0, const float* g, int64_t group_stride, int32_t n_groups, int64_t n, float s,
                     float eps, float sign, const float* vnorm2, float* eps_n, const float* pre, float acc, float* theta_out,
                     void* stream);
/* vhp = (ga - gb)/eps_n[g]; gt = g + cf*vhp; avg += (gt-avg)/(counter0+j+1)   (modules.py:232-240 + training.py:45-47) */
int fb_mt_fd_combine_accumulate(float* avg, const float* g, const float* ga, const float* gb, int64_t group_stride,
                                int32_t n_groups, int64_t n, const float* eps_n, float cf, int32_t counter0, void* stream);
/* The two halves of fb_mt_fd_combine_accumulate as separate steps, for hyp.batch_clip (training.py:166-167 clips the REGULARISED
 * chunk gradient before it is averaged):  fb_mt_fd_combine: g[j] += cf*(ga[j]-gb[j])/eps_n[j] in place;
 * fb_mt_chunk_clip (_clip_gradient_list, training/utils.py:4-19, p = 2): with norm = sqrt(sq[j]): if norm > clip then
 * g[j] *= clip/(norm + 1e-6) and clipped[j] = 1 else clipped[j] = 0. */
int fb_mt_fd_combine(float* g, const float* ga, const float* gb, int64_t group_stride, int32_t n_groups, int64_t n, const float* eps_n,
                     float cf, void* stream);
int fb_mt_chunk_clip(float* g, int64_t group_stride, int32_t n_groups, int64_t n, const float* sq, float clip, float* clipped, void* stream);
/* out[0] = |a|^2, out[1] = |b|^2 over [0,n) (grad norm for clipping, training.py:202-204; param_norm, :92) */
int fb_mt_norms2(const float* a, const float* b, int64_t n, float* out, float* ws, void* stream);
/* clip (training.py:206-207) + torch.optim.SGD step with weight decay, momentum, dampening, Nesterov.
 * gnorm2: device scalar |grad|^2 (whole vector); grad_clip < 0 disables clipping.  Operates on [0,n) of the
 * pointers given (callers pass shard offsets for the sharded multi-GPU update).  grad is overwritten with the
 * clipped gradient (the closure contract exposes p.grad). */
int fb_mt_clip_sgd(float* theta, float* grad, float* mom, int64_t n, const float* gnorm2, float grad_clip, float lr,
                   float weight_decay, float momentum, float dampening, int32_t nesterov, int32_t first_step, void* stream);
/* y = a*x (+ y*b)  flat helpers used by the sharded path (scale local mean by K_r/K) */
int fb_mt_scale(float* x, int64_t n, float a, void* stream);
/* SAM around the closure (additional_optimizers/sam.py:56-82).  ascent: with g_c = grad * clip coefficient (as
 * fb_mt_clip_sgd, training.py:198-206; grad_clip < 0: none) e_w = g_c * rho / (|g_c| + 1e-12), theta += e_w; restore: theta -= e_w. */
int fb_mt_sam_ascent(float* theta, const float* grad, float* e_w, int64_t n, const float* gnorm2, float grad_clip, float rho, void* stream);
int fb_mt_sam_restore(float* theta, const float* e_w, int64_t n, void* stream);
/* gradient-modification options that are off by default (training.py:187-211):
 *   fb_mt_absmax2  : out[0] = (max|a_i|)^2 -- the L-infinity clip norm (grad_clip_norm=inf, training.py:199-200) in the slot of |a|^2
 *   fb_mt_norm_bias: external norm bias on ONE parameter tensor (training.py:188-196); pnorm2 = device |theta|^2 of all parameters
 * and the model EMA used for evaluation (training/utils.py:22-29): ema = momentum*ema + one_minus*src. */
int fb_mt_absmax2(const float* a, int64_t n, float* out, float* ws, void* stream);
/* out[0] = ((sum |a_i|^p)^(1/p))^2 -- the clip norm for grad_clip_norm = p other than 2 / inf (training.py:201-204) */
int fb_mt_pnorm2(const float* a, int64_t n, float p, float* out, float* ws, void* stream);
int fb_mt_norm_bias(float* grad, const float* theta, int64_t n, const float* pnorm2, float strength, float bias, int32_t norm_type, void* stream);
int fb_mt_ema(float* ema, const float* src, int64_t n, float momentum, float one_minus, void* stream);
/* gradient noise of the closure (training.py:212-215; it acts on the clipped gradient, so the clip is applied in place first):
 *   fb_mt_clip_scale : grad *= grad_clip / (|grad| + 1e-6) if |grad| > grad_clip   (the in-place form of fb_mt_clip_sgd's clip)
 *   fb_mt_grad_noise : mode 0: grad += strength * noise ; mode 1: grad *= 1 + strength * noise   (noise drawn by the caller) */
int fb_mt_clip_scale(float* grad, int64_t n, const float* gnorm2, float grad_clip, void* stream);
int fb_mt_grad_noise(float* grad, const float* noise, int64_t n, float strength, int32_t mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FB_ENGINE_H */
