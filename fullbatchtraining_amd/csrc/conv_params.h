// Kernel-argument block shared by the implicit-GEMM convolution kernels.
#pragma once
struct ConvParams {
    const char* src; const char* wgt; char* dst; const char* addend; float* stat;
    int n_img, Hs, Ws, Cs, Hd, Wd, Cd;
    int R, S, stride, pad, mode;
    int qH, qW, os, ss, M;
    int imgs_per_wset; long long wset_stride_bytes;
    int addend_mode, n_mblocks;
    const char* zeros;   // >= 128 bytes of zeros in device memory (source of padding rows for LDS-direct loads)
    const float* amax_src; const float* amax_wgt;   // f32h (fp16x2 split): largest magnitudes per chunk of src / per weight set (fb_absmax)
    int amax_imgs;                                   // images per entry of amax_src
    const unsigned char* addend_mask;                // addend_mode 1: the addend counts only where its ReLU bit is set (fb_conv_args.addend_mask; 1 byte per 16-byte vector)
};
const void* fb_zero_page();                                                   // runtime.cpp
int fb_launch_igemm_glds(const ConvParams& p, int classes, int dtype, hipStream_t st);   // conv_igemm_glds.hip
int fb_igemm_glds_fits(const ConvParams& p, int dtype);                                  // conv_igemm_glds.hip: would fb_launch_igemm_glds take these sizes?
