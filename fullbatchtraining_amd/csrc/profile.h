// Kernel classes timed by the optional HIP-event profiler (runtime.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
enum { FB_PROF_IGEMM_FWD = 0, FB_PROF_IGEMM_DGRAD = 1, FB_PROF_WGRAD = 2, FB_PROF_BN_APPLY = 3, FB_PROF_BN_BWD_REDUCE = 4, FB_PROF_BN_BWD_APPLY = 5,
       FB_PROF_BN_BWD_FUSED = 6, FB_PROF_CLASSES = 7 };
// shape words filed with a launch (fb_profile_read_launches): convolutions {n_img, Hs, Ws, Cs, Hd, Wd, Cd, R, stride, flags, kernel};
// BatchNorm passes {pixels / 128, C, pixels_per_group / 128, dtype, residual?, mask?, dy_out?, pooled?, 0, 0, kernel}
enum { FB_PROF_INFO = 11 };
// which kernel served a convolution launch (last shape word)
enum { FB_K_IGEMM_V1 = 1, FB_K_IGEMM_GLDS = 2, FB_K_HALO4 = 3, FB_K_HALO5 = 4, FB_K_S2_DGRAD_QUAD = 5, FB_K_CONV1X1_K32 = 6, FB_K_CONV1X1_STREAM = 7, FB_K_S2_FWD = 8, FB_K_CONV1X1_PIPE = 9, FB_K_CONV1X1_GEMM = 10,
       FB_K_WGRAD_GENERIC = 16, FB_K_WGRAD3X3_V1 = 17, FB_K_WGRAD3X3_V2 = 18, FB_K_WGRAD1X1 = 19 };
int fb_prof_begin(int cls, hipStream_t st, const int32_t* info = nullptr);
void fb_prof_kernel(int id, int kernel);
void fb_prof_end(int id, hipStream_t st);
