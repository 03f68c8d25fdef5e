// Kernel classes timed by the optional HIP-event profiler (runtime.cpp).
#pragma once
#include <hip/hip_runtime.h>
enum { FB_PROF_IGEMM_FWD = 0, FB_PROF_IGEMM_DGRAD = 1, FB_PROF_WGRAD = 2, FB_PROF_CLASSES = 3 };
int fb_prof_begin(int cls, hipStream_t st);
void fb_prof_end(int id, hipStream_t st);
