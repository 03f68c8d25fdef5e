// Error string + optional per-kernel-class timing with HIP events on the launch stream (used by bench.py's roofline leg).
#include <stdlib.h>
#include <vector>

#include "common.h"
#include "profile.h"
#include "conv_params.h"

thread_local char fb_err_buf[512] = "";
extern "C" const char* fb_last_error_string(void) { return fb_err_buf; }
extern "C" int fb_abi_version(void) { return 13; }

// ---- workspace sizes (floats) ------------------------------------------------------------------------------------------
extern "C" int64_t fb_ws_conv_stat_floats(const fb_conv_args* a) {
    return a ? 2 * (((int64_t)a->n_img * a->Hd * a->Wd + 127) / 128) * a->Cd : 0;
}
extern "C" int64_t fb_ws_wgrad_slab_floats(const fb_wgrad_args* a) {
    if (!a || a->imgs_per_group <= 0) return 0;
    return (int64_t)(a->n_img / a->imgs_per_group) * a->split_k * a->Cd * a->R * a->S * a->Cs;
}
extern "C" int64_t fb_ws_bn_partial_floats(int64_t n_pixels, int32_t C) { return 2 * ((n_pixels + 127) / 128) * C; }
extern "C" int64_t fb_ws_mt_floats(int32_t n_groups) { return (int64_t)(n_groups > 2 ? n_groups : 2) * FB_MT_BLOCKS; }

// CUs the persistent kernels size their grids for: the device's, minus FB_CU_RESERVE (default 0).  With one or two resident workgroups of a
// persistent convolution on EVERY CU a kernel of another stream (the gradient exchange's RCCL kernels under the last backward pass) only gets
// a slot when a whole launch ends; a reserve leaves that many CUs' worth of slots open (measured: bench.py `exchange`).
// FB_EXPERIMENTAL=1 (read once): the switches that turn ON a kernel form which lost its same-box A/B (profiles/r*_notes.md) only act together with it -- the
// default dispatch documented in DESIGN.md section 4 is what runs otherwise.  (Switches that fall BACK to an older established kernel -- FB_DISABLE_*, FB_C1S_PIPE=0,
// FB_C1G=0, FB_H4_COMPACT=0 ... -- are A/B switches between shipped forms and act on their own.)
bool fb_experimental() {
    static const bool on = getenv("FB_EXPERIMENTAL") != nullptr && atoi(getenv("FB_EXPERIMENTAL")) != 0;
    return on;
}
static const char* fb_exp_getenv(const char* name) { return fb_experimental() ? getenv(name) : nullptr; }
const char* fb_getenv_experimental(const char* name) { return fb_exp_getenv(name); }
int fb_persistent_cus() {
    // per DEVICE (a process may drive several: the first caller's device must not size the grids of all of them); hipGetDevice is a
    // thread-local read, so the lookup stays out of the launch path's cost
    static int per_dev[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int n = per_dev[dev];
    if (n == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        const char* r = getenv("FB_CU_RESERVE");
        const int reserve = r ? atoi(r) : 0;
        n = cus - (reserve > 0 ? reserve : 0);
        if (n < 8) n = 8;
        per_dev[dev] = n;
    }
    return n;
}

bool fb_f32_split_enabled() {
    static const bool on = getenv("FB_F32_EXACT") == nullptr || getenv("FB_F32_EXACT")[0] == '0';
    return on;
}

namespace {
struct Pair { hipEvent_t a, b; int cls; int32_t info[FB_PROF_INFO]; };
bool g_on = false;
std::vector<Pair> g_pool;
size_t g_used = 0;
long long g_dropped[FB_PROF_CLASSES] = {0};
}  // namespace

// 4 KiB of zeros per process, allocated on first use (the one internal allocation of the library): padding source for
// LDS-direct (global_load_lds) tile loads, which cannot synthesise zeros.
const void* fb_zero_page() {
    static void* page = nullptr;
    if (!page) {
        if (hipMalloc(&page, 4096) != hipSuccess) { page = nullptr; return nullptr; }
        hipMemset(page, 0, 4096);
    }
    return page;
}

extern "C" int fb_profile_enable(int on, int capacity) {
    if (on && g_pool.size() < (size_t)capacity) {
        const size_t old = g_pool.size();
        g_pool.resize(capacity);
        for (size_t i = old; i < g_pool.size(); ++i) {
            if (hipEventCreate(&g_pool[i].a) != hipSuccess || hipEventCreate(&g_pool[i].b) != hipSuccess)
                FB_FAIL(FB_ERR_LAUNCH, "fb_profile_enable: hipEventCreate failed");
        }
    }
    g_on = on != 0;
    g_used = 0;
    for (int c = 0; c < FB_PROF_CLASSES; ++c) g_dropped[c] = 0;
    return FB_OK;
}

int fb_prof_begin(int cls, hipStream_t st, const int32_t* info) {
    if (!g_on) return -1;
    if (g_used >= g_pool.size()) { g_dropped[cls]++; return -1; }
    const int id = (int)g_used++;
    g_pool[id].cls = cls;
    for (int i = 0; i < FB_PROF_INFO; ++i) g_pool[id].info[i] = info ? info[i] : 0;
    hipEventRecord(g_pool[id].a, st);
    return id;
}
void fb_prof_kernel(int id, int kernel) {
    if (id >= 0) g_pool[id].info[FB_PROF_INFO - 1] = kernel;
}
void fb_prof_end(int id, hipStream_t st) {
    if (id >= 0) hipEventRecord(g_pool[id].b, st);
}

// Per-launch records of everything recorded since the last fb_profile_read (which resets; call this one first): row i of `info`
// ([cap][FB_PROF_INFO + 1] int32) = {class, the FB_PROF_INFO shape words the entry point filed (fb_engine.h), the last one = kernel id},
// ms[i] = elapsed milliseconds.  Returns the number of rows written (<= cap), negative on error; blocks until the launches finish.
extern "C" int64_t fb_profile_read_launches(int32_t* info, float* ms, int64_t cap) {
    int64_t n = 0;
    for (size_t i = 0; i < g_used && n < cap; ++i, ++n) {
        if (hipEventSynchronize(g_pool[i].b) != hipSuccess) FB_FAIL(FB_ERR_LAUNCH, "fb_profile_read_launches: event sync failed");
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_pool[i].a, g_pool[i].b) != hipSuccess) FB_FAIL(FB_ERR_LAUNCH, "fb_profile_read_launches: elapsed failed");
        info[n * (FB_PROF_INFO + 1)] = g_pool[i].cls;
        for (int k = 0; k < FB_PROF_INFO; ++k) info[n * (FB_PROF_INFO + 1) + 1 + k] = g_pool[i].info[k];
        ms[n] = t;
    }
    return n;
}

// Sums elapsed ms and launch counts per class for everything recorded since the last read; blocks until those launches finish.
extern "C" int fb_profile_read(double* ms, int64_t* launches, int64_t* dropped) {
    for (int c = 0; c < FB_PROF_CLASSES; ++c) { ms[c] = 0.0; launches[c] = 0; dropped[c] = g_dropped[c]; g_dropped[c] = 0; }
    for (size_t i = 0; i < g_used; ++i) {
        if (hipEventSynchronize(g_pool[i].b) != hipSuccess) FB_FAIL(FB_ERR_LAUNCH, "fb_profile_read: event sync failed");
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_pool[i].a, g_pool[i].b) != hipSuccess) FB_FAIL(FB_ERR_LAUNCH, "fb_profile_read: elapsed failed");
        ms[g_pool[i].cls] += t;
        launches[g_pool[i].cls] += 1;
    }
    g_used = 0;
    return FB_OK;
}
