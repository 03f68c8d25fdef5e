// Native launch executor: command lists of library calls, recorded once and replayed with one host call.
//
// The reference drives its hot loop from Python, one ATen dispatch per operator (fullbatch/training/training.py:144-174: a Python loop
// over chunks around model forward / autograd.grad / foreach ops).  Here one chunk group's forward + backward is ~430 (ResNet-18) to
// ~6000 (ResNet-152) asynchronous launches; through ctypes every launch costs 6-20 us of interpreter time, which the GPU hides for
// ResNet-18 and does not for ResNet-152 (host-bound: 990 ms of enqueue for a 1070 ms step).  The sequence is static -- same entry
// points, same arguments, same buffers every step -- so the host side records it once (every fb_* call with its argument words and
// the index of the stream it went to, every cross-stream event record / wait) and replays it natively: one ctypes call per chunk
// group, ~1-3 us per launch.  Not a hipGraph: the kernels stay ordinary stream launches (events, profiling brackets and rocprofv3
// see them as before), the list just removes the interpreter from the loop.
#include <stdlib.h>

#include <memory>
#include <utility>
#include <vector>

#include "common.h"

namespace {

template <typename T> inline T word_as(uint64_t w) {
    T v;
    memcpy(&v, &w, sizeof(T));
    return v;
}

using Thunk = int (*)(const uint64_t*, void*);

// Bind<decltype(&f), &f>::call(words, stream): f(words[0] as A0, ..., words[N-2] as A(N-2), stream)
template <typename F, F f> struct Bind;
template <typename... A, int (*f)(A...)> struct Bind<int (*)(A...), f> {
    static constexpr int N = sizeof...(A);
    template <size_t I, typename T> static T arg(const uint64_t* w, void* st) {
        if constexpr (I == N - 1) return (T)st;
        else return word_as<T>(w[I]);
    }
    template <size_t... I> static int run(const uint64_t* w, void* st, std::index_sequence<I...>) { return f(arg<I, A>(w, st)...); }
    static int call(const uint64_t* w, void* st) { return run(w, st, std::index_sequence_for<A...>{}); }
};

struct Entry { const char* name; Thunk fn; int nargs; };
#define FB_ENTRY(name) {#name, &Bind<decltype(&name), &name>::call, Bind<decltype(&name), &name>::N}
// every asynchronous entry point whose arguments are scalars, device pointers or ONE leading argument struct (copied into the list)
const Entry kEntries[] = {
    FB_ENTRY(fb_conv2d), FB_ENTRY(fb_conv2d_wgrad), FB_ENTRY(fb_absmax), FB_ENTRY(fb_wgrad_reduce), FB_ENTRY(fb_weight_prep),
    FB_ENTRY(fb_bn_fwd_finalize), FB_ENTRY(fb_bn_apply), FB_ENTRY(fb_bn_running_update), FB_ENTRY(fb_bn_bwd_reduce),
    FB_ENTRY(fb_bn_bwd_finalize), FB_ENTRY(fb_bn_bwd_apply), FB_ENTRY(fb_avgpool2_fwd), FB_ENTRY(fb_maxpool3s2_fwd),
    FB_ENTRY(fb_maxpool3s2_bwd), FB_ENTRY(fb_head_pool), FB_ENTRY(fb_head_loss), FB_ENTRY(fb_head_bwd), FB_ENTRY(fb_mt_sqnorm),
    FB_ENTRY(fb_mt_accumulate), FB_ENTRY(fb_mt_fd_perturb), FB_ENTRY(fb_mt_fd_combine_accumulate), FB_ENTRY(fb_mt_fd_combine),
    FB_ENTRY(fb_mt_chunk_clip), FB_ENTRY(fb_bn_eval_coeffs), FB_ENTRY(fb_mt_norms2), FB_ENTRY(fb_mt_clip_sgd), FB_ENTRY(fb_mt_scale), FB_ENTRY(fb_bn_bwd_fused), FB_ENTRY(fb_bn_bwd_reduce2), FB_ENTRY(fb_bn_bwd_apply2),
    FB_ENTRY(fb_conv2d_wgrad_chain), FB_ENTRY(fb_mt_accumulate_sum), FB_ENTRY(fb_mt_accumulate_skip),
    FB_ENTRY(fb_maxpool3s2_fwd_idx), FB_ENTRY(fb_maxpool3s2_bwd_idx),
};
constexpr int kNumEntries = sizeof(kEntries) / sizeof(kEntries[0]);

enum { CMD_CALL = 0, CMD_EVENT_RECORD = 1, CMD_EVENT_WAIT = 2 };
struct Cmd { int32_t kind, fn, stream, ev; uint32_t off, n; };
struct CmdList {
    std::vector<Cmd> cmds;
    std::vector<uint64_t> words;
    std::vector<std::unique_ptr<uint64_t[]>> blobs;   // copies of argument structs (fb_conv_args / fb_wgrad_args)
};

std::vector<hipEvent_t> g_events;   // process-wide event table: ids are shared by eager calls and by every list

}  // namespace

extern "C" int32_t fb_cmd_fn_id(const char* name) {
    for (int i = 0; i < kNumEntries; ++i)
        if (strcmp(kEntries[i].name, name) == 0) return i;
    return -1;
}
extern "C" int32_t fb_cmd_fn_nargs(int32_t fn) { return fn >= 0 && fn < kNumEntries ? kEntries[fn].nargs : -1; }

// ---- events (timing disabled): cross-stream ordering of eager launches and of recorded lists alike ------------------------------
extern "C" int32_t fb_event_new(void) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        snprintf(fb_err_buf, sizeof(fb_err_buf), "fb_event_new: hipEventCreateWithFlags failed");
        return -1;
    }
    g_events.push_back(e);
    return (int32_t)g_events.size() - 1;
}
extern "C" int32_t fb_event_count(void) { return (int32_t)g_events.size(); }
extern "C" int fb_event_record(int32_t ev, void* stream) {
    if (ev < 0 || ev >= (int32_t)g_events.size()) FB_FAIL(FB_ERR_ARG, "fb_event_record: unknown event %d", ev);
    if (hipEventRecord(g_events[ev], (hipStream_t)stream) != hipSuccess) FB_FAIL(FB_ERR_LAUNCH, "fb_event_record: hipEventRecord failed");
    return FB_OK;
}
extern "C" int fb_event_wait(int32_t ev, void* stream) {
    if (ev < 0 || ev >= (int32_t)g_events.size()) FB_FAIL(FB_ERR_ARG, "fb_event_wait: unknown event %d", ev);
    if (hipStreamWaitEvent((hipStream_t)stream, g_events[ev], 0) != hipSuccess) FB_FAIL(FB_ERR_LAUNCH, "fb_event_wait: hipStreamWaitEvent failed");
    return FB_OK;
}

// ---- command lists ---------------------------------------------------------------------------------------------------------------
extern "C" void* fb_cmdlist_create(void) { return new CmdList(); }
extern "C" void fb_cmdlist_destroy(void* cl) { delete (CmdList*)cl; }
extern "C" int64_t fb_cmdlist_size(const void* cl) { return cl ? (int64_t)((const CmdList*)cl)->cmds.size() : 0; }

// words: one 64-bit word per argument of the entry point EXCEPT the trailing stream (pointers and integers zero/sign-extended, float
// and double as their bit patterns in the low bytes).  blob (optional): the argument struct words[0] points to; it is copied and
// words[0] redirected to the copy.
extern "C" int fb_cmdlist_add_call(void* handle, int32_t fn, const uint64_t* words, int32_t n_words, int32_t stream_idx, const void* blob,
                                   int32_t blob_bytes) {
    CmdList* cl = (CmdList*)handle;
    if (!cl || fn < 0 || fn >= kNumEntries) FB_FAIL(FB_ERR_ARG, "fb_cmdlist_add_call: bad list or function id %d", fn);
    if (n_words != kEntries[fn].nargs - 1) FB_FAIL(FB_ERR_ARG, "fb_cmdlist_add_call: %s takes %d words, got %d", kEntries[fn].name, kEntries[fn].nargs - 1, n_words);
    Cmd c{CMD_CALL, fn, stream_idx, -1, (uint32_t)cl->words.size(), (uint32_t)n_words};
    cl->words.insert(cl->words.end(), words, words + n_words);
    if (blob && blob_bytes > 0) {
        const size_t nw = ((size_t)blob_bytes + 7) / 8;
        cl->blobs.emplace_back(new uint64_t[nw]);
        memcpy(cl->blobs.back().get(), blob, blob_bytes);
        cl->words[c.off] = (uint64_t)(uintptr_t)cl->blobs.back().get();
    }
    cl->cmds.push_back(c);
    return FB_OK;
}
extern "C" int fb_cmdlist_add_event(void* handle, int32_t kind, int32_t ev, int32_t stream_idx) {
    CmdList* cl = (CmdList*)handle;
    if (!cl || (kind != CMD_EVENT_RECORD && kind != CMD_EVENT_WAIT)) FB_FAIL(FB_ERR_ARG, "fb_cmdlist_add_event: bad list or kind %d", kind);
    if (ev < 0 || ev >= (int32_t)g_events.size()) FB_FAIL(FB_ERR_ARG, "fb_cmdlist_add_event: unknown event %d", ev);
    cl->cmds.push_back(Cmd{kind, -1, stream_idx, ev, 0, 0});
    return FB_OK;
}

// Issues every command in recorded order; streams[i] is the hipStream_t behind stream index i of the recording.
extern "C" int fb_cmdlist_replay(const void* handle, void* const* streams, int32_t n_streams) {
    const CmdList* cl = (const CmdList*)handle;
    if (!cl || !streams) FB_FAIL(FB_ERR_ARG, "fb_cmdlist_replay: null list or stream table");
    const size_t n = cl->cmds.size();
    for (size_t i = 0; i < n; ++i) {
        const Cmd& c = cl->cmds[i];
        if (c.stream < 0 || c.stream >= n_streams) FB_FAIL(FB_ERR_ARG, "fb_cmdlist_replay: command %zu wants stream %d of %d", i, c.stream, n_streams);
        void* st = streams[c.stream];
        if (c.kind == CMD_CALL) {
            const int rc = kEntries[c.fn].fn(cl->words.data() + c.off, st);
            if (rc != FB_OK) {
                char inner[400];
                snprintf(inner, sizeof(inner), "%s", fb_err_buf);
                FB_FAIL(rc, "fb_cmdlist_replay: command %zu (%s) failed: %s", i, kEntries[c.fn].name, inner);
            }
        } else if (c.kind == CMD_EVENT_RECORD) {
            if (hipEventRecord(g_events[c.ev], (hipStream_t)st) != hipSuccess) FB_FAIL(FB_ERR_LAUNCH, "fb_cmdlist_replay: hipEventRecord failed at command %zu", i);
        } else {
            if (hipStreamWaitEvent((hipStream_t)st, g_events[c.ev], 0) != hipSuccess) FB_FAIL(FB_ERR_LAUNCH, "fb_cmdlist_replay: hipStreamWaitEvent failed at command %zu", i);
        }
    }
    return FB_OK;
}
