"""ctypes binding of ``libfbengine.so`` (C ABI declared in ``include/fb_engine.h``).

There is no CPU fallback: importing this module without the built library, or calling into it without a GPU, raises.
Wrappers take ``torch`` tensors only to read ``data_ptr()`` -- memory and streams are PyTorch-ROCm plumbing.
"""
import ctypes as C
import os
import struct

import torch

FB_F32, FB_BF16 = 0, 1
EXPECTED_ABI = 13         # fb_abi_version() the ctypes structs / signatures below were written for
MT_BLOCKS = 1024
_LIB_PATH = os.environ.get("FB_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libfbengine.so")     # (FB_LIB_PATH: A/B builds, tools/build_variant.py)

c_void_p, c_int, c_i64, c_float, c_double = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double


class ConvArgs(C.Structure):
    _fields_ = [("src", c_void_p), ("wgt", c_void_p), ("dst", c_void_p), ("addend", c_void_p), ("stat_partial", c_void_p),
                ("n_img", c_int), ("Hs", c_int), ("Ws", c_int), ("Cs", c_int), ("Hd", c_int), ("Wd", c_int), ("Cd", c_int),
                ("R", c_int), ("S", c_int), ("stride", c_int), ("pad", c_int), ("mode", c_int),
                ("imgs_per_wset", c_int), ("wset_stride", c_i64), ("addend_mode", c_int), ("dtype", c_int), ("addend_mask", c_void_p),
                ("bst_x", c_void_p), ("bst_mask", c_void_p), ("amax_src", c_void_p), ("amax_wgt", c_void_p), ("amax_imgs", c_int)]


class WgradArgs(C.Structure):
    _fields_ = [("x", c_void_p), ("dy", c_void_p), ("dw_partial", c_void_p),
                ("n_img", c_int), ("Hs", c_int), ("Ws", c_int), ("Cs", c_int), ("Hd", c_int), ("Wd", c_int), ("Cd", c_int),
                ("R", c_int), ("S", c_int), ("stride", c_int), ("pad", c_int),
                ("imgs_per_group", c_int), ("split_k", c_int), ("dtype", c_int), ("group_stride", c_i64), ("amax_x", c_void_p), ("amax_dy", c_void_p),
                ("bn_x", c_void_p), ("bn_mask", c_void_p), ("bn_coef", c_void_p)]


_SIGS = {
    "fb_conv2d": [C.POINTER(ConvArgs), c_void_p],
    "fb_conv2d_wgrad": [C.POINTER(WgradArgs), c_void_p],
    "fb_absmax": [c_void_p, c_i64, c_int, c_i64, c_int, c_void_p, c_void_p],
    "fb_wgrad_reduce": [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "fb_weight_prep": [c_void_p, c_i64, c_i64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "fb_bn_fwd_finalize": [c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_void_p, c_i64, c_float, c_void_p, c_void_p, c_int,
                           c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "fb_bn_apply": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_i64, c_int, c_void_p, c_void_p, c_int,
                    c_int, c_void_p, c_void_p, c_void_p],
    "fb_bn_running_update": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_void_p, c_int, c_int, c_float, c_void_p],
    "fb_bn_bwd_reduce": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_i64, c_int, c_i64, c_int, c_void_p],
    "fb_bn_bwd_finalize": [c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                           c_i64, c_void_p, c_int, c_void_p],
    "fb_bn_bwd_apply": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_int, c_void_p, c_void_p, c_void_p],
    "fb_bn_bwd_reduce2": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_i64, c_int, c_i64, c_int,
                          c_void_p],
    "fb_bn_bwd_apply2": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_int, c_void_p],
    "fb_bn_bwd_fused": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p,
                        c_i64, c_int, c_i64, c_double, c_int, c_void_p, c_void_p, c_void_p],
    "fb_stem_patches": [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                        C.POINTER(c_float), c_int, c_void_p],
    "fb_avgpool2_fwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "fb_maxpool3s2_fwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "fb_maxpool3s2_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "fb_maxpool3s2_fwd_idx": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "fb_maxpool3s2_bwd_idx": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "fb_head_pool": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "fb_head_loss": [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                     c_float, c_int, c_void_p],
    "fb_head_bwd": [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                    c_void_p],
    "fb_mt_sqnorm": [c_void_p, c_i64, c_int, c_i64, c_float, c_void_p, c_float, c_void_p, c_void_p, c_void_p],
    "fb_mt_accumulate": [c_void_p, c_void_p, c_i64, c_int, c_i64, c_int, c_void_p, c_void_p, c_void_p],
    "fb_mt_fd_perturb": [c_void_p, c_void_p, c_i64, c_int, c_i64, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                         c_void_p],
    "fb_mt_fd_combine_accumulate": [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_void_p, c_float, c_int, c_void_p],
    "fb_mt_fd_combine": [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_void_p, c_float, c_void_p],
    "fb_mt_chunk_clip": [c_void_p, c_i64, c_int, c_i64, c_void_p, c_float, c_void_p, c_void_p],
    "fb_bn_eval_coeffs": [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p],
    "fb_head_tta": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "fb_mt_norms2": [c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p],
    "fb_mt_clip_sgd": [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_float, c_float, c_float, c_float, c_float, c_int, c_int, c_void_p],
    "fb_mt_scale": [c_void_p, c_i64, c_float, c_void_p],
    "fb_mt_sam_ascent": [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_float, c_float, c_void_p],
    "fb_mt_sam_restore": [c_void_p, c_void_p, c_i64, c_void_p],
    "fb_mt_absmax2": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p],
    "fb_mt_pnorm2": [c_void_p, c_i64, c_float, c_void_p, c_void_p, c_void_p],
    "fb_mt_norm_bias": [c_void_p, c_void_p, c_i64, c_void_p, c_float, c_float, c_int, c_void_p],
    "fb_mt_ema": [c_void_p, c_void_p, c_i64, c_float, c_float, c_void_p],
    "fb_mt_clip_scale": [c_void_p, c_i64, c_void_p, c_float, c_void_p],
    "fb_mt_grad_noise": [c_void_p, c_void_p, c_i64, c_float, c_int, c_void_p],
    "fb_conv2d_wgrad_chain": [C.POINTER(WgradArgs), c_int, c_void_p, c_void_p, c_void_p],
    "fb_mt_accumulate_sum": [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p],
    "fb_mt_accumulate_skip": [c_void_p, c_void_p, c_i64, c_int, c_i64, c_int, c_void_p, c_void_p] + [c_i64] * 8 + [c_void_p],
}
EXPORTS = tuple(_SIGS) + ("fb_last_error_string", "fb_abi_version", "fb_profile_enable", "fb_profile_read", "fb_ws_conv_stat_floats",
                          "fb_ws_wgrad_slab_floats", "fb_ws_bn_partial_floats", "fb_ws_mt_floats", "fb_bn_bwd_reduce_rows", "fb_conv_masked_addend_supported", "fb_conv_bwd_stat_supported",
                          "fb_bn_apply_can_pool", "fb_ws_bn_amax_floats", "fb_profile_read_launches", "fb_cmd_fn_id", "fb_cmd_fn_nargs", "fb_event_new", "fb_event_count", "fb_bn_bwd_fused_supported", "fb_ws_bn_bwd_fused_floats", "fb_ws_bn_bwd_fused_ints", "fb_wgrad_bn_fused_supported", "fb_wgrad_chain_supported",
                          "fb_event_record", "fb_event_wait", "fb_cmdlist_create", "fb_cmdlist_destroy", "fb_cmdlist_size", "fb_cmdlist_add_call",
                          "fb_cmdlist_add_event", "fb_cmdlist_replay")
PROF_CLASSES = ("igemm_fwd", "igemm_dgrad", "wgrad", "bn_apply", "bn_bwd_reduce", "bn_bwd_apply", "bn_bwd_fused")
PROF_INFO = 11
PROF_KERNELS = {0: "?", 1: "conv_igemm_kernel (register-staged)", 2: "conv_igemm_v3_kernel", 3: "conv3x3s1_halo4_kernel", 4: "conv3x3s1_c64_halo5_kernel",
                5: "conv3x3s2_dgrad_quad_kernel", 6: "conv1x1_k32_kernel", 7: "conv1x1_stream_kernel", 8: "conv3x3s2_fwd_kernel", 9: "conv1x1_pipe_kernel", 10: "conv1x1_gemm_kernel",
                16: "conv_wgrad_kernel", 17: "conv_wgrad3x3_kernel", 18: "conv_wgrad3x3_v2_kernel", 19: "conv_wgrad1x1_kernel"}


def profile_enable(on, capacity=32768):
    lib = load()
    lib.fb_profile_enable.argtypes, lib.fb_profile_enable.restype = [c_int, c_int], c_int
    if lib.fb_profile_enable(1 if on else 0, capacity) != 0:
        raise EngineError(lib.fb_last_error_string().decode())


def profile_read():
    """-> {class: (ms, launches, dropped)} for launches recorded since the previous read (synchronises on them)."""
    lib = load()
    lib.fb_profile_read.argtypes = [C.POINTER(c_double), C.POINTER(c_i64), C.POINTER(c_i64)]
    k = len(PROF_CLASSES)
    ms, n, d = (c_double * k)(), (c_i64 * k)(), (c_i64 * k)()
    if lib.fb_profile_read(ms, n, d) != 0:
        raise EngineError(lib.fb_last_error_string().decode())
    return {k: (ms[i], n[i], d[i]) for i, k in enumerate(PROF_CLASSES)}



def profile_read_launches(cap=1 << 18):
    """-> list of (class name, shape words tuple, ms) for every launch recorded since the previous ``profile_read`` (call this first:
    ``profile_read`` resets the records)."""
    lib = load()
    lib.fb_profile_read_launches.argtypes, lib.fb_profile_read_launches.restype = [C.POINTER(c_int), C.POINTER(c_float), c_i64], c_i64
    info, ms = (c_int * (cap * (PROF_INFO + 1)))(), (c_float * cap)()
    n = lib.fb_profile_read_launches(info, ms, cap)
    if n < 0:
        raise EngineError(lib.fb_last_error_string().decode())
    w = PROF_INFO + 1
    return [(PROF_CLASSES[info[i * w]], tuple(info[i * w + 1:(i + 1) * w]), ms[i]) for i in range(n)]


_lib = None


class EngineError(RuntimeError):
    pass


def load():
    """Load the shared library (no compute).  Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.isfile(_LIB_PATH):
            raise EngineError(f"{_LIB_PATH} is missing: run `python -m fullbatchtraining_amd.build` (needs hipcc). "
                              "The engine has no CPU or PyTorch fallback.")
        lib = C.CDLL(_LIB_PATH)
        for name, sig in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = sig, c_int
        lib.fb_last_error_string.restype = C.c_char_p
        lib.fb_abi_version.restype = c_int
        if lib.fb_abi_version() != EXPECTED_ABI:           # a stale in-tree .so would read the argument structs with another layout
            raise EngineError(f"{_LIB_PATH} has ABI version {lib.fb_abi_version()}, this package expects {EXPECTED_ABI}: rebuild it with "
                              "`python -m fullbatchtraining_amd.build --force`")
        lib.fb_ws_conv_stat_floats.argtypes, lib.fb_ws_conv_stat_floats.restype = [C.POINTER(ConvArgs)], c_i64
        lib.fb_ws_wgrad_slab_floats.argtypes, lib.fb_ws_wgrad_slab_floats.restype = [C.POINTER(WgradArgs)], c_i64
        lib.fb_ws_bn_partial_floats.argtypes, lib.fb_ws_bn_partial_floats.restype = [c_i64, c_int], c_i64
        lib.fb_ws_mt_floats.argtypes, lib.fb_ws_mt_floats.restype = [c_int], c_i64
        lib.fb_bn_bwd_reduce_rows.argtypes, lib.fb_bn_bwd_reduce_rows.restype = [c_i64, c_i64], c_int
        lib.fb_conv_masked_addend_supported.argtypes, lib.fb_conv_masked_addend_supported.restype = [C.POINTER(ConvArgs)], c_int
        lib.fb_conv_bwd_stat_supported.argtypes, lib.fb_conv_bwd_stat_supported.restype = [C.POINTER(ConvArgs)], c_int
        lib.fb_ws_bn_amax_floats.argtypes, lib.fb_ws_bn_amax_floats.restype = [c_i64, c_int, c_i64], c_i64
        lib.fb_bn_apply_can_pool.argtypes, lib.fb_bn_apply_can_pool.restype = [c_int, c_int, c_i64, c_int], c_int
        lib.fb_bn_bwd_fused_supported.argtypes, lib.fb_bn_bwd_fused_supported.restype = [c_i64, c_int, c_i64, c_int], c_int
        lib.fb_ws_bn_bwd_fused_floats.argtypes, lib.fb_ws_bn_bwd_fused_floats.restype = [c_i64, c_int, c_i64, c_int], c_i64
        lib.fb_ws_bn_bwd_fused_ints.argtypes, lib.fb_ws_bn_bwd_fused_ints.restype = [c_i64], c_i64
        lib.fb_wgrad_bn_fused_supported.argtypes, lib.fb_wgrad_bn_fused_supported.restype = [C.POINTER(WgradArgs)], c_int
        lib.fb_wgrad_chain_supported.argtypes, lib.fb_wgrad_chain_supported.restype = [C.POINTER(WgradArgs)], c_int
        lib.fb_cmd_fn_id.argtypes, lib.fb_cmd_fn_id.restype = [C.c_char_p], c_int
        lib.fb_cmd_fn_nargs.argtypes, lib.fb_cmd_fn_nargs.restype = [c_int], c_int
        lib.fb_event_new.argtypes, lib.fb_event_new.restype = [], c_int
        lib.fb_event_count.argtypes, lib.fb_event_count.restype = [], c_int
        lib.fb_event_record.argtypes, lib.fb_event_record.restype = [c_int, c_void_p], c_int
        lib.fb_event_wait.argtypes, lib.fb_event_wait.restype = [c_int, c_void_p], c_int
        lib.fb_cmdlist_create.argtypes, lib.fb_cmdlist_create.restype = [], c_void_p
        lib.fb_cmdlist_destroy.argtypes, lib.fb_cmdlist_destroy.restype = [c_void_p], None
        lib.fb_cmdlist_size.argtypes, lib.fb_cmdlist_size.restype = [c_void_p], c_i64
        lib.fb_cmdlist_add_call.argtypes, lib.fb_cmdlist_add_call.restype = [c_void_p, c_int, C.POINTER(C.c_uint64), c_int, c_int, c_void_p, c_int], c_int
        lib.fb_cmdlist_add_event.argtypes, lib.fb_cmdlist_add_event.restype = [c_void_p, c_int, c_int, c_int], c_int
        lib.fb_cmdlist_replay.argtypes, lib.fb_cmdlist_replay.restype = [c_void_p, C.POINTER(c_void_p), c_int], c_int
        _lib = lib
    return _lib


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke an entry point on torch's current stream; raises EngineError on a non-zero status.  While a ``Recorder`` is active the
    call is also appended to its command list (it still executes: the recording pass is an ordinary pass)."""
    lib = load()
    st = _stream()
    status = getattr(lib, name)(*args, st)
    if status != 0:
        raise EngineError(f"{name} failed ({status}): {lib.fb_last_error_string().decode()} [args: {args}]")
    if _recorder is not None:
        _recorder.add_call(name, args, st)


# ---------------------------------------------------------------------------------------------------------------------
# native launch executor (csrc/cmdlist.cpp): record a static sequence of calls once, replay it with one host call
# ---------------------------------------------------------------------------------------------------------------------
_recorder = None


def recording():
    return _recorder is not None


def current_recorder():
    return _recorder


def _check(status, what):
    if status != 0:
        raise EngineError(f"{what} failed ({status}): {load().fb_last_error_string().decode()}")


def event_new():
    ev = load().fb_event_new()
    if ev < 0:
        raise EngineError(load().fb_last_error_string().decode())
    return ev


def event_count():
    return int(load().fb_event_count())


def event_record(ev, stream=None):
    """Record library event ``ev`` on ``stream`` (a torch stream; default: the current one)."""
    st = _stream() if stream is None else stream.cuda_stream
    _check(load().fb_event_record(ev, st), "fb_event_record")
    if _recorder is not None:
        _recorder.add_event(1, ev, st)


def event_wait(ev, stream=None):
    """Make ``stream`` (default: the current one) wait for the last record of library event ``ev``."""
    st = _stream() if stream is None else stream.cuda_stream
    _check(load().fb_event_wait(ev, st), "fb_event_wait")
    if _recorder is not None:
        _recorder.add_event(2, ev, st)


def _pack_words(name, args):
    """One 64-bit word per argument (the convention of fb_cmdlist_add_call) + the argument struct to copy, if the call has one."""
    sig = _SIGS[name][:-1]
    if len(args) != len(sig):
        raise EngineError(f"{name}: {len(args)} arguments for a signature of {len(sig)}")
    words = (C.c_uint64 * len(sig))()
    blob = None
    for i, (t, a) in enumerate(zip(sig, args)):
        if t is c_float:
            words[i] = struct.unpack("<I", struct.pack("<f", float(a)))[0]
        elif t is c_double:
            words[i] = struct.unpack("<Q", struct.pack("<d", float(a)))[0]
        elif t is c_void_p or t is c_int or t is c_i64:
            words[i] = (0 if a is None else int(a)) & 0xFFFFFFFFFFFFFFFF
        elif i == 0 and hasattr(a, "_obj"):              # C.byref(ConvArgs / WgradArgs): the executor keeps its own copy
            blob = a._obj
        else:
            raise EngineError(f"{name}: argument {i} of type {t} cannot be recorded")
    return words, blob


class CommandList:
    """A recorded sequence of library calls and event operations (``Recorder``); ``replay`` issues it natively."""

    def __init__(self, handle, n_streams, keep, events=()):
        self.handle, self.n_streams, self.keep, self.events = handle, n_streams, keep, list(events)     # events: the ids recorded INTO this list
        self.used_streams = frozenset()

    def __len__(self):
        return int(load().fb_cmdlist_size(self.handle))

    def replay(self, streams):
        """``streams``: torch streams in the order the recorder was given them (None entries are allowed where the recording did not use them)."""
        if any(s is None and i in self.used_streams for i, s in enumerate(streams)):
            raise EngineError("CommandList.replay: the list has commands for a stream index that is None now")
        arr = (c_void_p * self.n_streams)(*[None if s is None else s.cuda_stream for s in streams])
        _check(load().fb_cmdlist_replay(self.handle, arr, self.n_streams), "fb_cmdlist_replay")

    def __del__(self):
        try:
            if self.handle and _lib is not None:
                _lib.fb_cmdlist_destroy(self.handle)
        except Exception:
            pass
        self.handle = None


class Recorder:
    """``with Recorder(streams) as rec: ...`` -- every ``call`` / ``event_record`` / ``event_wait`` issued inside (they execute as usual) is
    appended to a command list; ``rec.finish()`` returns it.  ``streams``: the torch streams the region may use; a call on any other
    stream raises.  ``keep(obj)``: objects (buffers) that must stay alive as long as the list does."""

    def __init__(self, streams):
        self.streams = list(streams)
        self.handles = [None if s is None else s.cuda_stream for s in self.streams]
        self.handle = load().fb_cmdlist_create()
        self.kept = []
        self.events, self.used = [], set()
        self._fn_ids = {}

    def __enter__(self):
        global _recorder
        if _recorder is not None:
            raise EngineError("Recorder: recordings do not nest")
        _recorder = self
        return self

    def __exit__(self, *exc):
        global _recorder
        _recorder = None
        if exc[0] is not None and self.handle:
            load().fb_cmdlist_destroy(self.handle)
            self.handle = None
        return False

    def _stream_index(self, st):
        try:
            return self.handles.index(st)
        except ValueError:
            raise EngineError("Recorder: a call was issued on a stream the recording does not know") from None

    def keep(self, obj):
        self.kept.append(obj)

    def add_call(self, name, args, st):
        fn = self._fn_ids.get(name)
        if fn is None:
            fn = self._fn_ids[name] = load().fb_cmd_fn_id(name.encode())
            if fn < 0:
                raise EngineError(f"Recorder: {name} cannot be part of a command list")
        words, blob = _pack_words(name, args)
        self.used.add(self._stream_index(st))
        _check(load().fb_cmdlist_add_call(self.handle, fn, words, len(words), self._stream_index(st), C.addressof(blob) if blob is not None else None,
                                          C.sizeof(blob) if blob is not None else 0), "fb_cmdlist_add_call")

    def add_event(self, kind, ev, st):
        self.used.add(self._stream_index(st))
        _check(load().fb_cmdlist_add_event(self.handle, kind, ev, self._stream_index(st)), "fb_cmdlist_add_event")

    def finish(self):
        out = CommandList(self.handle, len(self.streams), self.kept, self.events)
        out.used_streams = frozenset(self.used)
        self.handle = None
        return out


def dtype_code(dtype):
    if dtype == torch.float32:
        return FB_F32
    if dtype == torch.bfloat16:
        return FB_BF16
    raise EngineError(f"unsupported compute dtype {dtype}")


# ---------------------------------------------------------------------------------------------------------------------
# thin tensor-level wrappers (shapes are read from the tensors; NHWC activations)
# ---------------------------------------------------------------------------------------------------------------------
def conv2d(src, wgt, dst, R, S, stride, pad, mode, addend=None, addend_mode=0, stat_partial=None, imgs_per_wset=0, wset_stride=0,
           addend_mask=None, bst_x=None, bst_mask=None, amax_src=None, amax_wgt=None, amax_imgs=0):
    n, hs, ws, cs = src.shape
    _, hd, wd, cd = dst.shape
    a = ConvArgs(_ptr(src), _ptr(wgt), _ptr(dst), _ptr(addend), _ptr(stat_partial), n, hs, ws, cs, hd, wd, cd, R, S, stride, pad, mode,
                 imgs_per_wset, wset_stride, addend_mode, dtype_code(src.dtype), _ptr(addend_mask), _ptr(bst_x), _ptr(bst_mask), _ptr(amax_src), _ptr(amax_wgt), amax_imgs)
    if bst_x is not None and not load().fb_conv_bwd_stat_supported(C.byref(a)):
        raise EngineError("fb_conv2d: fused BatchNorm-backward reduction not supported for these arguments")
    call("fb_conv2d", C.byref(a))


def conv2d_wgrad(x, dy, dw_partial, R, S, stride, pad, imgs_per_group, split_k, group_stride=0, amax_x=None, amax_dy=None):
    n, hs, ws, cs = x.shape
    _, hd, wd, cd = dy.shape
    a = WgradArgs(_ptr(x), _ptr(dy), _ptr(dw_partial), n, hs, ws, cs, hd, wd, cd, R, S, stride, pad, imgs_per_group, split_k,
                  dtype_code(x.dtype), group_stride, _ptr(amax_x), _ptr(amax_dy))
    call("fb_conv2d_wgrad", C.byref(a))


def wgrad_reduce(dw_partial, out, out_group_stride, n_groups, split_k, Cd, taps, Cs_pad, Cs_real):
    call("fb_wgrad_reduce", _ptr(dw_partial), _ptr(out), out_group_stride, n_groups, split_k, Cd, taps, Cs_pad, Cs_real)


def weight_prep(master, wset_stride_in, wset_stride_out, n_wsets, Cout, taps, Cin_real, Cin_pad, w_fwd, w_dgrad, dtype, amax=None):
    call("fb_weight_prep", _ptr(master), wset_stride_in, wset_stride_out, n_wsets, Cout, taps, Cin_real, Cin_pad, _ptr(w_fwd),
         _ptr(w_dgrad), dtype_code(dtype), _ptr(amax))
