"""ctypes binding of libebfi_hip.so (the C ABI declared in include/ebfi_hip.h).

This is the only place the host code touches the native library.  Loading is lazy and LOUD: if the
shared object is missing, cannot be loaded or lacks a declared symbol, the first use of any op
raises ``EbfiNativeError`` -- there is no CPU or PyTorch fallback anywhere in this package.
"""
import ctypes
import os
import re
import subprocess
import threading

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # .../ebfi-be_amd
REPO_ROOT = os.path.dirname(PKG_ROOT)
# Development switches (EBFI_NO_*, EBFI_WGRAD_TR, ... -- A/B measurements and tests): honoured only in a process started with
# EBFI_DEV=1, like the library's own (csrc/common.hpp dev_getenv), so that a stray variable cannot change which kernels a
# production run, one of its ranks or one of its graph captures takes.
_DEV = os.environ.get("EBFI_DEV") == "1"


def dev_env(name, default=None):
    return os.environ.get(name, default) if _DEV else default


# EBFI_LIB_PATH (a second build of the library for same-box A/B runs, tools/ab_bench.sh) is a development switch like the
# others: ignored unless EBFI_DEV=1.  The ABI version of whatever is loaded must match in every case -- an older build whose
# entry points changed signature would be called with shifted arguments (round-3 advisory).
LIB_PATH = dev_env("EBFI_LIB_PATH") or os.path.join(PKG_ROOT, "lib", "libebfi_hip.so")
HEADER = os.path.join(REPO_ROOT, "include", "ebfi_hip.h")
BUILD_SCRIPT = os.path.join(PKG_ROOT, "csrc", "build.sh")

EBFI_F32, EBFI_F32_BF16MMA, EBFI_F32_BF16X3MMA = 0, 2, 3
EBFI_ERR_UNSUPPORTED = -3   # include/ebfi_hip.h ebfi_status
ABI_VERSION = 14         # include/ebfi_hip.h EBFI_ABI_VERSION


class EbfiNativeError(RuntimeError):
    pass


_lock = threading.Lock()
_lib = None

_c = ctypes
_vp, _i, _i64, _sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_size_t
_p64 = _c.POINTER(_c.c_int64)

# name -> (restype, argtypes); must list every function the header declares (tests check that)
SIGNATURES = {
    "ebfi_abi_version": (_i, []),
    "ebfi_last_error": (_c.c_char_p, []),
    "ebfi_fac_forward": (_i, [_vp, _p64, _p64, _vp, _p64, _p64, _i, _vp, _p64, _p64, _i, _vp]),
    "ebfi_fac_backward": (_i, [_vp, _p64, _p64, _vp, _p64, _p64, _i, _vp, _p64, _vp, _p64, _vp, _p64, _i, _vp]),
    "ebfi_fac_backward_ex": (_i, [_vp, _p64, _p64, _vp, _p64, _p64, _i, _vp, _p64, _vp, _p64, _vp, _p64, _c.c_float, _i, _vp]),
    "ebfi_dcn_forward": (_i, [_vp] * 6 + [_i] * 14 + [_i, _vp]),
    "ebfi_dcn_backward_workspace": (_sz, [_i] * 14 + [_i]),
    "ebfi_dcn_backward": (_i, [_vp] * 11 + [_i] * 14 + [_vp, _sz, _i, _vp]),
    "ebfi_conv2d_forward": (_i, [_vp] * 4 + [_i] * 8 + [_i, _c.c_float, _i, _vp]),
    "ebfi_conv2d_backward_data": (_i, [_vp] * 4 + [_i] * 8 + [_i, _c.c_float, _i, _vp]),
    "ebfi_conv2d_backward_weight_workspace": (_sz, [_i] * 8 + [_i]),
    "ebfi_conv2d_backward_weight": (_i, [_vp] * 5 + [_i] * 8 + [_i, _c.c_float, _vp, _sz, _i, _vp]),
    "ebfi_conv2d_backward_weight_ex": (_i, [_vp] * 6 + [_i] * 8 + [_i, _c.c_float, _vp, _sz, _i, _vp]),
    "ebfi_conv2d_bf16_workspace": (_sz, [_i, _i, _i]),
    "ebfi_conv2d_forward_bf16mma": (_i, [_vp] * 4 + [_i] * 8 + [_i, _c.c_float, _vp, _sz, _vp]),
    "ebfi_conv2d_backward_data_bf16mma": (_i, [_vp] * 4 + [_i] * 8 + [_i, _c.c_float, _vp, _sz, _vp]),
    "ebfi_conv2d_forward_bf16x3": (_i, [_vp] * 4 + [_i] * 8 + [_i, _c.c_float, _vp, _sz, _vp]),
    "ebfi_conv2d_backward_data_bf16x3": (_i, [_vp] * 4 + [_i] * 8 + [_i, _c.c_float, _vp, _sz, _vp]),
    "ebfi_conv2d_packed_bytes": (_sz, [_i, _i, _i, _i]),
    "ebfi_conv2d_pack_bf16x3": (_i, [_vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ebfi_pack_table_bf16": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "ebfi_f16_scales_finish": (_i, [_vp, _i, _vp, _vp]),
    "ebfi_pack_table_f16": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "ebfi_conv2d_packed_f16": (_i, [_vp, _vp, _sz, _vp, _vp] + [_i] * 8 + [_i, _c.c_float, _vp, _vp, _i, _c.c_float, _vp, _vp, _vp]),
    "ebfi_conv2d_backward_weight_f16g": (_i, [_vp] * 6 + [_i] * 8 + [_i, _c.c_float, _vp, _vp, _vp, _sz, _vp]),
    "ebfi_conv2d_backward_weight_f16g_ex": (_i, [_vp] * 6 + [_i] * 9 + [_i, _c.c_float, _vp, _vp, _vp, _sz, _vp]),
    "ebfi_to_c16": (_i, [_vp, _vp, _c.c_float, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ebfi_to_c16_cat2": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    "ebfi_conv2d_packed_x3_c16": (_i, [_vp, _vp, _sz, _vp, _vp] + [_i] * 8 + [_i, _c.c_float, _vp, _vp, _i, _c.c_float, _vp, _vp, _i, _vp]),
    "ebfi_conv2d_packed_f16_c16": (_i, [_vp, _i, _vp, _sz, _vp, _vp] + [_i] * 8 + [_i, _c.c_float, _vp, _vp, _i, _c.c_float,
                                        _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "ebfi_conv2d_backward_weight_f16c": (_i, [_vp, _vp, _i, _vp, _vp] + [_i] * 6 + [_vp, _vp, _vp, _sz, _vp]),
    "ebfi_conv2d_backward_weight_f16c_batch_workspace": (_sz, [_i, _vp, _vp]),
    "ebfi_conv2d_backward_weight_f16c_batch": (_i, [_i] + [_vp] * 9 + [_i, _i, _i, _vp, _sz, _vp]),
    "ebfi_fac_forward_p16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ebfi_fac_backward_p16": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_float, _i, _i, _i, _i, _i, _vp]),
    "ebfi_scale_residual_cat_forward_c16": (_i, [_vp] * 8 + [_i, _i, _i, _i, _i64, _vp]),
    "ebfi_scale_residual_cat_backward_slices": (_i, []),
    "ebfi_scale_residual_cat_backward_c16": (_i, [_vp] * 10 + [_i, _i, _i, _i, _i64, _c.c_float, _vp]),
    "ebfi_scale_residual_cat_backward_c16a": (_i, [_vp] * 10 + [_i, _i, _i, _i, _c.c_float, _vp]),
    "ebfi_conv2d_packed_x3_rc": (_i, [_vp, _vp, _sz, _vp, _vp] + [_i] * 6 + [_c.c_float, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "ebfi_conv2d_thin_forward": (_i, [_vp, _vp, _vp, _vp] + [_i] * 8 + [_i, _c.c_float, _vp]),
    "ebfi_kernelconv_fac_fused_x3": (_i, [_vp, _vp, _sz, _vp, _vp, _vp] + [_i] * 6 + [_c.c_float, _vp]),
    "ebfi_kernelconv_fac_fused_f16": (_i, [_vp, _i, _vp, _sz, _vp, _vp, _vp] + [_i] * 6 + [_c.c_float, _vp, _vp, _vp]),
    "ebfi_conv2d_packed_x3": (_i, [_vp, _vp, _sz, _vp, _vp] + [_i] * 8 + [_i, _c.c_float, _vp, _vp, _i, _c.c_float, _vp]),
    "ebfi_conv2d_backward_weight_x3g": (_i, [_vp] * 4 + [_i] * 8 + [_vp, _sz, _vp]),
    "ebfi_scale_residual_cat_forward_ex": (_i, [_vp] * 6 + [_i, _i, _i64, _i64, _vp]),
    "ebfi_scale_residual_cat_backward_ex": (_i, [_vp] * 10 + [_i, _i, _i64, _i64, _i64, _i, _c.c_float, _vp]),
    "ebfi_scalar_conv_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _c.c_float, _vp]),
    "ebfi_scalar_conv_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _c.c_float, _vp]),
    "ebfi_events_workspace": (_sz, [_i]),
    "ebfi_events_to_stack": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ebfi_frame2lap": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ebfi_frame2dcp": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ebfi_scale_residual_cat_forward": (_i, [_vp] * 6 + [_i, _i, _i64, _vp]),
    "ebfi_scale_residual_cat_backward": (_i, [_vp] * 10 + [_i, _i, _i64, _vp]),
    "ebfi_prodmean_forward": (_i, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "ebfi_prodmean_backward": (_i, [_vp] * 5 + [_i64, _i64, _vp]),
    "ebfi_gather_sum": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "ebfi_se_gate_workspace": (_c.c_size_t, [_i, _i, _i64]),
    "ebfi_se_gate_forward": (_i, [_vp] * 8 + [_i, _i, _i64, _i, _c.c_float, _vp]),
    "ebfi_se_gate_backward": (_i, [_vp] * 11 + [_i, _i, _i64, _i, _c.c_float, _vp]),
    "ebfi_conv2d_packed_x3_shuffled": (_i, [_vp, _vp, _c.c_size_t, _vp, _vp] + [_i] * 6 + [_c.c_float, _i, _vp]),
    "ebfi_conv2d_packed_f16_shuffled": (_i, [_vp, _i, _vp, _c.c_size_t, _vp, _vp] + [_i] * 6 + [_c.c_float, _vp, _i, _c.c_float, _vp, _vp, _i, _vp]),
    "ebfi_se_gate_forward_ps": (_i, [_vp] * 7 + [_i, _i, _i, _i, _i, _c.c_float, _vp]),
    "ebfi_se_gate_backward_ps": (_i, [_vp] * 10 + [_i, _i, _i, _i, _i, _c.c_float, _vp]),
    "ebfi_groupnorm_workspace": (_sz, [_i, _i]),
    "ebfi_groupnorm_forward": (_i, [_vp] * 6 + [_i, _i, _i64, _i, _c.c_float, _vp, _sz, _vp]),
    "ebfi_groupnorm_backward": (_i, [_vp] * 8 + [_i, _i, _i64, _i, _vp, _sz, _vp]),
    "ebfi_census_partials": (_i64, [_i, _i, _i]),
    "ebfi_census_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ebfi_census_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ebfi_gauss5_forward": (_i, [_vp, _vp, _i64, _i, _i, _c.c_float, _vp]),
    "ebfi_gauss5_backward": (_i, [_vp, _vp, _i64, _i, _i, _c.c_float, _vp]),
    "ebfi_ed_head_workspace": (_sz, [_i, _i, _i64]),
    "ebfi_ed_head_forward": (_i, [_vp] * 7 + [_i, _i, _i64, _i, _c.c_float, _vp, _sz, _vp]),
    "ebfi_ed_head_backward": (_i, [_vp] * 11 + [_i, _i, _i64, _i, _vp, _sz, _vp]),
    "ebfi_conv2d_backward_data_s2_bf16x3": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ebfi_adam_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i64] + [_c.c_double] * 4 + [_vp]),
    "ebfi_adam_step_guarded": (_i, [_vp, _vp, _vp, _vp, _vp, _i64] + [_c.c_double] * 4 + [_vp, _vp, _vp]),
    "ebfi_pad2d_backward": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "ebfi_grad_gather": (_i, [_vp, _vp, _i, _vp, _i64, _i, _vp, _vp]),
    "ebfi_laploss_workspace_floats": (_i64, [_i64, _i, _i, _i]),
    "ebfi_laploss_partials": (_i64, [_i64, _i, _i, _i]),
    "ebfi_laploss_forward": (_i, [_vp, _vp, _vp, _c.c_float, _c.c_float, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "ebfi_laploss_backward": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "ebfi_prof_enable": (None, [_i]),
    "ebfi_prof_set_capacity": (_i, [_i]),
    "ebfi_prof_reset": (None, []),
    "ebfi_prof_collect": (_i, [_c.POINTER(_i)]),
    "ebfi_prof_num_kernels": (_i, []),
    "ebfi_prof_get": (_i, [_i, _c.POINTER(_c.c_char_p), _c.POINTER(_i64), _c.POINTER(_c.c_double)]),
    "ebfi_prof_get_work": (_i, [_i, _c.POINTER(_c.c_double), _c.POINTER(_c.c_double)]),
}


def declared_symbols():
    """Function names declared in include/ebfi_hip.h (used by the export test)."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ebfi_[a-z0-9_]+)\s*\(", text)))


def build(verbose=False):
    """Compile libebfi_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["bash", BUILD_SCRIPT], stdout=out)
    return LIB_PATH


def _counted(fn, name, n):
    def call(*a):
        if len(a) != n:
            raise TypeError("%s takes %d arguments (include/ebfi_hip.h), %d given" % (name, n, len(a)))
        return fn(*a)
    call.__name__ = name
    return call


def lib():
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise EbfiNativeError(
                "native library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or ebfi-be_amd/csrc/build.sh).  There is no CPU fallback." % LIB_PATH)
        try:
            h = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise EbfiNativeError("cannot load %s: %s" % (LIB_PATH, e)) from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(h, name)
            except AttributeError as e:
                raise EbfiNativeError("%s does not export %s" % (LIB_PATH, name)) from e
            fn.restype = res
            fn.argtypes = args
            # ctypes accepts EXTRA positional arguments silently (they are passed after the declared ones and ignored by the
            # callee): a stray argument in front of the stream pointer put every image-reading data gradient on the NULL
            # stream for two hours of round 4.  Every entry point is therefore called through a wrapper that checks the count.
            setattr(h, name, _counted(fn, name, len(args)))
        if h.ebfi_abi_version() != ABI_VERSION:
            raise EbfiNativeError("ABI version mismatch: library %d, binding %d" % (h.ebfi_abi_version(), ABI_VERSION))
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().ebfi_last_error()
        raise EbfiNativeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def i64x4(vals):
    return (_c.c_int64 * 4)(*[int(v) for v in vals])


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return _vp(t.data_ptr()) if t is not None else _vp(0)


def stream_ptr(device=None):
    import torch
    return _vp(torch.cuda.current_stream(device).cuda_stream)


def dtype_code(t):
    import torch
    if t.dtype == torch.float32:
        return EBFI_F32
    raise EbfiNativeError("unsupported dtype %s (the ops take float32 tensors, like the reference's)" % t.dtype)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NotImplementedError(
                "ebfi_amd ops run on an MI355X through libebfi_hip.so only; got a %s tensor "
                "(the reference's FAC has no CPU path either: KernelConv2D.py:38-39)" % t.device)


# ------------------------------------------------------------------ profiler helpers
def prof_enable(on=True):
    lib().ebfi_prof_enable(1 if on else 0)


def prof_reset():
    lib().ebfi_prof_reset()


def prof_fold():
    """Fold the pending event pairs into the per-kernel totals (call after torch.cuda.synchronize(), e.g. once per
    step, so that the bounded pending list never overflows).  Returns the number of launches dropped so far."""
    dropped = _i(0)
    lib().ebfi_prof_collect(ctypes.byref(dropped))
    return dropped.value


def prof_collect(strict=True):
    """{label: (launches, total_ms, algorithmic flops, algorithmic bytes)}; call after torch.cuda.synchronize().
    Raises when launches were dropped (pending list full): totals would be silently truncated otherwise."""
    h = lib()
    dropped = _i(0)
    h.ebfi_prof_collect(ctypes.byref(dropped))
    if strict and dropped.value:
        raise EbfiNativeError("profiler dropped %d launches (pending capacity exceeded): fold once per step with "
                              "prof_fold() or raise ebfi_prof_set_capacity" % dropped.value)
    out = {}
    for k in range(h.ebfi_prof_num_kernels()):
        name, n, ms = _c.c_char_p(), _i64(0), _c.c_double(0)
        fl, by = _c.c_double(0), _c.c_double(0)
        h.ebfi_prof_get(k, ctypes.byref(name), ctypes.byref(n), ctypes.byref(ms))
        h.ebfi_prof_get_work(k, ctypes.byref(fl), ctypes.byref(by))
        out[name.value.decode()] = (n.value, ms.value, fl.value, by.value)
    out["__dropped__"] = (dropped.value, 0.0, 0.0, 0.0)
    return out
