"""ORACLE (test infrastructure only).

ctypes front-end of the C restatements in this directory (fac_ref*.c, dcn_ref*.c) plus the
autograd wrappers the CPU baseline needs.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product package
(``ebfi-be_amd/``) never does.

Reference semantics restated here:
  * FAC op + module: models/FAC/kernelconv2d/KernelConv2D.py:12-58 (Function), :77-87 (module:
    ReplicationPad2d(K//2) then the op).
  * DCNv2 op: models/DCNv2/dcn_v2.py:17-67 (Function), src/cuda/dcn_v2_cuda.cu:20-216.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile the C restatements (gcc, a second or two)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []),
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _sfx(t):
    if t.dtype == torch.float32:
        return "_f32"
    if t.dtype == torch.float64:
        return "_f64"
    raise TypeError("oracle supports float32/float64 only, got %s" % t.dtype)


def _p(t):
    assert t.device.type == "cpu"
    return ctypes.c_void_p(t.data_ptr())


def _i64x4(vals):
    return (ctypes.c_int64 * 4)(*[int(v) for v in vals])


# ----------------------------------------------------------------------------- FAC
def fac_forward(inp, kern, K):
    """inp [B,C,Ho+K-1,Wo+K-1] (already padded), kern [B,C*K*K,Ho,Wo] -> [B,C,Ho,Wo]."""
    B, C, Hi, Wi = inp.shape
    Ho, Wo = kern.shape[2], kern.shape[3]
    assert kern.shape[1] == C * K * K and Hi - K == Ho - 1 and Wi - K == Wo - 1
    out = torch.zeros(B, C, Ho, Wo, dtype=inp.dtype)
    fn = getattr(lib(), "fac_ref_forward" + _sfx(inp))
    fn(_p(inp), _i64x4(inp.stride()), _p(kern), _i64x4(kern.stride()), _p(out),
       _i64x4(out.stride()), ctypes.c_int64(B), ctypes.c_int64(C), ctypes.c_int64(Ho),
       ctypes.c_int64(Wo), ctypes.c_int(K))
    return out


def fac_backward(inp, kern, K, gout):
    B, C, Hi, Wi = inp.shape
    Ho, Wo = kern.shape[2], kern.shape[3]
    gin = torch.zeros(inp.shape, dtype=inp.dtype)
    gk = torch.zeros(kern.shape, dtype=kern.dtype)
    fn = getattr(lib(), "fac_ref_backward" + _sfx(inp))
    fn(_p(inp), _i64x4(inp.stride()), _p(kern), _i64x4(kern.stride()), _p(gout),
       _i64x4(gout.stride()), _p(gin), _i64x4(gin.stride()), _p(gk), _i64x4(gk.stride()),
       ctypes.c_int64(B), ctypes.c_int64(C), ctypes.c_int64(Ho), ctypes.c_int64(Wo),
       ctypes.c_int(K))
    return gin, gk


class FacRefFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, kern, K):
        inp, kern = inp.contiguous(), kern.contiguous()
        ctx.K = K
        ctx.save_for_backward(inp, kern)
        return fac_forward(inp, kern, K)

    @staticmethod
    def backward(ctx, gout):
        inp, kern = ctx.saved_tensors
        gin, gk = fac_backward(inp, kern, ctx.K, gout.contiguous())
        return gin, gk, None


def fac_module(x, kern, K):
    """KernelConv2D.forward: replicate-pad by K//2, then the op (KernelConv2D.py:85-87)."""
    r = (K - 1) // 2
    xp = torch.nn.functional.pad(x, (r, r, r, r), mode="replicate")
    return FacRefFunction.apply(xp, kern, K)


# ----------------------------------------------------------------------------- DCNv2
def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def dcn_out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw):
    return ((H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1,
            (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1)


def _geom_args(B, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw, dg, Co=None):
    vals = [B, C, H, W] + ([Co] if Co is not None else []) + [kh, kw, sh, sw, ph, pw, dh, dw, dg]
    return [ctypes.c_int(int(v)) for v in vals]


def dcn_im2col(x, offset, mask, ksize, stride, padding, dilation, dg):
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = map(_pair, (ksize, stride, padding, dilation))
    x, offset, mask = x.contiguous(), offset.contiguous(), mask.contiguous()
    B, C, H, W = x.shape
    Ho, Wo = dcn_out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw)
    col = torch.zeros(B, C * kh * kw, Ho * Wo, dtype=x.dtype)
    fn = getattr(lib(), "dcn_ref_im2col" + _sfx(x))
    fn(_p(x), _p(offset), _p(mask), _p(col),
       *_geom_args(B, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw, dg))
    return col


def dcn_forward(x, weight, bias, offset, mask, stride, padding, dilation, dg):
    (sh, sw), (ph, pw), (dh, dw) = map(_pair, (stride, padding, dilation))
    x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
    offset, mask = offset.contiguous(), mask.contiguous()
    B, C, H, W = x.shape
    Co, Cw, kh, kw = weight.shape
    assert Cw == C
    Ho, Wo = dcn_out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw)
    assert offset.shape == (B, dg * 2 * kh * kw, Ho, Wo), offset.shape
    assert mask.shape == (B, dg * kh * kw, Ho, Wo), mask.shape
    out = torch.zeros(B, Co, Ho, Wo, dtype=x.dtype)
    fn = getattr(lib(), "dcn_ref_forward" + _sfx(x))
    fn(_p(x), _p(weight), _p(bias), _p(offset), _p(mask), _p(out),
       *_geom_args(B, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw, dg, Co=Co))
    return out


def dcn_backward(x, weight, bias, offset, mask, gout, stride, padding, dilation, dg):
    (sh, sw), (ph, pw), (dh, dw) = map(_pair, (stride, padding, dilation))
    x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
    offset, mask, gout = offset.contiguous(), mask.contiguous(), gout.contiguous()
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    gx, gw, gb = torch.zeros_like(x), torch.zeros_like(weight), torch.zeros_like(bias)
    go, gm = torch.zeros_like(offset), torch.zeros_like(mask)
    fn = getattr(lib(), "dcn_ref_backward" + _sfx(x))
    fn(_p(x), _p(weight), _p(bias), _p(offset), _p(mask), _p(gout), _p(gx), _p(go), _p(gm),
       _p(gw), _p(gb), *_geom_args(B, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw, dg, Co=Co))
    return gx, go, gm, gw, gb


class DcnRefFunction(torch.autograd.Function):
    """Argument order of the reference op: models/DCNv2/dcn_v2.py:19-21."""

    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias, stride, padding, dilation, dg):
        ctx.cfg = (stride, padding, dilation, dg)
        ctx.save_for_backward(x, offset, mask, weight, bias)
        return dcn_forward(x, weight, bias, offset, mask, stride, padding, dilation, dg)

    @staticmethod
    def backward(ctx, gout):
        x, offset, mask, weight, bias = ctx.saved_tensors
        stride, padding, dilation, dg = ctx.cfg
        gx, go, gm, gw, gb = dcn_backward(x, weight, bias, offset, mask, gout, stride, padding,
                                          dilation, dg)
        return gx, go, gm, gw, gb, None, None, None, None


dcn_v2_conv = DcnRefFunction.apply
