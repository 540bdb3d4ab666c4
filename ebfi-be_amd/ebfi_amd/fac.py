"""Filter-Adaptive Convolution -- host side.

Mirrors the reference's ``models/FAC/kernelconv2d/KernelConv2D.py``: ``KernelConv2DFunction``
(:12-58, ``apply(input_pad, kernel, kernel_size)`` -> output, backward returns
``(grad_input, grad_kernel, None)``) and the ``KernelConv2D`` module (:77-87, replicate-pad by
K//2 then the op).  Same argument checks, same error on CPU tensors (NotImplementedError,
:38-39,55-56); the compute is ``ebfi_fac_forward`` / ``ebfi_fac_backward`` of libebfi_hip.so.
"""
import torch
from torch import nn
from torch.autograd import Function

from . import _native as N


def fac_forward(input_pad, kernel, kernel_size, out=None):
    """Raw op on already padded input.  Any strides are accepted by the native side."""
    N.require_gpu(input_pad, kernel)
    B, C = input_pad.size(0), input_pad.size(1)
    Ho, Wo = kernel.size(2), kernel.size(3)
    if out is None:
        out = torch.empty((B, C, Ho, Wo), dtype=input_pad.dtype, device=input_pad.device)
    if out.numel() == 0:
        return out
    with torch.cuda.device_of(input_pad):
        rc = N.lib().ebfi_fac_forward(
            N.ptr(input_pad), N.i64x4(input_pad.shape), N.i64x4(input_pad.stride()),
            N.ptr(kernel), N.i64x4(kernel.shape), N.i64x4(kernel.stride()), int(kernel_size),
            N.ptr(out), N.i64x4(out.shape), N.i64x4(out.stride()),
            N.dtype_code(input_pad), N.stream_ptr(input_pad.device))
    N.check(rc, "ebfi_fac_forward")
    return out


def fac_backward(input_pad, kernel, kernel_size, grad_output, need_input=True, need_kernel=True, kernel_leaky_slope=None):
    """`kernel_leaky_slope`: the filters came out of a LeakyReLU(slope) layer and grad_kernel is wanted as the gradient of
    that layer's pre-activation (None = plain grad_kernel)."""
    N.require_gpu(input_pad, kernel, grad_output)
    gin = torch.empty_like(input_pad, memory_format=torch.contiguous_format) if need_input else None
    gk = torch.empty_like(kernel, memory_format=torch.contiguous_format) if need_kernel else None
    unit = N.i64x4((0, 0, 0, 1))
    if kernel.numel() == 0:
        return gin, gk
    with torch.cuda.device_of(input_pad):
        rc = N.lib().ebfi_fac_backward_ex(
            N.ptr(input_pad), N.i64x4(input_pad.shape), N.i64x4(input_pad.stride()),
            N.ptr(kernel), N.i64x4(kernel.shape), N.i64x4(kernel.stride()), int(kernel_size),
            N.ptr(grad_output), N.i64x4(grad_output.stride()),
            N.ptr(gin), N.i64x4(gin.stride()) if gin is not None else unit,
            N.ptr(gk), N.i64x4(gk.stride()) if gk is not None else unit,
            1.0 if kernel_leaky_slope is None else float(kernel_leaky_slope),
            N.dtype_code(input_pad), N.stream_ptr(input_pad.device))
    N.check(rc, "ebfi_fac_backward")
    return gin, gk


def fac_rows_fold_weight(w, ksize=5, tile=32):
    """[C*K*K, Cin, 3, 3] -> [C*tile, Cin, 3, 3]: one FAC channel per `tile`-row matrix tile (K*K filter rows + zero rows): the
    row layout `ebfi_kernelconv_fac_fused_x3` expects.  A 0/1 linear map (usable as a weight-bank fold)."""
    kk = ksize * ksize
    c = w.shape[0] // kk
    return torch.nn.functional.pad(w.reshape(c, kk, *w.shape[1:]), (0, 0) * (w.dim() - 1) + (0, tile - kk)).reshape(c * tile, *w.shape[1:])


def fac_rows_fold_bias(b, ksize=5, tile=32):
    kk = ksize * ksize
    return torch.nn.functional.pad(b.reshape(-1, kk), (0, tile - kk)).reshape(-1)


def kernelconv_fac_fused(cat, feat, site, kernel_size, slope):
    """filters = LeakyReLU(conv3x3(cat)) applied to the replicate-padded `feat` as ONE kernel (no filter tensor; inference
    only -- no autograd node).  `site`: the weight bank's "facrows" images of the KernelConv layer."""
    N.require_gpu(cat, feat)
    cat, feat = cat.contiguous(), feat.contiguous()
    B, Cin, H, W = (int(v) for v in cat.shape)
    C = int(feat.shape[1])
    if site.M != C * 32 or site.K != Cin or tuple(feat.shape) != (B, C, H, W):
        raise RuntimeError("fused KernelConv -> FAC: packed weight [%d,%d] does not match input %s / feature %s"
                           % (site.M, site.K, tuple(cat.shape), tuple(feat.shape)))
    out = torch.empty_like(feat)
    with torch.cuda.device_of(cat):
        rc = N.lib().ebfi_kernelconv_fac_fused_x3(N.ptr(cat), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(feat),
                                                  N.ptr(out), B, Cin, H, W, C, int(kernel_size), float(slope),
                                                  N.stream_ptr(cat.device))
    N.check(rc, "ebfi_kernelconv_fac_fused_x3")
    return out


class KernelConv2DFunction(Function):
    @staticmethod
    def forward(ctx, input, kernel, kernel_size, kernel_leaky_slope=None):
        """`kernel_leaky_slope` (extension of the reference signature, default None = reference behaviour): see fac_backward;
        only for a `kernel` whose ONLY consumer is this op and whose producer runs with conv.conv_bias_act(...,
        grad_is_preact=True)."""
        ctx.kernel_size = kernel_size
        ctx.kernel_leaky_slope = kernel_leaky_slope
        assert input.is_contiguous()
        assert kernel.is_contiguous()
        assert ctx.kernel_size == int((kernel.size(1) / input.size(1)) ** 0.5)
        assert input.size(2) - kernel_size == kernel.size(2) - 1
        assert input.size(3) - kernel_size == kernel.size(3) - 1
        if not input.is_cuda:
            raise NotImplementedError()  # as the reference: no CPU version
        ctx.save_for_backward(input, kernel)
        return fac_forward(input, kernel, kernel_size)

    @staticmethod
    def backward(ctx, grad_output):
        input, kernel = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        if not grad_output.is_cuda:
            raise NotImplementedError()
        gin, gk = fac_backward(input, kernel, ctx.kernel_size, grad_output,
                               ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.kernel_leaky_slope)
        return gin, gk, None, None


class KernelConv2D(nn.Module):
    def __init__(self, kernel_size):
        super().__init__()
        assert kernel_size % 2 == 1
        self.kernel_size = kernel_size
        r = (kernel_size - 1) // 2
        self.pad = nn.ReplicationPad2d([r, r, r, r])

    def forward(self, input, kernel, kernel_leaky_slope=None):
        return KernelConv2DFunction.apply(self.pad(input), kernel, self.kernel_size, kernel_leaky_slope)
