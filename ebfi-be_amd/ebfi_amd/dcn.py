"""Modulated deformable convolution v2 -- host side.

Mirrors the reference's ``models/DCNv2/dcn_v2.py``: the autograd op ``_DCNv2`` / ``dcn_v2_conv``
(:17-95, argument order ``(input, offset, mask, weight, bias, stride, padding, dilation,
deformable_groups)``), and the modules ``DCNv2`` (:98-146), ``DCN`` (:149-194) and ``DCN_sep``
(:197-227) with the same parameter names (``weight``, ``bias``, ``conv_offset_mask.*``), the same
initialisation and the same argument checks as the extension wrappers
(src/cuda/dcn_v2_cuda.cu:38-62,110-111).  Compute: ``ebfi_dcn_forward`` / ``ebfi_dcn_backward``.

Not mirrored (out of scope): PS-ROI pooling and the ONNX wrapper.
"""
import logging
import math

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import _native as N

logger = logging.getLogger("base")


def _geometry(input, weight, stride, padding, dilation, dg):
    B, C, H, W = input.shape
    Co, Cw, kh, kw = weight.shape
    if C != Cw:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (C, Cw))
    (sh, sw), (ph, pw), (dh, dw) = stride, padding, dilation
    return [int(v) for v in (B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg)]


def _out_hw(geo):
    B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg = geo
    return ((H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1, (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1)


def _check_offset_mask(geo, offset, mask):
    B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg = geo
    Ho, Wo = _out_hw(geo)
    if tuple(offset.shape) != (B, dg * 2 * kh * kw, Ho, Wo):
        raise RuntimeError("offset must be %s, got %s" % ((B, dg * 2 * kh * kw, Ho, Wo), tuple(offset.shape)))
    if tuple(mask.shape) != (B, dg * kh * kw, Ho, Wo):
        raise RuntimeError("mask must be %s, got %s" % ((B, dg * kh * kw, Ho, Wo), tuple(mask.shape)))


PRODUCTS = ("fp32", "bf16x3")


def dcn_v2_forward(input, weight, bias, offset, mask, stride, padding, dilation, dg, product="fp32"):
    """`product`: matrix-core operands of the 64 x C*kh*kw x 64 product -- "fp32" (exact, the kernel the reference's
    known-answer tests pin; also the faster one) or "bf16x3" (split precision, ~1e-5).  It is an explicit
    argument of the op, never taken from ambient process state."""
    if product not in PRODUCTS:
        raise ValueError("product must be one of %s" % (PRODUCTS,))
    for name, t in (("input", input), ("weight", weight), ("bias", bias), ("offset", offset), ("mask", mask)):
        if not t.is_cuda:
            raise NotImplementedError("%s must be a GPU tensor: libebfi_hip.so has no CPU path" % name)
    geo = _geometry(input, weight, stride, padding, dilation, dg)
    _check_offset_mask(geo, offset, mask)
    # the reference reads these through raw data pointers (no contiguity check); be explicit instead
    input, weight, bias = input.contiguous(), weight.contiguous(), bias.contiguous()
    offset, mask = offset.contiguous(), mask.contiguous()
    Ho, Wo = _out_hw(geo)
    out = torch.empty((geo[0], geo[4], Ho, Wo), dtype=input.dtype, device=input.device)
    code = N.dtype_code(input)
    if code == N.EBFI_F32 and product == "bf16x3":
        code = N.EBFI_F32_BF16X3MMA
    with torch.cuda.device_of(input):
        rc = N.lib().ebfi_dcn_forward(N.ptr(input), N.ptr(weight), N.ptr(bias), N.ptr(offset), N.ptr(mask),
                                      N.ptr(out), *geo, code, N.stream_ptr(input.device))
    N.check(rc, "ebfi_dcn_forward")
    return out


def dcn_v2_backward(input, weight, bias, offset, mask, grad_output, stride, padding, dilation, dg):
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")
    for name, t in (("input", input), ("weight", weight), ("bias", bias), ("offset", offset), ("mask", mask),
                    ("grad_output", grad_output)):
        if not t.is_cuda:
            raise NotImplementedError("%s must be a GPU tensor: libebfi_hip.so has no CPU path" % name)
    geo = _geometry(input, weight, stride, padding, dilation, dg)
    _check_offset_mask(geo, offset, mask)
    offset, mask, grad_output = offset.contiguous(), mask.contiguous(), grad_output.contiguous()
    gx, gw, gb = torch.empty_like(input), torch.empty_like(weight), torch.empty_like(bias)
    go, gm = torch.empty_like(offset), torch.empty_like(mask)
    lib = N.lib()
    need = lib.ebfi_dcn_backward_workspace(*geo, N.dtype_code(input))
    ws = torch.empty(max(int(need), 4), dtype=torch.uint8, device=input.device)
    with torch.cuda.device_of(input):
        rc = lib.ebfi_dcn_backward(N.ptr(input), N.ptr(weight), N.ptr(bias), N.ptr(offset), N.ptr(mask),
                                   N.ptr(grad_output), N.ptr(gx), N.ptr(go), N.ptr(gm), N.ptr(gw), N.ptr(gb),
                                   *geo, N.ptr(ws), int(need), N.dtype_code(input), N.stream_ptr(input.device))
    N.check(rc, "ebfi_dcn_backward")
    return gx, go, gm, gw, gb


class _DCNv2(Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups):
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.kernel_size = _pair(weight.shape[2:4])
        ctx.deformable_groups = deformable_groups
        output = dcn_v2_forward(input, weight, bias, offset, mask, ctx.stride, ctx.padding, ctx.dilation,
                                deformable_groups)
        ctx.save_for_backward(input, offset, mask, weight, bias)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask, weight, bias = ctx.saved_tensors
        gx, go, gm, gw, gb = dcn_v2_backward(input, weight, bias, offset, mask, grad_output, ctx.stride,
                                             ctx.padding, ctx.dilation, ctx.deformable_groups)
        return gx, go, gm, gw, gb, None, None, None, None


dcn_v2_conv = _DCNv2.apply


class DCNv2(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        fan = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        bound = 1.0 / math.sqrt(fan)
        self.weight.data.uniform_(-bound, bound)
        self.bias.data.zero_()

    def _taps(self):
        return self.deformable_groups * self.kernel_size[0] * self.kernel_size[1]

    def _apply_op(self, input, offset, mask):
        return dcn_v2_conv(input, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                           self.deformable_groups)

    def forward(self, input, offset, mask):
        assert 2 * self._taps() == offset.shape[1]
        assert self._taps() == mask.shape[1]
        return self._apply_op(input, offset, mask)


class _SelfOffsetMixin:
    """conv_offset_mask: zero-initialised conv producing 3*dg*kh*kw channels, split in thirds as
    (o1, o2, mask); offset = cat(o1, o2), mask = sigmoid(mask)  (dcn_v2.py:165-182)."""

    def _make_offset_conv(self):
        self.conv_offset_mask = nn.Conv2d(self.in_channels, 3 * self._taps(), kernel_size=self.kernel_size,
                                          stride=self.stride, padding=self.padding, bias=True)
        self.init_offset()

    def init_offset(self):
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    def _offset_mask(self, feat):
        o1, o2, mask = torch.chunk(self.conv_offset_mask(feat), 3, dim=1)
        return torch.cat((o1, o2), dim=1), torch.sigmoid(mask)


class DCN(DCNv2, _SelfOffsetMixin):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        self._make_offset_conv()

    def forward(self, input):
        offset, mask = self._offset_mask(input)
        return self._apply_op(input, offset, mask)


class DCN_sep(DCNv2, _SelfOffsetMixin):
    """Offsets and masks come from a second feature map (dcn_v2.py:197-227)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        self._make_offset_conv()

    def forward(self, input, fea):
        offset, mask = self._offset_mask(fea)
        offset_mean = torch.mean(torch.abs(offset))
        # the reference's magnitude warning is a host sync (:221-223): skipped while a hipGraph is being captured
        if not (offset.is_cuda and torch.cuda.is_current_stream_capturing()) and offset_mean > 100:
            logger.warning("Offset mean is {}, larger than 100.".format(offset_mean))
        return self._apply_op(input, offset, mask)
