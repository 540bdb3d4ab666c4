"""fp16 operand storage of the backward pass: the "c16" images (csrc/c16.hpp, include/ebfi_hip.h).

A tensor [B, C, H, W] (C % 16 == 0, W % 4 == 0) is kept as fp16 [B, C/16, H, 2, W, 8] (16-channel blocks; a row holds the
channels 0..7 of its pixels, then the channels 8..15) times the power-of-two scale of its slot in the scale book
(ebfi_amd.f16scale) -- the layout both fp16 backward kernels stage, so their producers copy 16-byte pieces instead of
converting fp32 planes.  Images are WRITTEN by the kernels that produce the tensors (conv epilogues, the fused ResidualControl
stages, `to_c16` for everything else); this module only allocates them and wraps the standalone conversion.
"""
import torch

from . import _native as N


def empty(B, C, H, W, device):
    if C % 16 != 0:
        raise ValueError("c16 images hold multiples of 16 channels (got %d)" % C)
    if W % 4 != 0:
        raise ValueError("c16 images need rows of whole pixel quads (W = %d)" % W)
    if N.dev_env("EBFI_C16_POISON") == "1":            # (development: every element a writer misses shows up as NaN)
        return torch.full((B, C // 16, H, 2, W, 8), float("nan"), dtype=torch.float16, device=device)
    return torch.empty((B, C // 16, H, 2, W, 8), dtype=torch.float16, device=device)


def to_c16(x, slot_ptr, mask_y=None, mask_slope=0.0, out=None):
    """fp32 [B,C,H,W] -> image scaled by the slot's scale (|max| recorded into the slot); `mask_y`: multiply by the
    LeakyReLU(mask_slope) derivative of that tensor first (a pre-activation gradient)."""
    N.require_gpu(x)
    x = x.contiguous()
    B, C, H, W = (int(v) for v in x.shape)
    out = empty(B, C, H, W, x.device) if out is None else out
    if x.numel() == 0:
        return out
    with torch.cuda.device_of(x):
        rc = N.lib().ebfi_to_c16(N.ptr(x), N.ptr(mask_y.contiguous() if mask_y is not None else None), float(mask_slope),
                                 N.ptr(out), slot_ptr, B, C, H, W, N.stream_ptr(x.device))
    N.check(rc, "ebfi_to_c16")
    return out


def to_c16_cat2(a, b, slot_ptr):
    """The image of cat([a, b], 1) from its two fp32 parts (the concatenated tensor is never written); bit-identical to
    to_c16(torch.cat([a, b], 1), slot_ptr)."""
    N.require_gpu(a)
    a, b = a.contiguous(), b.contiguous()
    B, C0, H, W = (int(v) for v in a.shape)
    C1 = int(b.shape[1])
    if b.shape[0] != B or tuple(b.shape[2:]) != (H, W):
        raise ValueError("to_c16_cat2: parts of %s and %s" % (tuple(a.shape), tuple(b.shape)))
    out = empty(B, C0 + C1, H, W, a.device)
    if out.numel() == 0:
        return out
    with torch.cuda.device_of(a):
        rc = N.lib().ebfi_to_c16_cat2(N.ptr(a), C0, N.ptr(b), C1, N.ptr(out), slot_ptr, B, H, W, N.stream_ptr(a.device))
    N.check(rc, "ebfi_to_c16_cat2")
    return out


def from_c16(img, scale=1.0):
    """Image -> fp32 [B,C,H,W] / scale (tests and diagnostics; plain torch)."""
    B, CB, H, _, W, _ = img.shape
    return img.permute(0, 1, 3, 5, 2, 4).reshape(B, CB * 16, H, W).float() / scale
