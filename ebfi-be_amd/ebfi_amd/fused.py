"""Autograd wrappers of the fused inter-convolution stages (csrc/fuse.hip)."""
import torch

from . import _native as N


class _ScaleResidualCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a0, s0, a1, s1, x):
        a0, a1, x = a0.contiguous(), a1.contiguous(), x.contiguous()
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // max(B * C, 1)
        s0c, s1c = s0.reshape(B, C).contiguous(), s1.reshape(B, C).contiguous()
        out = torch.empty((B, 2 * C) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
        with torch.cuda.device_of(x):
            rc = N.lib().ebfi_scale_residual_cat_forward(N.ptr(a0), N.ptr(s0c), N.ptr(a1), N.ptr(s1c), N.ptr(x), N.ptr(out), B, C, HW,
                                                         N.stream_ptr(x.device))
        N.check(rc, "ebfi_scale_residual_cat_forward")
        ctx.save_for_backward(a0, s0c, a1, s1c)
        ctx.s_shapes = (s0.shape, s1.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        a0, s0c, a1, s1c = ctx.saved_tensors
        g = g.contiguous()
        B, C = a0.shape[0], a0.shape[1]
        HW = a0.numel() // max(B * C, 1)
        ga0, ga1, gx = torch.empty_like(a0), torch.empty_like(a0), torch.empty_like(a0)
        gs0, gs1 = torch.empty_like(s0c), torch.empty_like(s1c)
        with torch.cuda.device_of(g):
            rc = N.lib().ebfi_scale_residual_cat_backward(N.ptr(g), N.ptr(a0), N.ptr(s0c), N.ptr(a1), N.ptr(s1c), N.ptr(ga0),
                                                          N.ptr(ga1), N.ptr(gx), N.ptr(gs0), N.ptr(gs1), B, C, HW,
                                                          N.stream_ptr(g.device))
        N.check(rc, "ebfi_scale_residual_cat_backward")
        return ga0, gs0.view(ctx.s_shapes[0]), ga1, gs1.view(ctx.s_shapes[1]), gx


def scale_residual_cat(a0, s0, a1, s1, x):
    """cat([s0*a0 + x, s1*a1 + x], 1) for [B,C,H,W] maps and per-(sample, channel) scales s* [B,C,1,1]."""
    if x.is_cuda and x.dtype == torch.float32 and a0.dtype == torch.float32 and a1.dtype == torch.float32 and x.dim() == 4 and \
            (x.shape[2] * x.shape[3]) % 4 == 0 and s0.numel() == x.shape[0] * x.shape[1] == s1.numel() and \
            a0.shape == x.shape == a1.shape and not torch.is_autocast_enabled():
        return _ScaleResidualCat.apply(a0, s0.float(), a1, s1.float(), x)
    return torch.cat([s0 * a0 + x, s1 * a1 + x], dim=1)
