"""Autograd wrappers of the fused inter-convolution stages (csrc/fuse.hip)."""
import torch

from . import _native as N


class _ScaleResidualCat(torch.autograd.Function):
    """s1 / x may be None: unit scale on the second half / no residual."""

    @staticmethod
    def forward(ctx, a0, s0, a1, s1, x):
        a0, a1 = a0.contiguous(), a1.contiguous()
        x = x.contiguous() if x is not None else None
        B, C = a0.shape[0], a0.shape[1]
        HW = a0.numel() // max(B * C, 1)
        s0c = s0.reshape(B, C).contiguous()
        s1c = s1.reshape(B, C).contiguous() if s1 is not None else None
        out = torch.empty((B, 2 * C) + tuple(a0.shape[2:]), dtype=a0.dtype, device=a0.device)
        with torch.cuda.device_of(a0):
            rc = N.lib().ebfi_scale_residual_cat_forward(N.ptr(a0), N.ptr(s0c), N.ptr(a1), N.ptr(s1c), N.ptr(x), N.ptr(out), B, C, HW,
                                                         N.stream_ptr(a0.device))
        N.check(rc, "ebfi_scale_residual_cat_forward")
        ctx.save_for_backward(a0, s0c, a1 if s1 is not None else None, s1c)
        ctx.meta = (s0.shape, s1.shape if s1 is not None else None, x is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        a0, s0c, a1, s1c = ctx.saved_tensors
        s0_shape, s1_shape, has_x = ctx.meta
        g = g.contiguous()
        B, C = a0.shape[0], a0.shape[1]
        HW = a0.numel() // max(B * C, 1)
        ga0, ga1 = torch.empty_like(a0), torch.empty_like(a0)
        gx = torch.empty_like(a0) if has_x else None
        gs0 = torch.empty_like(s0c)
        gs1 = torch.empty_like(s1c) if s1c is not None else None
        with torch.cuda.device_of(g):
            rc = N.lib().ebfi_scale_residual_cat_backward(N.ptr(g), N.ptr(a0), N.ptr(s0c), N.ptr(a1), N.ptr(s1c), N.ptr(ga0),
                                                          N.ptr(ga1), N.ptr(gx), N.ptr(gs0), N.ptr(gs1), B, C, HW,
                                                          N.stream_ptr(g.device))
        N.check(rc, "ebfi_scale_residual_cat_backward")
        return ga0, gs0.view(s0_shape), ga1, (gs1.view(s1_shape) if gs1 is not None else None), gx


class _ScalarConvBank(torch.autograd.Function):
    """A bank of S ConvLayer(k=1) + LeakyReLU layers on one [B, K] input in ONE launch each way (csrc/fuse.hip scalar_conv_*):
    ResidualControl's scalar-conditioned channel scales Conv1[i](Ex) / Conv2[i](T) (model_singleframe.py:85-94, :127-129).
    forward(v, slope, w_0, b_0, ..., w_{S-1}, b_{S-1}) -> [S, B, C]; the gradients of the S weights / biases are views of two
    packed tensors."""

    @staticmethod
    def forward(ctx, v, slope, *params):
        import ctypes
        ws, bs = params[0::2], params[1::2]
        S, (B, K), C = len(ws), v.shape, ws[0].shape[0]
        v = v.contiguous()
        # (the kernel indexes weight_s[c * K + k]: a [C, K, 1, 1] conv weight as it is stored)
        ws = [w.contiguous() for w in ws]
        bs = [None if b is None else b.contiguous() for b in bs]
        wp = (ctypes.c_void_p * S)(*[w.data_ptr() for w in ws])
        bp = (ctypes.c_void_p * S)(*[None if b is None else b.data_ptr() for b in bs])
        out = torch.empty(S, B, C, dtype=v.dtype, device=v.device)
        with torch.cuda.device_of(v):
            rc = N.lib().ebfi_scalar_conv_forward(N.ptr(v), wp, bp, N.ptr(out), S, B, K, C, float(slope), N.stream_ptr(v.device))
        N.check(rc, "ebfi_scalar_conv_forward")
        ctx.save_for_backward(v, out, *ws)
        ctx.meta = (float(slope), [w.shape for w in params[0::2]], [b is not None for b in bs])
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes
        v, out = ctx.saved_tensors[:2]
        ws = ctx.saved_tensors[2:]
        slope, wshapes, has_b = ctx.meta
        S, (B, K), C = len(ws), v.shape, out.shape[2]
        g = g.contiguous()
        wp = (ctypes.c_void_p * S)(*[w.data_ptr() for w in ws])
        gw = torch.empty(S, C, K, dtype=v.dtype, device=v.device) if any(ctx.needs_input_grad[2::2]) else None
        gb = torch.empty(S, C, dtype=v.dtype, device=v.device) if any(ctx.needs_input_grad[3::2]) else None
        gv = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        with torch.cuda.device_of(g):
            rc = N.lib().ebfi_scalar_conv_backward(N.ptr(v), wp, N.ptr(out), N.ptr(g), N.ptr(gw), N.ptr(gb), N.ptr(gv), S, B, K, C, slope,
                                                   N.stream_ptr(g.device))
        N.check(rc, "ebfi_scalar_conv_backward")
        grads = []
        for s in range(S):
            grads.append(gw[s].view(wshapes[s]) if gw is not None and ctx.needs_input_grad[2 + 2 * s] else None)
            grads.append(gb[s] if gb is not None and has_b[s] and ctx.needs_input_grad[3 + 2 * s] else None)
        return (gv, None) + tuple(grads)


def scalar_conv_usable(v, weights):
    """[B, K] fp32 GPU input with K <= 8 into <= 32 layers of one shape [C, K(,1,1)]: what ebfi_scalar_conv_* takes."""
    w0 = weights[0]
    if N.dev_env("EBFI_NO_SCALAR_CONV", "0") == "1":       # (development switch: the torch einsum form, for A/B runs)
        return False
    return (v.is_cuda and v.dtype == torch.float32 and v.dim() == 2 and 1 <= v.shape[1] <= 8 and 1 <= len(weights) <= 32 and
            all(w.dtype == torch.float32 and w.is_cuda and w.shape == w0.shape and w.numel() == w0.shape[0] * v.shape[1] for w in weights) and
            len(weights) * v.shape[0] * w0.shape[0] <= (1 << 20) and not torch.is_autocast_enabled())


def scalar_conv_bank(v, weights, biases, slope):
    """leaky_relu(bias_s + v @ weight_s^T, slope) for every layer s of the bank -> [S, B, C]."""
    if scalar_conv_usable(v, weights):
        params = []
        for w, b in zip(weights, biases):
            params += [w, b]
        return _ScalarConvBank.apply(v, slope, *params)
    w = torch.stack([x.flatten(1) for x in weights])                     # [S, C, K]
    s = torch.einsum("bk,sck->sbc", v, w)
    if biases[0] is not None:
        s = s + torch.stack(list(biases))[:, None, :]
    return torch.nn.functional.leaky_relu(s, slope)


def _fusable(a0, a1, s0):
    return (a0.is_cuda and a0.dtype == torch.float32 and a1.dtype == torch.float32 and a0.dim() == 4 and a0.shape == a1.shape and
            (a0.shape[2] * a0.shape[3]) % 4 == 0 and s0.numel() == a0.shape[0] * a0.shape[1] and not torch.is_autocast_enabled())


def scale_residual_cat(a0, s0, a1, s1, x):
    """cat([s0*a0 + x, s1*a1 + x], 1) for [B,C,H,W] maps and per-(sample, channel) scales s* [B,C,1,1]."""
    if _fusable(a0, a1, s0) and s1.numel() == s0.numel() and x.shape == a0.shape and x.dtype == torch.float32:
        return _ScaleResidualCat.apply(a0, s0.float(), a1, s1.float(), x)
    return torch.cat([s0 * a0 + x, s1 * a1 + x], dim=1)


def scale_cat(a0, s0, a1):
    """cat([s0*a0, a1], 1): the channel-attention + concat stage of ExposureDecision (model_singleframe.py:68-72)."""
    if _fusable(a0, a1, s0):
        return _ScaleResidualCat.apply(a0, s0.float(), a1, None, None)
    return torch.cat([s0 * a0, a1], dim=1)


class _ProdMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        B, C = a.shape[0], a.shape[1]
        HW = a.numel() // max(B * C, 1)
        out = torch.empty((B, C, 1, 1), dtype=a.dtype, device=a.device)
        with torch.cuda.device_of(a):
            rc = N.lib().ebfi_prodmean_forward(N.ptr(a), N.ptr(b), N.ptr(out), B * C, HW, N.stream_ptr(a.device))
        N.check(rc, "ebfi_prodmean_forward")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        B, C = a.shape[0], a.shape[1]
        HW = a.numel() // max(B * C, 1)
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        with torch.cuda.device_of(a):
            rc = N.lib().ebfi_prodmean_backward(N.ptr(a), N.ptr(b), N.ptr(g), N.ptr(ga), N.ptr(gb), B * C, HW, N.stream_ptr(a.device))
        N.check(rc, "ebfi_prodmean_backward")
        return ga, gb


def product_mean(a, b):
    """AdaptiveAvgPool2d(1)(a * b) -> [B,C,1,1] without materialising the product."""
    if a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 4 and a.shape == b.shape and \
            (a.shape[2] * a.shape[3]) % 4 == 0 and not torch.is_autocast_enabled():
        return _ProdMean.apply(a, b)
    return (a * b).mean(dim=(2, 3), keepdim=True)


class _EDHead(torch.autograd.Function):
    """cat([ev * sigmoid(AVGPool(GN(ev) * GN(bl))), bl], 1) of ExposureDecision (model_singleframe.py:66-72) as one node on
    csrc/edhead.hip: plane moments instead of normalised maps forward, closed-form gradients of both maps backward."""

    @staticmethod
    def forward(ctx, ev, bl, gamma, beta, groups, eps):
        ev, bl = ev.contiguous(), bl.contiguous()
        B, C = ev.shape[0], ev.shape[1]
        HW = ev.numel() // (B * C)
        lib = N.lib()
        out = torch.empty((B, 2 * C) + tuple(ev.shape[2:]), dtype=torch.float32, device=ev.device)
        atten = torch.empty(B * C, dtype=torch.float32, device=ev.device)
        stats = torch.empty(B * C * 5 + B * groups * 4, dtype=torch.float64, device=ev.device)
        need = int(lib.ebfi_ed_head_workspace(B, C, HW))
        ws = torch.empty(need, dtype=torch.uint8, device=ev.device)
        with torch.cuda.device_of(ev):
            rc = lib.ebfi_ed_head_forward(N.ptr(ev), N.ptr(bl), N.ptr(gamma), N.ptr(beta), N.ptr(out), N.ptr(atten), N.ptr(stats),
                                          B, C, HW, groups, float(eps), N.ptr(ws), need, N.stream_ptr(ev.device))
        N.check(rc, "ebfi_ed_head_forward")
        ctx.groups = groups
        ctx.save_for_backward(ev, bl, gamma, beta, atten, stats)
        return out

    @staticmethod
    def backward(ctx, g):
        ev, bl, gamma, beta, atten, stats = ctx.saved_tensors
        g = g.contiguous()
        B, C = ev.shape[0], ev.shape[1]
        HW = ev.numel() // (B * C)
        lib = N.lib()
        gev, gbl = torch.empty_like(ev), torch.empty_like(bl)
        affine = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        gg = torch.empty(C, dtype=torch.float32, device=ev.device) if affine else None
        gb = torch.empty(C, dtype=torch.float32, device=ev.device) if affine else None
        need = int(lib.ebfi_ed_head_workspace(B, C, HW))
        ws = torch.empty(need, dtype=torch.uint8, device=ev.device)
        with torch.cuda.device_of(ev):
            rc = lib.ebfi_ed_head_backward(N.ptr(g), N.ptr(ev), N.ptr(bl), N.ptr(gamma), N.ptr(beta), N.ptr(atten), N.ptr(stats),
                                           N.ptr(gev), N.ptr(gbl), N.ptr(gg), N.ptr(gb), B, C, HW, ctx.groups, N.ptr(ws), need,
                                           N.stream_ptr(ev.device))
        N.check(rc, "ebfi_ed_head_backward")
        return gev, gbl, gg, gb, None, None


def ed_head(ev, bl, gn):
    """`gn`: the nn.GroupNorm ExposureDecision applies to both maps.  None when the tensors do not qualify."""
    import os
    if not (ev.is_cuda and ev.dtype == torch.float32 and bl.dtype == torch.float32 and ev.dim() == 4 and ev.shape == bl.shape and
            (ev.shape[2] * ev.shape[3]) % 4 == 0 and gn.weight is not None and gn.bias is not None and ev.shape[1] <= 1024 and
            ev.shape[1] == gn.num_channels and not torch.is_autocast_enabled() and N.dev_env("EBFI_NO_EDHEAD") is None):
        return None
    return _EDHead.apply(ev, bl, gn.weight, gn.bias, gn.num_groups, gn.eps)


class _Pad2d(torch.autograd.Function):
    """nn.ReflectionPad2d(p) / nn.ReplicationPad2d(p) whose BACKWARD is a gather in a fixed order (csrc/imgops.hip pad2d_bwd):
    torch's own backward of either pad scatters with atomic adds, which made two runs of the same training step differ in the
    last bit (and, amplified by the L1 / census kinks over a few optimiser steps, by percent in the smallest parameters)."""

    @staticmethod
    def forward(ctx, x, p, mode):
        ctx.p, ctx.shape, ctx.mode = int(p), tuple(x.shape), mode
        return torch.nn.functional.pad(x, (p, p, p, p), mode=mode)

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        g = g.contiguous()
        out = torch.empty(ctx.shape, dtype=g.dtype, device=g.device)
        with torch.cuda.device_of(g):
            rc = N.lib().ebfi_pad2d_backward(N.ptr(g), N.ptr(out), B * C, H, W, ctx.p, 1 if ctx.mode == "replicate" else 0,
                                             N.stream_ptr(g.device))
        N.check(rc, "ebfi_pad2d_backward")
        return out, None, None


def _native_pad_ok(x):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4


def reflect_pad2d(x, p):
    """ReflectionPad2d(p) with the deterministic native adjoint on GPU fp32 tensors; plain F.pad otherwise."""
    if _native_pad_ok(x) and p < x.shape[-1] and p < x.shape[-2]:
        return _Pad2d.apply(x, p, "reflect")
    return torch.nn.functional.pad(x, (p, p, p, p), mode="reflect")


def replicate_pad2d(x, p):
    """ReplicationPad2d(p) with the deterministic native adjoint on GPU fp32 tensors; plain F.pad otherwise."""
    if _native_pad_ok(x):
        return _Pad2d.apply(x, p, "replicate")
    return torch.nn.functional.pad(x, (p, p, p, p), mode="replicate")
