"""Depth-2 3-D convolutions of the detail branch as 2-D convolutions on the gfx950 conv kernels.

UNet3d_18 (reference models/Ours/model_singleframe.py:170-223, models/model_misc/resnet_3D.py)
stacks (blurry frame, Sharp) on a depth axis of length 2 and never changes that length (temporal
stride 1, depth padding 1 for depth-3 kernels).  A contiguous [B, C, 2, H, W] tensor IS a
[B, 2C, H, W] tensor (channel = c*2 + d), so

  Conv3d k=(3,kh,kw), pad_d=1     ==  Conv2d with W2[(co,d),(ci,din)] = W[co,ci,din-d+1]
  Conv3d k=(1,1,1)                ==  Conv2d 1x1 with a block-diagonal weight
  ConvTranspose3d k=(3,4,4), stride (1,2,2), pad (1,1,1)
                                  ==  Conv2d 3x3 (pad 1) to 8*Cout channels + PixelShuffle(2):
                                      output parity (py,px) picks 2x2 of the 4x4 taps

The folds are 0/1 linear maps on the weights (autograd routes the gradient back to the original
Conv3d / ConvTranspose3d parameters, so state_dict and optimizer are untouched).  ``fold_*_weight`` below
state them as tensor ops; on the GPU they run as ONE gather launch each way (`ebfi_gather_sum`) through
index tables derived once per weight shape by pushing element ids through those same tensor ops.
The convolutions themselves run through ebfi_amd.conv (fused bias + activation).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import _native as N
from . import conv, weightbank


def usable(x):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.shape[2] == 2 and \
        not torch.is_autocast_enabled() and x.shape[-1] * x.shape[-2] >= 2


def fold_conv3d_weight(w):
    """[Co,Ci,kd,kh,kw] -> [2Co,2Ci,kh,kw] for depth-2 inputs with depth padding kd//2."""
    Co, Ci, kd, kh, kw = w.shape
    if kd == 3:
        d0, d1 = w[:, :, 1:3], w[:, :, 0:2]            # out depth 0 sees kd = din+1, depth 1 sees kd = din
    elif kd == 1:
        z = torch.zeros_like(w)
        d0, d1 = torch.cat([w, z], 2), torch.cat([z, w], 2)
    else:
        raise ValueError("depth kernel %d" % kd)
    return torch.stack([d0, d1], dim=1).reshape(2 * Co, 2 * Ci, kh, kw)


_MAPS = {}


def _index_tables(kind, fold_fn, shape, device):
    """(forward idx [n_out] int32, adjoint idx [n_src, R] int32, folded shape) of the 0/1 map `fold_fn`."""
    key = (kind, tuple(shape), str(device))
    if key not in _MAPS:
        n = int(np.prod(shape))
        ids = torch.arange(1, n + 1, dtype=torch.float64).view(*shape)       # element id + 1; 0 = structural zero
        folded = fold_fn(ids)
        fwd = folded.round().to(torch.int64).flatten().numpy() - 1
        dest = np.nonzero(fwd >= 0)[0]
        order = np.argsort(fwd[dest], kind="stable")
        srcs, dest = fwd[dest][order], dest[order]
        counts = np.bincount(srcs, minlength=n)
        R = int(max(1, counts.max()))
        starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
        inv = np.full((n, R), -1, dtype=np.int32)
        inv[srcs, np.arange(len(srcs)) - starts[srcs]] = dest
        _MAPS[key] = (torch.from_numpy(fwd.astype(np.int32)).to(device), torch.from_numpy(inv).to(device), R,
                      tuple(folded.shape))
    return _MAPS[key]


class _LinearMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, fwd_idx, inv_idx, R, out_shape):
        w = w.contiguous()
        out = torch.empty(out_shape, dtype=w.dtype, device=w.device)
        with torch.cuda.device_of(w):
            rc = N.lib().ebfi_gather_sum(N.ptr(w), N.ptr(fwd_idx), N.ptr(out), out.numel(), 1, N.stream_ptr(w.device))
        N.check(rc, "ebfi_gather_sum")
        ctx.inv, ctx.R, ctx.in_shape = inv_idx, R, w.shape
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        gw = torch.empty(ctx.in_shape, dtype=g.dtype, device=g.device)
        with torch.cuda.device_of(g):
            rc = N.lib().ebfi_gather_sum(N.ptr(g), N.ptr(ctx.inv), N.ptr(gw), gw.numel(), ctx.R, N.stream_ptr(g.device))
        N.check(rc, "ebfi_gather_sum")
        return gw, None, None, None, None


def folded(kind, fold_fn, w):
    """`fold_fn(w)`; for fp32 GPU tensors through the cached index tables (one launch forward, one backward)."""
    if not (w.is_cuda and w.dtype == torch.float32):
        return fold_fn(w)
    fwd_idx, inv_idx, R, out_shape = _index_tables(kind, fold_fn, w.shape, w.device)
    return _LinearMap.apply(w, fwd_idx, inv_idx, R, out_shape)


def _rep2(b):
    return b.repeat_interleave(2)


def _rep8(b):
    return b.repeat_interleave(8)


def conv3d_d2(x, m, act=conv.ACT_NONE, slope=0.0):
    """x [B,Ci,2,H,W] through nn.Conv3d `m` (+ fused activation) -> [B,Co,2,Ho,Wo]."""
    B, Ci, D, H, W = x.shape
    kd, kh, kw = m.kernel_size
    assert D == 2 and m.stride[0] == 1 and m.padding[0] == kd // 2 and kh == kw and m.stride[1] == m.stride[2]
    assert m.padding[1] == m.padding[2] and m.dilation == (1, 1, 1) and m.groups == 1
    x2 = x.reshape(B, 2 * Ci, H, W)
    s = m.stride[1]
    if kh == 1 and s != 1:                               # 1x1 strided shortcut: subsample, then 1x1
        x2 = x2[:, :, ::s, ::s].contiguous()
        s = 1
    site = weightbank.lookup(m.weight, "conv3d")         # folded images kept in the weight bank (one pack launch per step)
    if conv.site_usable(site, x2, s):
        y2 = conv.conv_site(x2, site, m.padding[1], act, slope, [m.weight], [m.bias] if m.bias is not None else [])
        return y2.view(B, m.out_channels, 2, y2.shape[-2], y2.shape[-1])
    b2 = folded("rep2", _rep2, m.bias) if m.bias is not None else None
    y2 = conv.conv_bias_act(x2, folded("conv3d", fold_conv3d_weight, m.weight), b2, s, m.padding[1], act, slope)
    return y2.view(B, m.out_channels, 2, y2.shape[-2], y2.shape[-1])


_KY = {}


def _tap_select(device):
    """4-tap transposed-conv kernel -> 3 window rows per output parity: (index, mask) of length 6."""
    if device not in _KY:
        _KY[device] = (torch.tensor([3, 1, 0, 0, 2, 0], device=device),
                       torch.tensor([1., 1., 0., 0., 1., 1.], device=device))
    return _KY[device]


def fold_conv_transpose3d_weight(wt):
    """[Ci,Co,3,4,4] (stride (1,2,2), pad (1,1,1)) -> [8Co, 2Ci, 3, 3]; out channel = ((co*2+d)*2+py)*2+px."""
    Ci, Co, kd, kh, kw = wt.shape
    assert (kd, kh, kw) == (3, 4, 4)
    idx, mask = _tap_select(wt.device)
    wd = torch.stack([wt[:, :, [1, 0]], wt[:, :, [2, 1]]], dim=2)           # [Ci,Co,d,din,4,4], kd = d-din+1
    wy = wd.index_select(4, idx) * mask.view(6, 1)                           # [Ci,Co,d,din,(py,dy),4]
    wx = wy.index_select(5, idx) * mask                                      # [Ci,Co,d,din,(py,dy),(px,dx)]
    w8 = wx.view(Ci, Co, 2, 2, 2, 3, 2, 3).permute(1, 2, 4, 6, 0, 3, 5, 7)   # [Co,d,py,px,Ci,din,dy,dx]
    return w8.reshape(8 * Co, 2 * Ci, 3, 3)


def _conv_transpose3d_unshuffled(x, m):
    """x [B,Ci,2,H,W] through nn.ConvTranspose3d `m` as the folded 3x3 convolution: [B, 8*Co, H, W], channel = (co, d, py, px)."""
    B, Ci, D, H, W = x.shape
    assert D == 2 and m.kernel_size == (3, 4, 4) and m.stride == (1, 2, 2) and m.padding == (1, 1, 1)
    assert m.output_padding == (0, 0, 0) and m.dilation == (1, 1, 1) and m.groups == 1
    site = weightbank.lookup(m.weight, "convT3d")
    if conv.site_usable(site, x.reshape(B, 2 * Ci, H, W)):
        return conv.conv_site(x.reshape(B, 2 * Ci, H, W), site, 1, conv.ACT_NONE, 0.0, [m.weight], [m.bias] if m.bias is not None else [])
    b8 = folded("rep8", _rep8, m.bias) if m.bias is not None else None
    return conv.conv_bias_act(x.reshape(B, 2 * Ci, H, W), folded("convT3d", fold_conv_transpose3d_weight, m.weight), b8, 1, 1,
                              conv.ACT_NONE, 0.0)


def conv_transpose3d_d2(x, m):
    """x [B,Ci,2,H,W] through nn.ConvTranspose3d `m` -> [B,Co,2,2H,2W]."""
    B, Ci, D, H, W = x.shape
    return F.pixel_shuffle(_conv_transpose3d_unshuffled(x, m), 2).view(B, m.out_channels, 2, 2 * H, 2 * W)


class _SEGateShuffled(torch.autograd.Function):
    """_SEGate on the UNSHUFFLED output of a folded transposed convolution: y2 [B, 8C, h, w] -> act(gate * shuffle(y2))
    [B, C, 2, 2h, 2w]; the gradient returns in y2's layout (csrc/segate.hip, PsGeom).  The PixelShuffle copy never runs."""

    @staticmethod
    def forward(ctx, y2, weight, bias, act, slope):
        y2 = y2.contiguous()
        B, C8, h, w = (int(v) for v in y2.shape)
        C = C8 // 8
        n = 8 * h * w
        w2 = weight.reshape(C, C).contiguous()
        out = torch.empty(B, C, 2, 2 * h, 2 * w, dtype=y2.dtype, device=y2.device)
        mean = torch.empty(B * C, dtype=y2.dtype, device=y2.device)
        gate = torch.empty(B * C, dtype=y2.dtype, device=y2.device)
        ws = torch.empty(max(int(N.lib().ebfi_se_gate_workspace(B, C, n)), 1), dtype=y2.dtype, device=y2.device)
        with torch.cuda.device_of(y2):
            rc = N.lib().ebfi_se_gate_forward_ps(N.ptr(y2), N.ptr(w2), N.ptr(bias), N.ptr(out), N.ptr(mean), N.ptr(gate), N.ptr(ws),
                                                 B, C, h, w, act, slope, N.stream_ptr(y2.device))
        N.check(rc, "ebfi_se_gate_forward_ps")
        ctx.cfg = (B, C, h, w, act, slope, bias is not None, weight.shape)
        ctx.save_for_backward(y2, w2, gate, mean, out if act != 0 else None)
        return out

    @staticmethod
    def backward(ctx, g):
        y2, w2, gate, mean, out = ctx.saved_tensors
        B, C, h, w, act, slope, has_bias, wshape = ctx.cfg
        g = g.contiguous()
        gy = torch.empty_like(y2)
        gw = torch.empty_like(w2)
        gb = torch.empty(C, dtype=y2.dtype, device=y2.device) if has_bias else None
        ws = torch.empty(max(int(N.lib().ebfi_se_gate_workspace(B, C, 8 * h * w)), 1), dtype=y2.dtype, device=y2.device)
        with torch.cuda.device_of(y2):
            rc = N.lib().ebfi_se_gate_backward_ps(N.ptr(g), N.ptr(out), N.ptr(y2), N.ptr(w2), N.ptr(gate), N.ptr(mean), N.ptr(gy),
                                                  N.ptr(gw), N.ptr(gb), N.ptr(ws), B, C, h, w, act, slope, N.stream_ptr(y2.device))
        N.check(rc, "ebfi_se_gate_backward_ps")
        return gy, gw.view(wshape), gb, None, None


def conv_transpose3d_se(x, m, attn_conv, act=0, slope=0.0):
    """upConv3D's transposed convolution + SEGating (+ LeakyReLU) (model_3DUnet / resnet_3D upConv3D; the decoder stages of
    UNet3d_18, model_singleframe.py:200-221): the gate reads the folded convolution's output through the pixel shuffle."""
    B, Ci, D, H, W = x.shape
    C = m.out_channels
    if N.dev_env("EBFI_NO_SEGATE_SHUFFLE", "0") != "1" and N.dev_env("EBFI_NO_SEGATE", "0") != "1" and x.is_cuda \
            and x.dtype == torch.float32 and W % 2 == 0 and B * C <= 4096 and attn_conv.in_channels == attn_conv.out_channels == C \
            and not torch.is_autocast_enabled():
        return _SEGateShuffled.apply(_conv_transpose3d_unshuffled(x, m), attn_conv.weight, attn_conv.bias, int(act), float(slope))
    return se_gate(conv_transpose3d_d2(x, m), attn_conv, None, act, slope)


class _SEGate(torch.autograd.Function):
    """act(x * sigmoid(W mean(x) + b) (+ res)) as two launches forward, two backward (csrc/segate.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, act, slope):
        x = x.contiguous()
        res = res.contiguous() if res is not None else None
        B, C = int(x.shape[0]), int(x.shape[1])
        n = x.numel() // (B * C)
        w2 = weight.reshape(C, C).contiguous()
        out = torch.empty_like(x)
        mean = torch.empty(B * C, dtype=x.dtype, device=x.device)
        gate = torch.empty(B * C, dtype=x.dtype, device=x.device)
        ws = torch.empty(max(int(N.lib().ebfi_se_gate_workspace(B, C, n)), 1), dtype=x.dtype, device=x.device)
        with torch.cuda.device_of(x):
            rc = N.lib().ebfi_se_gate_forward(N.ptr(x), N.ptr(w2), N.ptr(bias), N.ptr(res), N.ptr(out), N.ptr(mean), N.ptr(gate),
                                              N.ptr(ws), B, C, n, act, slope, N.stream_ptr(x.device))
        N.check(rc, "ebfi_se_gate_forward")
        ctx.cfg = (B, C, n, act, slope, res is not None, bias is not None, weight.shape)
        ctx.save_for_backward(x, w2, gate, mean, out if act != 0 else None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w2, gate, mean, out = ctx.saved_tensors
        B, C, n, act, slope, has_res, has_bias, wshape = ctx.cfg
        g = g.contiguous()
        gx = torch.empty_like(x)
        gres = torch.empty_like(x) if has_res else None
        gw = torch.empty_like(w2)
        gb = torch.empty(C, dtype=x.dtype, device=x.device) if has_bias else None
        ws = torch.empty(max(int(N.lib().ebfi_se_gate_workspace(B, C, n)), 1), dtype=x.dtype, device=x.device)
        with torch.cuda.device_of(x):
            rc = N.lib().ebfi_se_gate_backward(N.ptr(g), N.ptr(out), N.ptr(x), N.ptr(w2), N.ptr(gate), N.ptr(mean), N.ptr(gx), N.ptr(gres),
                                               N.ptr(gw), N.ptr(gb), N.ptr(ws), B, C, n, act, slope, N.stream_ptr(x.device))
        N.check(rc, "ebfi_se_gate_backward")
        return gx, gw.view(wshape), gb, gres, None, None


def se_gate(x, attn_conv, res=None, act=0, slope=0.0):
    """SEGating (resnet_3D.py:89-105) x * sigmoid(W @ mean(x) + b), optionally followed by `+ res` and an activation
    (act 1 = LeakyReLU(slope), slope 0 = ReLU): what BasicBlock (:137-141) and the decoder stages of UNet3d_18
    (model_singleframe.py:213-221) do right after the gate."""
    B, C = x.shape[0], x.shape[1]
    n = x.numel() // max(B * C, 1)
    import os
    if N.dev_env("EBFI_NO_SEGATE", "0") != "1" and \
            x.is_cuda and x.dtype == torch.float32 and n % 4 == 0 and B * C <= 4096 and attn_conv.in_channels == attn_conv.out_channels == C \
            and (res is None or res.shape == x.shape) and not torch.is_autocast_enabled():
        return _SEGate.apply(x, attn_conv.weight, attn_conv.bias, res, int(act), float(slope))
    pooled = x.mean(dim=(2, 3, 4))
    w = attn_conv.weight.view(attn_conv.out_channels, attn_conv.in_channels)
    gate = torch.sigmoid(F.linear(pooled, w, attn_conv.bias))
    y = x * gate[:, :, None, None, None]
    if res is not None:
        y = y + res
    return F.leaky_relu(y, slope) if act == 1 else y
