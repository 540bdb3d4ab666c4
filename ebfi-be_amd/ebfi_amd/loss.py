"""Training loss of the hot path (device side).

Reference: loss/restore.py:149-213 (LaplacianLoss: 5-level Laplacian pyramid, L1 *sum*, level
weight 2**i), :111-145 (Ternary census, 7x7), combined as in train_ours.py:258-268.  On the GPU the 5x5
blur and the whole census term run as kernel pairs of libebfi_hip.so (csrc/imgops.hip); CPU tensors (host-logic
tests) take the equivalent shifted-slice formulation below.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native as N


class _Gauss5(torch.autograd.Function):
    """5x5 binomial blur with reflect padding on the gfx950 kernel pair (csrc/imgops.hip)."""

    @staticmethod
    def forward(ctx, x, factor):
        x = x.contiguous()
        ctx.factor = float(factor)
        out = torch.empty_like(x)
        H, W = x.shape[-2:]
        with torch.cuda.device_of(x):
            rc = N.lib().ebfi_gauss5_forward(N.ptr(x), N.ptr(out), x.numel() // (H * W), H, W, ctx.factor,
                                             N.stream_ptr(x.device))
        N.check(rc, "ebfi_gauss5_forward")
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        gin = torch.empty_like(g)
        H, W = g.shape[-2:]
        with torch.cuda.device_of(g):
            rc = N.lib().ebfi_gauss5_backward(N.ptr(g), N.ptr(gin), g.numel() // (H * W), H, W, ctx.factor,
                                              N.stream_ptr(g.device))
        N.check(rc, "ebfi_gauss5_backward")
        return gin, None


class _Census(torch.autograd.Function):
    """Whole Ternary loss (transform of both images, distance, mask, mean) on the kernel pair of csrc/imgops.hip."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = x.contiguous(), y.contiguous()
        B, C, H, W = x.shape
        lib = N.lib()
        partial = torch.empty(int(lib.ebfi_census_partials(B, H, W)), dtype=torch.float32, device=x.device)
        with torch.cuda.device_of(x):
            rc = lib.ebfi_census_forward(N.ptr(x), N.ptr(y), N.ptr(partial), B, C, H, W, N.stream_ptr(x.device))
        N.check(rc, "ebfi_census_forward")
        ctx.save_for_backward(x, y)
        return partial.sum() / float(B * H * W)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        B, C, H, W = x.shape
        g = g.contiguous().float().reshape(1)
        gx = torch.empty_like(x)
        with torch.cuda.device_of(x):
            rc = N.lib().ebfi_census_backward(N.ptr(x), N.ptr(y), N.ptr(g), N.ptr(gx), B, C, H, W, N.stream_ptr(x.device))
        N.check(rc, "ebfi_census_backward")
        return gx, None


class _LapLoss(torch.autograd.Function):
    """coef_a * Lap(a, target) + coef_b * Lap(b, target) (restore.py:166-213) as ONE pyramid of the difference planes
    [a - target ; b - target] on csrc/laploss.hip: the pyramid is linear, so lap_i(x) - lap_i(y) = lap_i(x - y)."""

    LEVELS = 5

    @staticmethod
    def usable(a, b, target):
        H, W = a.shape[-2:]
        m = 1 << (_LapLoss.LEVELS - 1)
        ok = lambda t: t.is_cuda and t.dtype == torch.float32 and t.shape == a.shape
        return a.dim() == 4 and ok(a) and ok(target) and (b is None or ok(b)) and H % m == 0 and W % m == 0 and \
            2 * H // m >= 3 and 2 * W // m >= 3 and not target.requires_grad and N.dev_env("EBFI_NO_LAPLOSS") is None

    @staticmethod
    def forward(ctx, a, b, target, coef_a, coef_b):
        a, target = a.contiguous(), target.contiguous()
        b = b.contiguous() if b is not None else None
        B, C, H, W = a.shape
        ppt, L = B * C, _LapLoss.LEVELS
        planes = ppt * (2 if b is not None else 1)
        lib = N.lib()
        ws = torch.empty(int(lib.ebfi_laploss_workspace_floats(planes, H, W, L)), dtype=torch.float32, device=a.device)
        partial = torch.empty(int(lib.ebfi_laploss_partials(planes, H, W, L)), dtype=torch.float32, device=a.device)
        with torch.cuda.device_of(a):
            rc = lib.ebfi_laploss_forward(N.ptr(a), N.ptr(b) if b is not None else None, N.ptr(target), float(coef_a),
                                          float(coef_b), N.ptr(ws), N.ptr(partial), ppt, H, W, L, N.stream_ptr(a.device))
        N.check(rc, "ebfi_laploss_forward")
        ctx.ws, ctx.shape, ctx.two = ws, (B, C, H, W), b is not None
        return partial.sum()

    @staticmethod
    def backward(ctx, g):
        if ctx.ws is None:
            raise RuntimeError("laploss: the workspace of this forward was already consumed by a backward pass")
        B, C, H, W = ctx.shape
        n = 2 if ctx.two else 1
        g = g.contiguous().float().reshape(1)
        out = torch.empty((n * B, C, H, W), dtype=torch.float32, device=g.device)
        with torch.cuda.device_of(out):
            rc = N.lib().ebfi_laploss_backward(N.ptr(g), N.ptr(ctx.ws), N.ptr(out), n * B * C, H, W, _LapLoss.LEVELS,
                                               N.stream_ptr(out.device))
        N.check(rc, "ebfi_laploss_backward")
        ctx.ws = None
        return out[:B], (out[B:] if ctx.two else None), None, None, None


class GaussianConv(nn.Module):
    """5x5 binomial blur with reflect padding (restore.py:149-163).  The kernel is separable
    ([1,4,6,4,1]/16 twice), so it is applied as shifted-slice sums: no library convolution involved."""

    def __init__(self):
        super().__init__()
        k1 = torch.tensor([1., 4., 6., 4., 1.])
        self.kernel = nn.Parameter((k1[:, None] * k1[None, :] / 256).repeat(3, 1, 1, 1), requires_grad=False)
        self.taps = (1.0 / 16, 4.0 / 16, 6.0 / 16, 4.0 / 16, 1.0 / 16)

    def forward(self, x, factor=1):
        H, W = x.shape[-2:]
        if x.is_cuda and x.dtype == torch.float32 and H >= 3 and W >= 3:
            return _Gauss5.apply(x, factor)
        p = F.pad(x, (2, 2, 2, 2), mode="reflect")     # CPU tensors (host-logic tests): shifted-slice sums
        h = sum(t * p[..., :, j:j + W] for j, t in enumerate(self.taps))
        v = sum(t * h[..., i:i + H, :] for i, t in enumerate(self.taps))
        return v * factor if factor != 1 else v


class LaplacianPyramid(nn.Module):
    def __init__(self, max_level=5):
        super().__init__()
        self.gaussian_conv = GaussianConv()
        self.max_level = max_level

    def expand(self, x):
        B, C, H, W = x.shape
        up = x.new_zeros(B, C, 2 * H, 2 * W)
        up[:, :, ::2, ::2] = x                      # zero insertion (restore.py:189-197)
        return self.gaussian_conv(up, factor=4)

    def forward(self, X):
        pyramid, cur = [], X
        for _ in range(self.max_level - 1):
            reduced = F.avg_pool2d(self.gaussian_conv(cur), 2)
            pyramid.append(cur - self.expand(reduced))
            cur = reduced
        pyramid.append(cur)
        return pyramid


class LaplacianLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.lap = LaplacianPyramid()

    def forward(self, x, y, y_pyramid=None):
        """y_pyramid: optional precomputed pyramid of the target (it is shared by both loss terms of a step)."""
        yp = y_pyramid if y_pyramid is not None else self.lap(y)
        return sum((2 ** i) * F.l1_loss(a, b, reduction="sum") for i, (a, b) in enumerate(zip(self.lap(x), yp)))


class Ternary(nn.Module):
    def __init__(self, patch_size=7):
        super().__init__()
        self.patch_size = patch_size
        n = patch_size * patch_size
        self.register_buffer("w", torch.eye(n).view(n, 1, patch_size, patch_size), persistent=False)

    def transform(self, t):
        g = t.mean(dim=1, keepdim=True)
        H, W = g.shape[-2:]
        k, r = self.patch_size, self.patch_size // 2
        gp = F.pad(g, (r, r, r, r))                  # zero padding, like conv2d(padding=r)
        patches = torch.cat([gp[..., i:i + H, j:j + W] for i in range(k) for j in range(k)], dim=1)
        d = patches - g
        return d / torch.sqrt(0.81 + d ** 2)

    def forward(self, x, y, y_transform=None):
        if x.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32 and self.patch_size == 7 and \
                not y.requires_grad and x.shape[-1] > 6 and x.shape[-2] > 6:
            return _Census.apply(x, y)
        ty = y_transform if y_transform is not None else self.transform(y).detach()
        diff = self.transform(x) - ty
        dist = (diff ** 2 / (0.1 + diff ** 2)).mean(dim=1, keepdim=True)
        p = self.patch_size // 2
        mask = torch.zeros_like(dist)
        mask[:, :, p:-p, p:-p] = 1
        return (dist * mask).mean()


class TrainLoss(nn.Module):
    """train_ours.py:258-268; the model returns (SharpPre, Sharp) = its (Sharp, Final)."""

    def __init__(self, detail_enabled=True):
        super().__init__()
        self.Lap, self.census, self.detail_enabled = LaplacianLoss(), Ternary(), detail_enabled

    def forward(self, sharp_pre, sharp, target, iteration=0, accu_step=1):
        c_sharp, c_pre = (0.1, 1.0) if iteration < 10e3 else (1.0, 0.1)
        if not self.detail_enabled:
            c_sharp, c_pre, sharp_pre = 1.0, 0.0, None
        if _LapLoss.usable(sharp, sharp_pre, target):
            # one difference pyramid for both Laplacian terms; the census kernels work on the images themselves
            total = _LapLoss.apply(sharp, sharp_pre, target, c_sharp, c_pre) + c_sharp * self.census(sharp, target)
            if sharp_pre is not None:
                total = total + c_pre * self.census(sharp_pre, target)
            return total / accu_step
        with torch.no_grad():     # the target side of both terms is the same: compute it once
            yp = self.Lap.lap(target)
            ty = None if target.is_cuda else self.census.transform(target)   # the GPU census kernel works on the images
        term = lambda p: self.Lap(p, target, yp) + self.census(p, target, ty)
        if sharp_pre is None:
            return term(sharp) / accu_step
        return (c_sharp * term(sharp) + c_pre * term(sharp_pre)) / accu_step
