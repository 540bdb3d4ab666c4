"""Training loss of the hot path (device side).

Reference: loss/restore.py:149-213 (LaplacianLoss: 5-level Laplacian pyramid, L1 *sum*, level
weight 2**i), :111-145 (Ternary census, 7x7), combined as in train_ours.py:258-268.  Plain
PyTorch-ROCm ops for now -- many small depthwise convs; fusing them is section 8(f) rank 3.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class GaussianConv(nn.Module):
    def __init__(self):
        super().__init__()
        k1 = torch.tensor([1., 4., 6., 4., 1.])
        self.kernel = nn.Parameter((k1[:, None] * k1[None, :] / 256).repeat(3, 1, 1, 1), requires_grad=False)

    def forward(self, x, factor=1):
        c = x.shape[1]
        return F.conv2d(F.pad(x, (2, 2, 2, 2), mode="reflect"), factor * self.kernel[:c], groups=c)


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

    def forward(self, x, y):
        return sum((2 ** i) * F.l1_loss(a, b, reduction="sum")
                   for i, (a, b) in enumerate(zip(self.lap(x), self.lap(y))))


class Ternary(nn.Module):
    def __init__(self, patch_size=7):
        super().__init__()
        self.patch_size = patch_size
        n = patch_size * patch_size
        self.register_buffer("w", torch.eye(n).view(n, 1, patch_size, patch_size), persistent=False)

    def transform(self, t):
        g = t.mean(dim=1, keepdim=True)
        d = F.conv2d(g, self.w.to(g.dtype), padding=self.patch_size // 2) - g
        return d / torch.sqrt(0.81 + d ** 2)

    def forward(self, x, y):
        diff = self.transform(x) - self.transform(y).detach()
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
        term = lambda p: self.Lap(p, target) + self.census(p, target)
        if not self.detail_enabled:
            return term(sharp) / accu_step
        if iteration < 10e3:
            return (0.1 * term(sharp) + term(sharp_pre)) / accu_step
        return (term(sharp) + 0.1 * term(sharp_pre)) / accu_step
