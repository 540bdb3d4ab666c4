"""ORACLE (test infrastructure only): functional restatement of the training loss.

  LaplacianLoss  loss/restore.py:149-213  (5-level Laplacian pyramid, L1 *sum*, level weight 2**i)
  Ternary        loss/restore.py:111-145  (7x7 census transform, soft Hamming distance, mean)
  combination    train_ours.py:258-268    (Lap + census on Sharp and SharpPre, 0.1 weighting that
                                           flips after 10k iterations, / accu_step)

Pinned against the reference classes (tests/golden/loss_*.npz via make_golden.py).
"""
import torch
import torch.nn.functional as F

_G5 = torch.tensor([[1., 4., 6., 4., 1.],
                    [4., 16., 24., 16., 4.],
                    [6., 24., 36., 24., 6.],
                    [4., 16., 24., 16., 4.],
                    [1., 4., 6., 4., 1.]]) / 256


def _gauss(x, factor=1.0):
    c = x.shape[1]
    k = (factor * _G5).to(x.dtype).repeat(c, 1, 1, 1)
    return F.conv2d(F.pad(x, (2, 2, 2, 2), mode="reflect"), k, groups=c)


def _expand(x):
    """Zero-insertion x2 upsampling followed by 4 x Gaussian (restore.py:189-199)."""
    B, C, H, W = x.shape
    up = torch.zeros(B, C, 2 * H, 2 * W, dtype=x.dtype)
    up[:, :, ::2, ::2] = x
    return _gauss(up, 4.0)


def laplacian_pyramid(x, levels=5):
    pyr, cur = [], x
    for _ in range(levels - 1):
        red = F.avg_pool2d(_gauss(cur), 2)
        pyr.append(cur - _expand(red))
        cur = red
    pyr.append(cur)
    return pyr


def laplacian_loss(x, y):
    return sum((2 ** i) * (a - b).abs().sum()
               for i, (a, b) in enumerate(zip(laplacian_pyramid(x), laplacian_pyramid(y))))


def _census(t, patch=7):
    g = t.mean(dim=1, keepdim=True)
    n = patch * patch
    w = torch.eye(n, dtype=t.dtype).view(n, 1, patch, patch)
    d = F.conv2d(g, w, padding=patch // 2) - g
    return d / torch.sqrt(0.81 + d ** 2)


def census_loss(x, y, patch=7):
    diff = _census(x, patch) - _census(y, patch).detach()
    dist = (diff ** 2 / (0.1 + diff ** 2)).mean(dim=1, keepdim=True)
    p = patch // 2
    mask = torch.zeros_like(dist)
    mask[:, :, p:-p, p:-p] = 1
    return (dist * mask).mean()


def train_loss(sharp_pre, sharp, target, iteration=0, accu_step=1, detail_enabled=True):
    """train_ours.py:258-268.  (model returns (SharpPre, Sharp) = (Sharp, Final).)"""
    term = lambda p: laplacian_loss(p, target) + census_loss(p, target)
    if not detail_enabled:
        return term(sharp) / accu_step
    if iteration < 10e3:
        return (0.1 * term(sharp) + term(sharp_pre)) / accu_step
    return (term(sharp) + 0.1 * term(sharp_pre)) / accu_step
