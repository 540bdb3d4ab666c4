"""Pins the DCNv2 oracle (oracle/dcn_ref*.c) with the reference's own known-answer tests
(models/DCNv2/testcpu.py:32-67 zero-offset identity, :69-97 gradcheck) and with an independent
grid_sample formulation.  The reference C++ itself cannot be built in this image."""
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_ops


def dcn_grid_sample(x, offset, mask, weight, bias, stride, padding, dilation, dg):
    """Independent statement: bilinear sampling with zero padding == grid_sample(align_corners)."""
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho, Wo = ref_ops.dcn_out_hw(H, W, kh, kw, stride, stride, padding, padding, dilation, dilation)
    cpg = C // dg
    ys = torch.arange(Ho, dtype=x.dtype).view(1, Ho, 1) * stride - padding
    xs = torch.arange(Wo, dtype=x.dtype).view(1, 1, Wo) * stride - padding
    cols = []
    for g in range(dg):
        xg = x[:, g * cpg:(g + 1) * cpg]
        taps = []
        for i in range(kh):
            for j in range(kw):
                t = i * kw + j
                dy = offset[:, g * 2 * kh * kw + 2 * t]
                dx = offset[:, g * 2 * kh * kw + 2 * t + 1]
                m = mask[:, g * kh * kw + t]
                py = ys + i * dilation + dy
                px = xs + j * dilation + dx
                gy = 2 * py / max(H - 1, 1) - 1
                gx = 2 * px / max(W - 1, 1) - 1
                grid = torch.stack([gx, gy], dim=-1)
                s = F.grid_sample(xg, grid, mode="bilinear", padding_mode="zeros",
                                  align_corners=True)
                taps.append(s * m.unsqueeze(1))
        cols.append(torch.stack(taps, dim=2))            # [B,cpg,kk,Ho,Wo]
    col = torch.cat(cols, dim=1).reshape(B, C * kh * kw, Ho * Wo)
    out = torch.matmul(weight.view(Co, -1), col).view(B, Co, Ho, Wo)
    return out + bias.view(1, -1, 1, 1)


def test_zero_offset_identity():
    """models/DCNv2/testcpu.py:32-67: identity weight, zero offsets, mask 0.5 => 2*out == in."""
    torch.manual_seed(0)
    N, inC, inH, inW, outC, kH, kW = 2, 2, 4, 4, 2, 3, 3
    weight = torch.zeros(outC, inC, kH, kW)
    for p in range(inC):
        weight[p, p, kH // 2, kW // 2] = 1.0
    bias = torch.zeros(outC)
    x = torch.randn(N, inC, inH, inW)
    offset = torch.zeros(N, 2 * kH * kW, inH, inW)
    mask = torch.sigmoid(torch.zeros(N, kH * kW, inH, inW))
    out = ref_ops.dcn_forward(x, weight, bias, offset, mask, 1, 1, 1, 1) * 2
    assert (x - out).abs().max() < 1e-10


def test_zero_offset_im2col_is_unfold():
    torch.manual_seed(1)
    x = torch.randn(2, 3, 6, 7)
    off = torch.zeros(2, 18, 6, 7)
    msk = torch.ones(2, 9, 6, 7)
    col = ref_ops.dcn_im2col(x, off, msk, 3, 1, 1, 1, 1)
    assert torch.equal(col, F.unfold(x, 3, padding=1))


def _off_grid(offset, margin=0.01):
    """Bilinear sampling is not differentiable where a sample coordinate sits on a grid line;
    integer base positions mean that is where frac(offset) == 0.  Push such offsets away."""
    frac = offset - torch.floor(offset)
    near = (frac < margin) | (frac > 1 - margin)
    return torch.where(near, offset + 2.5 * margin, offset)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_reference_gradcheck(seed):
    """models/DCNv2/testcpu.py:69-97: same sizes, eps and tolerances.  Run in float64 with the
    offsets kept off the grid lines (the reference's fp32 run on randn offsets is flaky for
    exactly that reason: a finite-difference step of 1e-3 can straddle a kink)."""
    torch.manual_seed(seed)
    dt = torch.float64
    N, inC, inH, inW, outC, kH, kW, dg = 2, 2, 4, 4, 2, 3, 3, 1
    x = (torch.rand(N, inC, inH, inW, dtype=dt) * 0.01).requires_grad_()
    offset = _off_grid(torch.randn(N, dg * 2 * kW * kH, inH, inW, dtype=dt) * 2).requires_grad_()
    mask = torch.sigmoid(torch.rand(N, dg * kW * kH, inH, inW, dtype=dt)).requires_grad_()
    weight = torch.randn(outC, inC, kH, kW, dtype=dt).requires_grad_()
    bias = torch.rand(outC, dtype=dt).requires_grad_()
    assert torch.autograd.gradcheck(ref_ops.dcn_v2_conv, (x, offset, mask, weight, bias, 1, 1, 1, dg),
                                    eps=1e-3, atol=1e-4, rtol=1e-2)


@pytest.mark.parametrize("cfg", [
    dict(B=2, C=4, H=7, W=9, Co=3, k=3, s=1, p=1, d=1, dg=2),
    dict(B=1, C=6, H=9, W=8, Co=4, k=3, s=2, p=1, d=1, dg=3),
    dict(B=2, C=2, H=8, W=8, Co=2, k=3, s=1, p=2, d=2, dg=1),
    dict(B=1, C=4, H=6, W=6, Co=5, k=1, s=1, p=0, d=1, dg=4),
])
def test_forward_backward_vs_grid_sample(cfg):
    torch.manual_seed(3)
    B, C, H, W, Co, k, s, p, d, dg = (cfg[n] for n in "B C H W Co k s p d dg".split())
    Ho, Wo = ref_ops.dcn_out_hw(H, W, k, k, s, s, p, p, d, d)
    dt = torch.float64
    x = torch.randn(B, C, H, W, dtype=dt, requires_grad=True)
    # fractional offsets, some far outside the image to exercise the border rules
    offset = (torch.randn(B, dg * 2 * k * k, Ho, Wo, dtype=dt) * 3).requires_grad_()
    mask = torch.rand(B, dg * k * k, Ho, Wo, dtype=dt, requires_grad=True)
    weight = torch.randn(Co, C, k, k, dtype=dt, requires_grad=True)
    bias = torch.randn(Co, dtype=dt, requires_grad=True)
    ref = dcn_grid_sample(x, offset, mask, weight, bias, s, p, d, dg)
    out = ref_ops.dcn_forward(x.detach(), weight.detach(), bias.detach(), offset.detach(),
                              mask.detach(), s, p, d, dg)
    assert torch.allclose(out, ref, atol=1e-10)
    g = torch.randn_like(ref)
    grads_ref = torch.autograd.grad(ref, (x, offset, mask, weight, bias), g)
    grads = ref_ops.dcn_backward(x.detach(), weight.detach(), bias.detach(), offset.detach(),
                                 mask.detach(), g, s, p, d, dg)
    for a, b, name in zip(grads, grads_ref, ["x", "offset", "mask", "weight", "bias"]):
        assert torch.allclose(a, b, atol=1e-9), name


def test_integer_offsets_on_grid():
    """Sample positions exactly on grid points / exactly at -1 and H: the strict inequalities."""
    x = torch.arange(16, dtype=torch.float32).view(1, 1, 4, 4) + 1
    w = torch.ones(1, 1, 1, 1)
    b = torch.zeros(1)
    m = torch.ones(1, 1, 4, 4)
    off = torch.zeros(1, 2, 4, 4)
    off[0, 0] = -1.0       # dy = -1 -> row -1 for the first output row (h_im == -1: excluded)
    out = ref_ops.dcn_forward(x, w, b, off, m, 1, 0, 1, 1)
    assert torch.equal(out[0, 0, 0], torch.zeros(4))
    assert torch.equal(out[0, 0, 1:], x[0, 0, :3])
    off[0, 0] = 1.0        # row H for the last output row: excluded (h_im < H fails)
    out = ref_ops.dcn_forward(x, w, b, off, m, 1, 0, 1, 1)
    assert torch.equal(out[0, 0, 3], torch.zeros(4))
    off[0, 0] = 0.5        # halfway below the last row: lower corner is outside -> half weight
    out = ref_ops.dcn_forward(x, w, b, off, m, 1, 0, 1, 1)
    assert torch.allclose(out[0, 0, 3], 0.5 * x[0, 0, 3])
