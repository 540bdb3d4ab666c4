"""Pins the FAC oracle (oracle/fac_ref*.c).

The reference has no runnable test or golden vector for FAC; what it has is the gradcheck
protocol of models/FAC/kernelconv2d/KernelConv2D.py:61-74 (K in {1,3}, H,W in {8,10}, C = 1..10,
eps = 1e-1 because the op is bilinear).  That protocol is re-run here against the oracle, and the
oracle is cross-checked with an independent unfold formulation of
KernelConv2D_kernel.cu:25-53.
"""
import random

import pytest
import torch
import torch.nn.functional as F

from oracle import ref_ops


def fac_unfold(inp_pad, kern, K):
    B, C, Hi, Wi = inp_pad.shape
    Ho, Wo = Hi - K + 1, Wi - K + 1
    patches = F.unfold(inp_pad, K).view(B, C, K * K, Ho, Wo)
    return (patches * kern.view(B, C, K * K, Ho, Wo)).sum(2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("K", [1, 3, 5])
def test_forward_matches_unfold(K, dtype):
    torch.manual_seed(K)
    B, C, Ho, Wo = 2, 3, 7, 9
    x = torch.randn(B, C, Ho + K - 1, Wo + K - 1, dtype=dtype)
    k = torch.randn(B, C * K * K, Ho, Wo, dtype=dtype)
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    assert torch.allclose(ref_ops.fac_forward(x, k, K), fac_unfold(x, k, K), atol=tol, rtol=tol)


def test_backward_matches_autograd_of_unfold():
    torch.manual_seed(0)
    K, B, C, Ho, Wo = 5, 2, 4, 6, 8
    x = torch.randn(B, C, Ho + K - 1, Wo + K - 1, dtype=torch.float64, requires_grad=True)
    k = torch.randn(B, C * K * K, Ho, Wo, dtype=torch.float64, requires_grad=True)
    g = torch.randn(B, C, Ho, Wo, dtype=torch.float64)
    gx_ref, gk_ref = torch.autograd.grad(fac_unfold(x, k, K), (x, k), g)
    gx, gk = ref_ops.fac_backward(x.detach(), k.detach(), K, g)
    assert torch.allclose(gx, gx_ref, atol=1e-12) and torch.allclose(gk, gk_ref, atol=1e-12)


def test_strided_inputs():
    torch.manual_seed(1)
    K, B, C, Ho, Wo = 3, 2, 2, 5, 6
    x = torch.randn(B, Ho + K - 1, Wo + K - 1, C).permute(0, 3, 1, 2)       # NHWC storage
    k = torch.randn(B, Ho, Wo, C * K * K).permute(0, 3, 1, 2)
    assert not x.is_contiguous()
    assert torch.allclose(ref_ops.fac_forward(x, k, K),
                          ref_ops.fac_forward(x.contiguous(), k.contiguous(), K))


def test_reference_gradcheck_protocol():
    """KernelConv2D.py:61-74, float64 instead of CUDA float32 so the check is meaningful."""
    random.seed(0)
    torch.manual_seed(0)
    for i in range(10):
        B, C = random.randint(1, 4), i + 1
        K = random.choice([1, 3])
        H, W = random.choice([8, 10]), random.choice([8, 10])
        x = torch.randn(B, C, H + K - 1, W + K - 1, dtype=torch.float64, requires_grad=True)
        k = torch.randn(B, C * K * K, H, W, dtype=torch.float64, requires_grad=True)
        assert torch.autograd.gradcheck(lambda a, b: ref_ops.FacRefFunction.apply(a, b, K),
                                        (x, k), eps=1e-1, atol=1e-5, rtol=1e-3)


def test_module_replicate_pad():
    torch.manual_seed(2)
    x = torch.randn(1, 2, 4, 5)
    k = torch.zeros(1, 2 * 9, 4, 5)
    k[:, 0::9] = 1.0   # tap (0,0) only -> output = input shifted by (-1,-1) with edge replication
    out = ref_ops.fac_module(x, k, 3)
    xp = F.pad(x, (1, 1, 1, 1), mode="replicate")
    assert torch.equal(out, xp[:, :, 0:4, 0:5])
