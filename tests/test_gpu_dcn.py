"""GPU parity: DCNv2 HIP kernels (through the C ABI) vs the CPU oracle; the reference's own
known-answer tests (models/DCNv2/testcpu.py:32-67, :69-97, :169-180) re-run on the GPU path."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_ops  # noqa: E402


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _inputs(B, C, H, W, Co, k, s, p, d, dg, seed=0, off_scale=2.0):
    kh, kw = (k, k) if isinstance(k, int) else k
    ph, pw = (p, p) if isinstance(p, int) else p
    torch.manual_seed(seed)
    Ho, Wo = ref_ops.dcn_out_hw(H, W, kh, kw, s, s, ph, pw, d, d)
    x = torch.randn(B, C, H, W)
    off = torch.randn(B, dg * 2 * kh * kw, Ho, Wo) * off_scale
    msk = torch.sigmoid(torch.randn(B, dg * kh * kw, Ho, Wo))
    w = torch.randn(Co, C, kh, kw) * (1.0 / (C * kh * kw) ** 0.5)
    b = torch.randn(Co)
    g = torch.randn(B, Co, Ho, Wo)
    return x, off, msk, w, b, g


CASES = [
    dict(B=2, C=2, H=4, W=4, Co=2, k=3, s=1, p=1, d=1, dg=1),        # the reference's test size
    dict(B=2, C=16, H=13, W=17, Co=8, k=3, s=1, p=1, d=1, dg=2),     # ragged pixel tile, Co < 32
    dict(B=1, C=64, H=16, W=24, Co=64, k=3, s=1, p=1, d=1, dg=8),    # the model-shaped config
    dict(B=1, C=12, H=15, W=11, Co=70, k=3, s=2, p=1, d=1, dg=3),    # stride 2, Co > 64 (two co blocks)
    dict(B=2, C=4, H=9, W=9, Co=5, k=3, s=1, p=2, d=2, dg=1),        # dilation 2
    dict(B=1, C=8, H=8, W=8, Co=4, k=1, s=1, p=0, d=1, dg=4),        # 1x1
    dict(B=1, C=20, H=10, W=12, Co=6, k=3, s=1, p=1, d=1, dg=1),     # cpg=20 > 8 per chunk: sub-blocks
    dict(B=1, C=4, H=12, W=10, Co=3, k=(3, 5), s=1, p=(1, 2), d=1, dg=2),  # kh != kw, ph != pw (pad quirk)
    dict(B=1, C=2, H=9, W=9, Co=2, k=7, s=1, p=3, d=1, dg=1),        # 49 taps: one channel per chunk
    # the LDS-window forward (3x3, stride 1, pad 1, 8 channels per group): ragged 16 x 8 pixel tiles in both directions, offsets
    # wide enough (sigma 4 against a window reach of 6) that many samples take the global-gather path next to the window path
    dict(B=2, C=16, H=19, W=37, Co=40, k=3, s=1, p=1, d=1, dg=2, off_scale=4.0),
    dict(B=1, C=8, H=7, W=5, Co=3, k=3, s=1, p=1, d=1, dg=1, off_scale=0.5),   # an image smaller than one tile and its window
]


@pytest.mark.parametrize("cfg", CASES)
def test_forward_backward_vs_oracle(cfg):
    from ebfi_amd.dcn import dcn_v2_conv
    x, off, msk, w, b, g = _inputs(**cfg, seed=3)
    s, p, d, dg = cfg["s"], cfg["p"], cfg["d"], cfg["dg"]
    ref = ref_ops.dcn_forward(x, w, b, off, msk, s, p, d, dg)
    grads_ref = ref_ops.dcn_backward(x, w, b, off, msk, g, s, p, d, dg)
    t = [v.cuda().requires_grad_() for v in (x, off, msk, w, b)]
    out = dcn_v2_conv(t[0], t[1], t[2], t[3], t[4], s, p, d, dg)
    assert out.shape == ref.shape
    assert _rel(out.detach().cpu(), ref) < 2e-5
    out.backward(g.cuda())
    for v, r, name in zip(t, grads_ref, ["input", "offset", "mask", "weight", "bias"]):
        assert _rel(v.grad.cpu(), r) < 5e-5, name


def test_zero_offset_identity():
    """testcpu.py:32-67: identity weight + zero offsets + mask 0.5 => 2*out == in (tol 1e-10)."""
    from ebfi_amd.dcn import DCNv2
    torch.manual_seed(0)
    N, inC, inH, inW, outC, kH, kW = 2, 2, 4, 4, 2, 3, 3
    m = DCNv2(inC, outC, (kH, kW), stride=1, padding=1, dilation=1, deformable_groups=1).cuda()
    m.weight.data.zero_()
    m.bias.data.zero_()
    for q in range(outC):
        m.weight.data[q, q, kH // 2, kW // 2] = 1.0
    x = torch.randn(N, inC, inH, inW).cuda()
    offset = torch.zeros(N, 2 * kH * kW, inH, inW).cuda()
    mask = torch.sigmoid(torch.zeros(N, kH * kW, inH, inW)).cuda()
    out = m(x, offset, mask) * 2
    assert (x - out).abs().max() < 1e-10


def test_example_dconv_shapes_and_dcn_sep():
    """testcpu.py:169-180 (DCN(64,64,3,dg=2) on [2,64,128,128] + backward) and DCN_sep."""
    from ebfi_amd.dcn import DCN, DCN_sep
    torch.manual_seed(1)
    x = torch.randn(2, 64, 128, 128).cuda()
    dcn = DCN(64, 64, kernel_size=(3, 3), stride=1, padding=1, deformable_groups=2).cuda()
    out = dcn(x)
    target = torch.empty_like(out).uniform_(-0.01, 0.01)
    (target - out).mean().backward()
    assert out.shape == (2, 64, 128, 128) and torch.isfinite(dcn.weight.grad).all()
    # zero-initialised offset conv => plain conv with mask 0.5
    ref = 0.5 * torch.nn.functional.conv2d(x.cpu(), dcn.weight.detach().cpu(), None, 1, 1) + \
        dcn.bias.detach().cpu().view(1, -1, 1, 1)
    assert _rel(out.detach().cpu(), ref) < 2e-5
    sep = DCN_sep(64, 64, 3, 1, 1, deformable_groups=8).cuda()
    torch.nn.init.normal_(sep.conv_offset_mask.weight, std=0.02)
    y = sep(x, torch.randn_like(x))
    assert y.shape == (2, 64, 128, 128)


def test_gradients_vs_fp64_oracle_reference_protocol():
    """testcpu.py:69-97 sizes; the GPU op is fp32-only so instead of finite differences its
    analytic grads are compared with the float64 oracle (which passes that gradcheck)."""
    from ebfi_amd.dcn import dcn_v2_conv
    torch.manual_seed(2)
    N, inC, inH, inW, outC, kH, kW, dg = 2, 2, 4, 4, 2, 3, 3, 1
    x = torch.rand(N, inC, inH, inW) * 0.01
    off = torch.randn(N, dg * 2 * kW * kH, inH, inW) * 2
    frac = off - off.floor()
    off = torch.where((frac < 0.01) | (frac > 0.99), off + 0.025, off)
    msk = torch.sigmoid(torch.rand(N, dg * kW * kH, inH, inW))
    w, b = torch.randn(outC, inC, kH, kW), torch.rand(outC)
    g = torch.randn(N, outC, inH, inW)
    ref = ref_ops.dcn_backward(*(v.double() for v in (x, w, b, off, msk, g)), 1, 1, 1, dg)
    t = [v.cuda().requires_grad_() for v in (x, off, msk, w, b)]
    dcn_v2_conv(*t, 1, 1, 1, dg).backward(g.cuda())
    for v, r, name in zip(t, ref, ["input", "offset", "mask", "weight", "bias"]):
        assert _rel(v.grad.double().cpu(), r) < 1e-4, name


def test_border_rules_exact():
    """Samples exactly at -1 / H are excluded, half-outside samples get half weight."""
    from ebfi_amd.dcn import dcn_v2_conv
    x = (torch.arange(16, dtype=torch.float32).view(1, 1, 4, 4) + 1).cuda()
    w, b = torch.ones(1, 1, 1, 1).cuda(), torch.zeros(1).cuda()
    m = torch.ones(1, 1, 4, 4).cuda()
    off = torch.zeros(1, 2, 4, 4).cuda()
    off[0, 0] = -1.0
    out = dcn_v2_conv(x, off, m, w, b, 1, 0, 1, 1)
    assert torch.equal(out[0, 0, 0], torch.zeros(4).cuda()) and torch.equal(out[0, 0, 1:], x[0, 0, :3])
    off[0, 0] = 0.5
    out = dcn_v2_conv(x, off, m, w, b, 1, 0, 1, 1)
    assert torch.allclose(out[0, 0, 3], 0.5 * x[0, 0, 3])


def test_full_size_one_sample_vs_oracle_and_batch_consistency():
    """BASELINE op size (B=8, 64->64, 128x128, dg=8): sample 3 element-wise vs the oracle; the
    batch result equals per-sample results (samples are independent); grad_weight is the sum."""
    from ebfi_amd.dcn import dcn_v2_backward, dcn_v2_forward
    cfg = dict(B=8, C=64, H=128, W=128, Co=64, k=3, s=1, p=1, d=1, dg=8)
    x, off, msk, w, b, g = _inputs(**cfg, seed=123)
    xc, oc, mc, wc, bc, gc = (v.cuda() for v in (x, off, msk, w, b, g))
    out = dcn_v2_forward(xc, wc, bc, oc, mc, (1, 1), (1, 1), (1, 1), 8)
    n = 3
    ref = ref_ops.dcn_forward(x[n:n + 1], w, b, off[n:n + 1], msk[n:n + 1], 1, 1, 1, 8)
    assert _rel(out[n:n + 1].cpu(), ref) < 2e-5
    gx, go, gm, gw, gb = dcn_v2_backward(xc, wc, bc, oc, mc, gc, (1, 1), (1, 1), (1, 1), 8)
    r = ref_ops.dcn_backward(x[n:n + 1], w, b, off[n:n + 1], msk[n:n + 1], g[n:n + 1], 1, 1, 1, 8)
    assert _rel(gx[n:n + 1].cpu(), r[0]) < 5e-5
    assert _rel(go[n:n + 1].cpu(), r[1]) < 5e-5
    assert _rel(gm[n:n + 1].cpu(), r[2]) < 5e-5
    one = dcn_v2_backward(xc[n:n + 1].contiguous(), wc, bc, oc[n:n + 1].contiguous(), mc[n:n + 1].contiguous(),
                          gc[n:n + 1].contiguous(), (1, 1), (1, 1), (1, 1), 8)
    assert _rel(one[3].cpu(), r[3]) < 5e-5 and _rel(one[4].cpu(), r[4]) < 5e-5
    # grad_weight / grad_bias of the batch = sum over samples (deterministic slab reduction)
    acc_w, acc_b = torch.zeros_like(gw), torch.zeros_like(gb)
    for i in range(8):
        o = dcn_v2_backward(xc[i:i + 1].contiguous(), wc, bc, oc[i:i + 1].contiguous(), mc[i:i + 1].contiguous(),
                            gc[i:i + 1].contiguous(), (1, 1), (1, 1), (1, 1), 8)
        acc_w += o[3]
        acc_b += o[4]
    assert _rel(gw, acc_w) < 1e-4 and _rel(gb, acc_b) < 1e-4
    gw2 = dcn_v2_backward(xc, wc, bc, oc, mc, gc, (1, 1), (1, 1), (1, 1), 8)[3]
    assert torch.equal(gw, gw2)          # bit-reproducible


def test_argument_checks():
    from ebfi_amd.dcn import dcn_v2_conv
    x = torch.randn(1, 4, 6, 6).cuda()
    w, b = torch.randn(3, 4, 3, 3).cuda(), torch.zeros(3).cuda()
    with pytest.raises(RuntimeError):
        dcn_v2_conv(x, torch.zeros(1, 18, 5, 6).cuda(), torch.ones(1, 9, 6, 6).cuda(), w, b, 1, 1, 1, 1)
    with pytest.raises(RuntimeError):
        dcn_v2_conv(x, torch.zeros(1, 18, 6, 6).cuda(), torch.ones(1, 9, 6, 6).cuda(),
                    torch.randn(3, 5, 3, 3).cuda(), b, 1, 1, 1, 1)


def test_dcn_forward_split_precision_product_vs_oracle():
    """ebfi_dcn_forward with dtype EBFI_F32_BF16X3MMA (selected by the op's own `product` argument, never by the
    process-wide conv mode): the sampling is the exact
    fp32 path, the product runs as bf16 hi/lo pairs -- within 1e-4 of the oracle (the exact kernel stays the default and
    keeps the known-answer tests above)."""
    from ebfi_amd import _native as N
    from ebfi_amd import conv
    from ebfi_amd.dcn import dcn_v2_forward
    from oracle import ref_ops
    torch.manual_seed(3)
    for (B, C, H, W, Co, dg) in [(2, 64, 20, 36, 64, 8), (1, 16, 9, 13, 24, 2), (1, 64, 16, 64, 100, 8)]:
        x, w, b = torch.randn(B, C, H, W), torch.randn(Co, C, 3, 3) / (C * 9) ** 0.5, torch.randn(Co)
        off, msk = torch.randn(B, dg * 18, H, W) * 2, torch.sigmoid(torch.randn(B, dg * 9, H, W))
        ref = ref_ops.dcn_forward(x, w, b, off, msk, 1, 1, 1, dg)
        conv.set_compute_dtype("bf16x3")      # must NOT change what the DCN op runs: the product mode is its own argument
        N.prof_reset()
        N.prof_enable(True)
        try:
            plain = dcn_v2_forward(x.cuda(), w.cuda(), b.cuda(), off.cuda(), msk.cuda(), (1, 1), (1, 1), (1, 1), dg)
            torch.cuda.synchronize()
            assert "dcn_fwd_bf16x3" not in N.prof_collect()
            out = dcn_v2_forward(x.cuda(), w.cuda(), b.cuda(), off.cuda(), msk.cuda(), (1, 1), (1, 1), (1, 1), dg,
                                 product="bf16x3")
            torch.cuda.synchronize()
        finally:
            conv.set_compute_dtype("fp32")
            N.prof_enable(False)
        assert _rel(plain.cpu(), ref) < 1e-5
        assert "dcn_fwd_bf16x3" in N.prof_collect()
        assert _rel(out.cpu(), ref) < 1e-4


def test_non_finite_grad_output_propagates_through_grad_input():
    """Round-2 advisory: the fixed-point LDS box of dcn_bwd_data must not launder NaN / Inf in grad_output into finite
    numbers (__float2int_rn(NaN) == 0).  With a non-finite a-priori bound the chunk falls back to float global atomics: the
    touched cells of grad_input are then non-finite, exactly the cells the reference's atomicAdd col2im would poison."""
    from ebfi_amd.dcn import dcn_v2_backward
    x, off, msk, w, b, g = _inputs(B=1, C=8, H=16, W=16, Co=8, k=3, s=1, p=1, d=1, dg=1, seed=5, off_scale=0.5)
    g[0, 3, 7, 9] = float("nan")
    gi = dcn_v2_backward(*[v.cuda() for v in (x, w, b, off, msk, g)], (1, 1), (1, 1), (1, 1), 1)[0].cpu()
    ref = ref_ops.dcn_backward(x, w, b, off, msk, g, 1, 1, 1, 1)[0]
    bad, bad_ref = ~torch.isfinite(gi), ~torch.isfinite(ref)
    assert bad_ref.any() and torch.equal(bad, bad_ref)
    assert _rel(gi[~bad], ref[~bad_ref]) < 5e-5


def test_forward_with_non_finite_and_far_offsets_matches_the_reference_rule():
    """dcn_v2_im2col_cuda.cu:176-186: a sample contributes only if -1 < h < H and -1 < w < W -- a NaN, infinite or far-away
    sampling position compares false and contributes an exact 0, whatever the mask.  The forward kernel folds bilinear weight,
    mask and corner selection into four coefficients per (pixel, tap): they must be exactly 0 there (not NaN * 0)."""
    from ebfi_amd.dcn import dcn_v2_conv
    cfg = dict(B=1, C=16, H=12, W=20, Co=8, k=3, s=1, p=1, d=1, dg=2)
    x, off, msk, w, b, _ = _inputs(**cfg, seed=11)
    flat = off.view(-1)
    idx = torch.randperm(flat.numel(), generator=torch.Generator().manual_seed(5))[:200]
    vals = torch.tensor([float("nan"), float("inf"), -float("inf"), 1e9, -1e9, 3e38, 25.0, -25.0])
    flat[idx] = vals[torch.arange(200) % len(vals)]
    ref = ref_ops.dcn_forward(x, w, b, off, msk, 1, 1, 1, 2)
    assert torch.isfinite(ref).all()
    out = dcn_v2_conv(x.cuda(), off.cuda(), msk.cuda(), w.cuda(), b.cuda(), 1, 1, 1, 2)
    assert torch.isfinite(out).all()
    assert _rel(out.cpu(), ref) < 2e-5
