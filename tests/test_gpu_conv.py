"""GPU parity: conv + bias + activation kernels (C ABI) vs the CPU statement of the same op
(torch CPU conv2d, as used by oracle/model_ref.conv_layer), forward and all three gradients."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _ref(x, w, b, stride, pad, act, slope):
    y = F.conv2d(x, w, b, stride, pad)
    if act == 1:
        y = F.leaky_relu(y, slope)
    elif act == 2:
        y = torch.sigmoid(y)
    return y


CASES = [
    # B, Cin, H, W, Cout, k, s, p, act, bias
    (2, 64, 16, 64, 64, 3, 1, 1, 1, True),      # the ResidualControl shape, whole tiles
    (1, 128, 13, 70, 64, 3, 1, 1, 1, True),     # ragged tiles
    (2, 64, 12, 40, 200, 3, 1, 1, 1, True),     # Cout not a multiple of 64 (KernelConv-like)
    (1, 3, 32, 48, 64, 3, 2, 1, 1, True),       # FrameFeatExtract: stride 2, tiny Cin
    (1, 32, 34, 66, 64, 3, 2, 1, 1, True),      # EventFeatExtract: stride 2
    (2, 4, 24, 40, 64, 3, 1, 1, 1, True),       # BLFeatExtract: Cin=4
    (1, 64, 20, 36, 1, 3, 1, 1, 0, True),       # ExposureDecision.Conv1.1: Cout=1, no activation
    (1, 64, 17, 33, 3, 3, 1, 1, 2, True),       # last reconstruction conv: Cout=3, sigmoid
    (2, 64, 16, 32, 64, 1, 1, 0, 1, True),      # Modification.Conv1: 1x1
    (1, 20, 16, 32, 24, 3, 1, 1, 0, False),     # odd channel counts, no bias
    (1, 70, 16, 32, 64, 3, 1, 1, 1, True),      # Cin spills into a second 64-channel block
    (1, 6, 40, 72, 32, 7, 2, 3, 1, False),      # detail-branch stem (folded 3x7x7): 7x7 stride 2
    (1, 16, 38, 70, 3, 7, 1, 0, 0, True),       # detail-branch outconv: 7x7 valid conv on a reflection-padded map
    (1, 16, 11, 268, 3, 7, 1, 0, 0, True),      # ... 262 output columns: the 5-pixel-per-lane tiles of the direct kernel
    (2, 5, 9, 41, 4, 7, 1, 3, 0, False),        # direct kernel: same padding, 4 output channels, odd channel count
    (2, 12, 33, 47, 20, 3, 2, 1, 0, False),     # stride-2 data gradient through zero insertion, odd sizes
    (2, 8, 4, 5, 8, 3, 1, 1, 1, True),          # maps far smaller than one tile (deep levels of the detail branch)
    (1, 16, 8, 8, 12, 3, 1, 1, 0, False),
    (2, 8, 2, 2, 8, 3, 1, 1, 1, True),
    (1, 8, 3, 3, 4, 1, 1, 0, 1, True),
    (1, 6, 9, 7, 8, 3, 2, 1, 1, True),          # small stride-2
    (1, 6, 10, 12, 8, 7, 2, 3, 1, False),       # small 7x7 stride-2 stem
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p,act,has_bias", CASES)
def test_conv_forward_backward_vs_cpu(B, Cin, H, W, Cout, k, s, p, act, has_bias):
    from ebfi_amd.conv import conv_bias_act
    torch.manual_seed(B + Cin + H + W + Cout + k + s + act)
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout) * 0.1 if has_bias else None
    slope = 0.01
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if has_bias else None
    ref = _ref(xr, wr, br, s, p, act, slope)
    g = torch.randn_like(ref)
    ref.backward(g)
    xd, wd = x.cuda().requires_grad_(), w.cuda().requires_grad_()
    bd = b.cuda().requires_grad_() if has_bias else None
    out = conv_bias_act(xd, wd, bd, s, p, act, slope)
    assert out.shape == ref.shape
    assert _rel(out.detach(), ref.detach()) < 2e-5
    out.backward(g.cuda())
    assert _rel(xd.grad, xr.grad) < 5e-5
    assert _rel(wd.grad, wr.grad) < 5e-5
    if has_bias:
        assert _rel(bd.grad, br.grad) < 5e-5


THIN = [
    # B, Cin, H, W, Cout, k, s, p, act, bias, kernel role      (>= 64 K output pixels: below that the matrix-core kernels keep the layer)
    (1, 64, 256, 256, 3, 3, 1, 1, 2, True, "out"),       # last reconstruction conv at full size: Cout = 3, sigmoid
    (2, 32, 128, 256, 1, 3, 1, 1, 0, True, "out"),       # ExposureDecision's 64 -> 1 (one thin channel, four thick per workgroup)
    (1, 18, 256, 256, 4, 3, 1, 1, 1, False, "out"),      # 4 thin channels, thick count not a multiple of the block
    (1, 4, 256, 256, 64, 3, 1, 1, 1, True, "in"),        # ExposureDecision's blur-level conv: Cin = 4
    (1, 3, 256, 256, 17, 3, 1, 1, 1, True, "in"),        # Cin = 3, odd thick count
    (1, 1, 256, 512, 16, 3, 1, 1, 0, False, "in"),       # Cin = 1, two workgroup rows per image row
    (4, 3, 256, 256, 64, 3, 2, 1, 1, True, "ins2"),      # FrameFeatExtract: 3x3 stride 2
    (6, 4, 200, 256, 16, 3, 2, 1, 0, True, "ins2"),      # ragged last band
    # round 6 (conv2d_shift.inc.hpp): shapes only the matrix-core kernel with the taps on the thin side's row axis serves
    (1, 16, 262, 262, 3, 7, 1, 0, 0, True, "out7"),      # the detail branch's output conv on the reflection-padded map (rows not quad-aligned)
    (3, 16, 150, 230, 3, 7, 1, 0, 0, False, "out7"),     # ragged band and segment
    (3, 64, 130, 172, 3, 3, 1, 1, 2, True, "out"),       # ragged band / segment
    (2, 64, 150, 222, 1, 3, 1, 1, 0, True, "out"),       # one thin channel; rows that are not whole quads: dword loads
    (2, 4, 136, 250, 64, 3, 1, 1, 1, True, "in"),        # rows that are not whole quads: dword loads / stores of the side tensor
]


def _shift_role(Cin, Cout, k, s, p, role):
    """the role label under which csrc/conv2d_shift.inc.hpp serves the layer in the split-precision mode, or None"""
    if s != 1:
        return None
    if k == 7:
        return "out7" if (p == 0 and Cout == 3 and Cin == 16) else None
    if k == 3 and p == 1:
        if Cout in (1, 3) and Cin == 64:
            return "out"
        if Cin == 4 and Cout == 64:
            return "in"
    return None


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p,act,has_bias,role", THIN)
def test_thin_layer_weight_gradients_vs_cpu(B, Cin, H, W, Cout, k, s, p, act, has_bias, role, mode):
    """csrc/conv2d_thin.inc.hpp: 3x3 layers with <= 4 channels on one side take direct fp32 weight-gradient kernels that stream
    the thick tensor once; in the split-precision mode the shapes of the model (1 / 3 / 4 thin channels against 16..64, and the
    7x7 16 -> 3 layer) take csrc/conv2d_shift.inc.hpp instead: the matrix cores with one row per (thin channel, tap).  Against
    torch's CPU autograd at the sizes that select them; the launch is checked to BE the expected kernel, and the by-product
    grad * act'(out) feeds the data gradient as before."""
    from ebfi_amd import conv
    from ebfi_amd import _native as N
    torch.manual_seed(Cin * 7 + Cout + k + s)
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout) * 0.1 if has_bias else None
    ref = _ref(x, w, b, s, p, act, 0.01)
    g = torch.randn_like(ref)
    conv.set_compute_dtype(mode)
    try:
        xd, wd = x.cuda().requires_grad_(), w.cuda().requires_grad_()
        bd = b.cuda().requires_grad_() if has_bias else None
        N.prof_reset()
        N.prof_enable(True)
        out = conv.conv_bias_act(xd, wd, bd, s, p, act, 0.01)
        out.backward(g.cuda())
        torch.cuda.synchronize()
        N.prof_enable(False)
        prof = N.prof_collect()
    finally:
        conv.set_compute_dtype("fp32")
    shift = _shift_role(Cin, Cout, k, s, p, role) if mode == "bf16x3" else None
    if shift is not None:          # split precision: the matrix cores with the taps on the thin side's row axis
        assert "conv_wgrad_shift/" + shift in prof and not any(n.startswith("conv_wgrad_thin") for n in prof), sorted(prof)
        assert ("conv7_thin_dgrad" in prof) == (shift == "out7"), sorted(prof)      # and the 7x7 layer's data gradient in the same form
    elif k == 3 and W % 4 == 0 and 256 % (((W + 2 * p - k) // s + 1) // 4) == 0:      # (a thread = a column of quads: 256 threads are whole rows)
        assert "conv_wgrad_thin/" + role in prof, sorted(prof)
    else:                          # the exact mode outside the direct kernels' shapes: the fp32 matrix-core kernel
        assert "conv_wgrad_f32" in prof, sorted(prof)
    # the forward of the thin-OUT layers (<= 3 output channels) in the split-precision mode: taps on the matrix row axis
    assert ("conv_thin_out_fwd" in prof) == (mode == "bf16x3" and role == "out" and Cout <= 3 and 16 <= Cin <= 64 and Cin % 16 == 0), sorted(prof)
    assert _rel(out.detach(), ref) < 2e-5
    # the gradients of the op AS THE DEVICE EVALUATED IT: act' from the device's own output (a LeakyReLU output within rounding of
    # zero may carry the other sign than the CPU's: a different, equally valid subgradient -- not what this test is about)
    y = out.detach().cpu()
    gp = g * (torch.where(y > 0, 1.0, 0.01) if act == 1 else (y * (1 - y) if act == 2 else torch.ones_like(y)))
    gw_ref = torch.nn.grad.conv2d_weight(x, w.shape, gp, stride=s, padding=p)
    gx_ref = torch.nn.grad.conv2d_input(x.shape, w, gp, stride=s, padding=p)
    assert _rel(wd.grad, gw_ref) < 2e-5
    if has_bias:
        assert _rel(bd.grad, gp.sum((0, 2, 3))) < 2e-5
    assert _rel(xd.grad, gx_ref) < (5e-5 if mode == "fp32" else 2e-4)


def test_convlayer_uses_native_kernels_and_matches():
    from ebfi_amd import _native as N
    from ebfi_amd.model import ConvLayer
    torch.manual_seed(0)
    layer = ConvLayer(64, 64, 3, 1, 1, activation="LeakyReLU").cuda()
    x = torch.randn(1, 64, 32, 64).cuda()
    N.prof_reset()
    N.prof_enable(True)
    y = layer(x)
    y.sum().backward()
    torch.cuda.synchronize()
    N.prof_enable(False)
    names = set(N.prof_collect())
    assert {"conv_fwd_f32/fwd", "conv_wgrad_f32", "conv_wgrad_reduce_f32"} <= names
    ref = F.leaky_relu(F.conv2d(x.cpu(), layer.conv2d.weight.detach().cpu(), layer.conv2d.bias.detach().cpu(), 1, 1), 0.01)
    assert _rel(y.detach(), ref) < 2e-5


def test_full_size_deterministic_weight_grad():
    from ebfi_amd.conv import conv_bias_act
    torch.manual_seed(1)
    x = torch.randn(8, 64, 128, 128, device="cuda")
    w = (torch.randn(64, 64, 3, 3, device="cuda") / 24).requires_grad_()
    b = torch.zeros(64, device="cuda", requires_grad=True)
    g = torch.randn(8, 64, 128, 128, device="cuda")
    grads = []
    for _ in range(2):
        w.grad = None
        conv_bias_act(x, w, b, 1, 1, 1, 0.01).backward(g)
        grads.append(w.grad.clone())
    assert torch.equal(grads[0], grads[1])
    xs, ws = x[:1].cpu(), w.detach().cpu()
    ref = F.leaky_relu(F.conv2d(xs, ws, None, 1, 1), 0.01)
    assert _rel(conv_bias_act(x[:1].contiguous(), w.detach(), None, 1, 1, 1, 0.01), ref) < 2e-5


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,act", [(2, 64, 16, 64, 64, 3, 1), (1, 128, 13, 70, 200, 3, 1), (1, 20, 17, 33, 24, 3, 0),
                                                  (2, 64, 16, 32, 64, 1, 1), (1, 64, 17, 33, 3, 3, 2), (1, 4, 24, 40, 64, 3, 1)])
def test_bf16_mma_mode_vs_fp32_reference(B, Cin, H, W, Cout, k, act):
    """bf16 matrix-core operands (fp32 storage / accumulation): forward and data gradient within bf16
    rounding of the fp32 CPU statement (operands rounded to 8 significant bits => ~1e-2 relative)."""
    from ebfi_amd import conv
    torch.manual_seed(B + Cin + H + W + Cout + k)
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout) * 0.1
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = _ref(xr, wr, br, 1, k // 2, act, 0.01)
    g = torch.randn_like(ref)
    ref.backward(g)
    xd, wd, bd = x.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
    conv.set_compute_dtype("bf16")
    try:
        out = conv.conv_bias_act(xd, wd, bd, 1, k // 2, act, 0.01)
        out.backward(g.cuda())
    finally:
        conv.set_compute_dtype("fp32")
    assert _rel(out.detach(), ref.detach()) < 2e-2
    # The activation derivative is taken from the op's OWN (bf16-computed) output, as autograd does; a few
    # pre-activations within bf16 noise of zero flip slope w.r.t. the fp32 run, so the reference gradient is
    # evaluated with the same derivative mask: grad_in = conv^T(g * act'(out_gpu)).
    y = out.detach().cpu()
    gpre = g if act == 0 else g * (torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.01)) if act == 1 else y * (1 - y))
    gx_ref = torch.nn.grad.conv2d_input(x.shape, w, gpre, stride=1, padding=k // 2)
    assert _rel(xd.grad, gx_ref) < 2e-2
    gw_ref = torch.nn.grad.conv2d_weight(x, w.shape, gpre, stride=1, padding=k // 2)
    assert _rel(wd.grad, gw_ref) < 2e-2 and _rel(bd.grad, gpre.sum(dim=(0, 2, 3))) < 2e-2
    # exactness check: with operands that are already bf16-representable the bf16 path is exact up to fp32 accumulation
    xq, wq = x.bfloat16().float(), w.bfloat16().float()
    refq = _ref(xq, wq, b, 1, k // 2, act, 0.01)
    conv.set_compute_dtype("bf16")
    try:
        outq = conv.conv_bias_act(xq.cuda(), wq.cuda(), b.cuda(), 1, k // 2, act, 0.01)
    finally:
        conv.set_compute_dtype("fp32")
    assert _rel(outq, refq) < 2e-5


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,act", [(2, 64, 16, 64, 64, 3, 1), (1, 128, 13, 70, 200, 3, 1), (1, 20, 17, 33, 24, 3, 0),
                                                  (2, 64, 16, 32, 64, 1, 1), (1, 64, 17, 33, 3, 3, 2), (1, 4, 24, 40, 64, 3, 1),
                                                  (1, 70, 9, 130, 33, 3, 1), (2, 8, 4, 5, 8, 3, 1), (1, 16, 8, 8, 12, 3, 0),
                                                  (2, 8, 2, 2, 8, 1, 1), (1, 16, 38, 70, 3, 7, 0), (2, 20, 9, 33, 5, 7, 1),
                                                  # 7x7 with <= 16 output rows on conv7_x3: several channel chunks / 16 rows forward,
                                                  # the stem-shaped data gradient (6 <- 64 channels)
                                                  (2, 40, 21, 75, 16, 7, 0), (1, 6, 30, 40, 64, 7, 1),
                                                  # persistent forward workgroups walking SEVERAL pixel tiles each (ragged last
                                                  # round, 5 output-channel blocks) and weight-gradient workgroups of both shapes
                                                  (5, 64, 100, 130, 300, 3, 1), (3, 32, 70, 200, 64, 3, 0)])
def test_bf16x3_split_precision_mode_vs_fp32_reference(B, Cin, H, W, Cout, k, act):
    """Split-precision mode (bf16 hi + lo operand pairs, 3 MFMAs per product, fp32 accumulation) for forward and
    data and weight gradient.  Bar: 1e-4 of the fp32 CPU statement, ten times inside the 1e-3 parity tolerance of
    the path."""
    from ebfi_amd import _native as N
    from ebfi_amd import conv
    torch.manual_seed(B + Cin + H + W + Cout + k)
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout) * 0.1
    ref = _ref(x, w, b, 1, k // 2, act, 0.01)
    g = torch.randn_like(ref)
    xd, wd, bd = x.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
    conv.set_compute_dtype("bf16x3")
    N.prof_reset()
    N.prof_enable(True)
    try:
        out = conv.conv_bias_act(xd, wd, bd, 1, k // 2, act, 0.01)
        out.backward(g.cuda())
        torch.cuda.synchronize()
    finally:
        conv.set_compute_dtype("fp32")
        N.prof_enable(False)
    if k == 7:      # split precision where the product has <= 16 output rows, the exact fp32 kernels otherwise
        want = {"conv_wgrad_x3", "conv7_x3/fwd" if Cout <= 16 else "conv_fwd_f32/fwd", "conv7_x3/dgrad" if Cin <= 16 else "conv_fwd_f32/dgrad"}
    else:
        # (3x3 layers with multiples of 64 input channels: the wave-specialised weight gradient)
        want = {"conv_fwd_bf16x3_db/fwd", "conv_fwd_bf16x3_db/dgrad", "conv_wgrad_x3_ws" if k == 3 and Cin % 64 == 0 else "conv_wgrad_x3"}
    assert want <= set(N.prof_collect())
    assert _rel(out.detach(), ref) < 1e-4
    # derivative mask from the op's own output (a pre-activation within 1e-5 of zero may flip slope w.r.t. the CPU run)
    y = out.detach().cpu()
    gpre = g if act == 0 else g * (torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.01)) if act == 1 else y * (1 - y))
    assert _rel(xd.grad, torch.nn.grad.conv2d_input(x.shape, w, gpre, stride=1, padding=k // 2)) < 1e-4
    assert _rel(wd.grad, torch.nn.grad.conv2d_weight(x, w.shape, gpre, stride=1, padding=k // 2)) < 1e-4
    assert _rel(bd.grad, gpre.sum(dim=(0, 2, 3))) < 5e-5


@pytest.mark.parametrize("B,Cin,H,W,Cout,pad,act", [(2, 6, 64, 96, 64, 3, 1), (1, 3, 37, 51, 20, 3, 0), (1, 16, 24, 130, 33, 3, 1),
                                                    (2, 6, 18, 20, 8, 2, 0), (1, 1, 9, 9, 5, 0, 0)])
def test_stride2_7x7_data_gradient_by_parity_vs_cpu(B, Cin, H, W, Cout, pad, act):
    """The stem of the detail branch (7x7, stride 2, few input channels): its data gradient in split precision by output
    parity class (csrc/conv2d.hip conv7s2_dgrad_x3) instead of zero insertion; odd sizes, several channel chunks, pads 0..3."""
    from ebfi_amd import _native as N
    from ebfi_amd import conv
    torch.manual_seed(B + Cin + H + W + Cout)
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, 7, 7) / (Cin * 49) ** 0.5
    b = torch.randn(Cout) * 0.1
    ref = _ref(x, w, b, 2, pad, act, 0.01)
    g = torch.randn_like(ref)
    xd, wd, bd = x.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
    conv.set_compute_dtype("bf16x3")
    N.prof_reset()
    N.prof_enable(True)
    try:
        out = conv.conv_bias_act(xd, wd, bd, 2, pad, act, 0.01)
        out.backward(g.cuda())
        torch.cuda.synchronize()
    finally:
        conv.set_compute_dtype("fp32")
        N.prof_enable(False)
    prof = N.prof_collect()       # pad 3 (the stem): by parity; other pads: zero insertion + the stride-1 kernel
    assert prof.get("conv7_x3/dgrad_s2", (0,))[0] == (1 if pad == 3 else 0) and (pad == 3 or prof["conv7_x3/dgrad"][0] == 1)
    y = out.detach().cpu()
    gpre = g if act == 0 else g * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.01))
    assert _rel(out.detach(), ref) < 2e-5
    assert _rel(xd.grad, torch.nn.grad.conv2d_input(x.shape, w, gpre, stride=2, padding=pad)) < 1e-4
    assert _rel(wd.grad, torch.nn.grad.conv2d_weight(x, w.shape, gpre, stride=2, padding=pad)) < 2e-5


@pytest.mark.parametrize("mode", ["bf16x3", "fp32"])
def test_hd_config5_kernelconv_128_to_1600(mode):
    """BASELINE.json config 5 feature size: the 128 -> 1600 KernelConv at B=8, 360x640 (output 2.95e9 elements, 1.47 GB
    per sample: the per-sample 32-bit descriptor reach).  Forward slices against the CPU conv on samples at both ends of
    the batch, and the adjoint identities <conv(x), g> = <x, gx> = <w, gw> (+ bias) over the whole tensors tie the two
    gradients to the checked forward."""
    from ebfi_amd import conv
    torch.manual_seed(17)
    B, Cin, Cout, H, W = 8, 128, 1600, 360, 640
    x = torch.randn(B, Cin, H, W, device="cuda").requires_grad_()
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5).requires_grad_()
    b = torch.randn(Cout, device="cuda").requires_grad_()
    conv.set_compute_dtype(mode)
    try:
        y = conv.conv_bias_act(x, w, b, 1, 1, conv.ACT_NONE, 0.0)
        assert y.numel() > 2 ** 31
        g = torch.randn(B, Cout, 8, 8, device="cuda").repeat_interleave(45, 2).repeat_interleave(80, 3)   # blocky: cheap to draw
        y.backward(g)
    finally:
        conv.set_compute_dtype("fp32")
    tol = 1e-4 if mode == "bf16x3" else 5e-5
    for bi, co in ((0, 0), (7, 1550), (3, 777)):
        ref = F.conv2d(x[bi:bi + 1].detach().cpu(), w[co:co + 50].detach().cpu(), b[co:co + 50].detach().cpu(), 1, 1)
        assert _rel(y[bi:bi + 1, co:co + 50].detach(), ref) < tol, (bi, co)
    ddot = lambda p, q: sum((p[i].double() * q[i].double()).sum() for i in range(p.shape[0]))
    gsum = g.sum(dim=(0, 2, 3), dtype=torch.float64)
    lin = ddot(y.detach(), g) - (b.detach().double() * gsum).sum()      # <conv(x) without bias, g>
    assert abs((ddot(x.detach(), x.grad) - lin) / lin) < 1e-4
    assert abs((ddot(w.detach(), w.grad) - lin) / lin) < 1e-4
    assert _rel(b.grad, gsum.float().cpu()) < 1e-4


def test_packed_grouped_conv_with_epilogue_extras_vs_cpu():
    """ebfi_conv2d_packed_x3 / ebfi_conv2d_backward_weight_x3g, the building blocks of the hand-scheduled ResidualControl
    node: a GROUPED 3x3 convolution (groups = 2) on pre-packed weight images (weight bank with the grouped data-gradient
    layout), the epilogue extras out = act(conv + bias + addend) * act'(mask_y), the same convolution as a data gradient
    through the transposed images, and the grouped weight gradient -- each against torch's grouped conv2d on the CPU."""
    import ctypes
    from ebfi_amd import _native as N
    from ebfi_amd import weightbank
    torch.manual_seed(41)
    B, C, H, W = 3, 64, 20, 72                       # 2 groups of 64 -> 128 output channels; ragged 64-px tiles
    w = torch.nn.Parameter(torch.randn(2 * C, C, 3, 3) / (C * 9) ** 0.5)
    b = torch.nn.Parameter(torch.randn(2 * C) * 0.1)
    x = torch.randn(B, 2 * C, H, W)
    addend, mask_y, g = torch.randn(B, 2 * C, H, W), torch.randn(B, 2 * C, H, W), torch.randn(B, 2 * C, H, W)
    slope = 0.01
    lrelu_d = lambda y: torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope))
    ref = F.leaky_relu(F.conv2d(x, w, b, 1, 1, groups=2) + addend, slope) * lrelu_d(mask_y)
    wd, bd = torch.nn.Parameter(w.detach().cuda()), torch.nn.Parameter(b.detach().cuda())
    bank = weightbank.WeightBank([wd, bd])
    site = bank.register([wd], [bd], kind="grouped", groups=2)
    bank.refresh()
    lib = N.lib()
    st = N.stream_ptr(torch.device("cuda"))
    dev = lambda t: t.cuda().contiguous()
    xd, ad, md, gd = dev(x), dev(addend), dev(mask_y), dev(g)
    out = torch.empty(B, 2 * C, H, W, device="cuda")
    rc = lib.ebfi_conv2d_packed_x3(N.ptr(xd), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(out), B, C, H, W, 2 * C, 3, 1, 2,
                                   1, slope, N.ptr(ad), N.ptr(md), 1, slope, st)
    N.check(rc, "ebfi_conv2d_packed_x3")
    assert _rel(out, ref.detach()) < 1e-4
    # data gradient of the grouped conv = the same kernel over g with the transposed images (no bias / activation)
    gx_ref = torch.nn.grad.conv2d_input(x.shape, w.detach(), g, stride=1, padding=1, groups=2)
    gx = torch.empty_like(xd)
    rc = lib.ebfi_conv2d_packed_x3(N.ptr(gd), site.tr_ptr(), site.tr_bytes, N.ptr(None), N.ptr(gx), B, C, H, W, 2 * C, 3, 1, 2,
                                   0, 0.0, N.ptr(None), N.ptr(None), 0, 0.0, st)
    N.check(rc, "ebfi_conv2d_packed_x3 (data gradient)")
    assert _rel(gx, gx_ref) < 1e-4
    # grouped weight / bias gradient
    gw_ref = torch.nn.grad.conv2d_weight(x, w.shape, g, stride=1, padding=1, groups=2)
    need = int(lib.ebfi_conv2d_backward_weight_workspace(B, C, H, W, 2 * C, 3, 1, 1, N.EBFI_F32))
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    gw, gb = torch.empty(2 * C, C, 3, 3, device="cuda"), torch.empty(2 * C, device="cuda")
    rc = lib.ebfi_conv2d_backward_weight_x3g(N.ptr(xd), N.ptr(gd), N.ptr(gw), N.ptr(gb), B, C, H, W, 2 * C, 3, 1, 2, N.ptr(ws), need, st)
    N.check(rc, "ebfi_conv2d_backward_weight_x3g")
    assert _rel(gw, gw_ref) < 1e-4 and _rel(gb, g.sum(dim=(0, 2, 3))) < 5e-5
    # argument checks: groups that would split a workgroup's 64 output channels are refused, not mis-computed
    rc = lib.ebfi_conv2d_packed_x3(N.ptr(xd), site.fwd_ptr(), site.fwd_bytes, N.ptr(None), N.ptr(out), B, 32, H, W, 2 * C, 3, 1, 4,
                                   0, 0.0, N.ptr(None), N.ptr(None), 0, 0.0, st)
    assert rc == -1 and b"groups" in lib.ebfi_last_error()


# ------------------------------------------------------------------------------------------------ fp16 backward (round 3)
def _banked_layer(Cin, Cout, scale_w=1.0):
    """A 3x3 conv whose weight images live in a bank with a scale book attached (what Engine builds for training)."""
    from ebfi_amd import f16scale, weightbank
    w = torch.nn.Parameter((torch.randn(Cout, Cin, 3, 3) * (scale_w / (Cin * 9) ** 0.5)).cuda())
    b = torch.nn.Parameter((torch.randn(Cout) * 0.1 * scale_w).cuda())
    bank = weightbank.WeightBank([w, b])
    bank.register(w, b, "id")
    book = f16scale.ScaleBook("cuda")
    bank.attach_scale_book(book)
    bank.refresh()
    return w, b, bank, book


@pytest.mark.parametrize("B,Cin,H,W,Cout,act,xs,gs", [
    (2, 64, 16, 64, 64, 1, 1.0, 1.0),          # ResidualControl shape, whole tiles
    (1, 128, 13, 38, 64, 1, 1.0, 1.0),         # ragged tiles, two input-channel blocks; rows that do not split into 16-byte
                                               #     quads: the pair-word weight gradient and the split-precision data gradient
    (2, 64, 20, 36, 200, 0, 1.0, 1.0),         # Cout not a multiple of 64, no activation
    (2, 48, 16, 32, 32, 1, 1.0, 1.0),          # a folded detail-branch layer: partial 64-channel input block (zero-filled)
    (1, 32, 32, 32, 48, 0, 1.0, 1.0),          # ... 32 input channels: weight gradient fp16, data gradient split precision
    (1, 64, 16, 64, 128, 1, 1e-12, 1e-20),     # magnitudes far below fp16's range: what the x0.1 initialisation produces
    (1, 64, 8, 64, 64, 0, 3e4, 1e6),           # ... and far above it (no activation: at |y| ~ 1e5 the fp32 forward's last bits
                                               #     decide the sign of y ~ 0, a kink of the test, not of the kernels)
    (3, 64, 342, 160, 64, 0, 1.0, 1.0),        # 387 forward / 1290 weight-gradient tiles on 256 persistent workgroups: the
                                               #     XCD-contiguous tile permutation (xcd_tile) with a remainder (T % 8 != 0);
                                               #     no activation: among 10 M outputs a few |y| < 1e-5 flip the LeakyReLU mask
])
def test_fp16_backward_kernels_vs_cpu(B, Cin, H, W, Cout, act, xs, gs):
    """csrc/conv2d_f16.inc.hpp through the autograd node the model uses (conv.SiteConvBiasAct with a scale book active): data
    gradient and weight / bias gradient with ONE fp16 MFMA per product and power-of-two operand scales against the fp32 CPU
    conv.  fp16 operands: 2^-12 per product, a few 1e-4 of the result's scale -- at any magnitude of the operands, because the
    just-in-time calibrated scales move them into fp16's range (cases 4, 5 would be all-zero / all-inf without)."""
    from ebfi_amd import _native as N
    from ebfi_amd import conv
    # the pixel-major weight-gradient kernel serves every layer whose rows split into 16-byte quads; the pair-word kernel
    # takes the others (product rule of ebfi_conv2d_backward_weight_f16g: no development switch involved)
    force_tr = W % 4 == 0
    torch.manual_seed(B * 7 + Cin + Cout + H)
    w, b, bank, book = _banked_layer(Cin, Cout)
    x = torch.randn(B, Cin, H, W) * xs
    g = torch.randn(B, Cout, H, W) * gs
    xr, wr, br = x.clone().requires_grad_(), w.detach().cpu().clone().requires_grad_(), b.detach().cpu().clone().requires_grad_()
    yr = _ref(xr, wr, br, 1, 1, act, 0.01)
    yr.backward(g)
    xd = x.cuda().requires_grad_()
    conv.set_compute_dtype("bf16x3")
    try:
        N.prof_reset()
        N.prof_enable(True)
        with bank.active(), book.active():
            y = conv.conv_bias_act(xd, w, b, 1, 1, act, 0.01)
            y.backward(g.cuda())
            book.finish()
        torch.cuda.synchronize()
        N.prof_enable(False)
        prof = {k: v[0] for k, v in N.prof_collect().items() if v[0] > 0}
    finally:
        conv.set_compute_dtype("fp32")
    # both gradients took the fp16 kernels: the weight gradient in its pixel-major form (transposing LDS reads; with act != 0 it
    # folds act'(y) and writes grad * act' for the data gradient) or, switched, in the pair-word form
    assert prof.get("conv_wgrad_f16_tr/f32" if force_tr else "conv_wgrad_f16_ws") == 1, prof
    # the data gradient on fp16 operands (narrower ones, and rows without quads, keep the split-precision form); behind an
    # activation it reads grad * act' as the fp16 IMAGE the weight gradient wrote (Cout % 16 == 0), else the fp32 gradient
    f16_x = Cin >= 48 and W % 4 == 0
    img = f16_x and act != 0 and Cout % 16 == 0
    assert prof.get("conv_fwd_f16_ws/img_f32", 0) == (1 if img else 0) and \
        prof.get("conv_fwd_f16_ws/f32_f32", 0) == (1 if f16_x and not img else 0), prof
    assert "conv_wgrad_x3_ws" not in prof and ("conv_fwd_bf16x3_ws/dgrad" not in prof or W % 4 != 0)
    assert _rel(y.detach(), yr.detach()) < 1e-4             # (the split-precision forward: every tile written exactly once)
    assert _rel(xd.grad, xr.grad) < 1e-3 and _rel(w.grad, wr.grad) < 1e-3 and _rel(b.grad, br.grad) < 1e-5
    assert int(book.guard[0].item()) == 0                   # calibrated scales: nothing left the range
    # the maxima recorded by the kernels became the next scales: |max| * scale in [2, 4)
    gi, xi = book.index[((w.data_ptr(), "id"), "g")], book.index[((w.data_ptr(), "id"), "x")]
    gpre = g if act == 0 else g * torch.where(_ref(x, w.detach().cpu(), b.detach().cpu(), 1, 1, act, 0.01) > 0, 1.0, 0.01)
    for i, t in ((gi, gpre), (xi, x)):
        v = t.abs().max().item() * book.scale(i)
        assert 2 <= v < 4, (i, v)
        assert book.amax(i) == 0.0


def test_fp16_backward_overflow_raises_the_guard_and_adam_skips():
    """A stale scale that would push an operand past fp16's 65504 is detected by ebfi_f16_scales_finish from the recorded
    maximum: the guard flag is set, the guarded Adam launch leaves parameters and moments untouched and counts the skip;
    the scale is repaired for the next step, which goes through."""
    from ebfi_amd import conv
    from ebfi_amd.dp import FlatAdam, FlatGradBucket
    torch.manual_seed(3)
    w, b, bank, book = _banked_layer(64, 64)
    net = torch.nn.ParameterList([w, b])
    opt, bucket = FlatAdam(list(net.parameters()), lr=1e-2), FlatGradBucket(net)
    bank2 = None
    x = torch.randn(1, 64, 16, 64).cuda()
    g = torch.randn(1, 64, 16, 64).cuda()
    from ebfi_amd import weightbank
    bank2 = weightbank.WeightBank(opt.params, flat=opt.flat.data)
    w2, b2 = opt.params
    bank2.register(w2, b2, "id")
    bank2.attach_scale_book(book)
    conv.set_compute_dtype("bf16x3")
    try:
        def step(scale_g):
            bucket.zero()
            bank2.refresh()
            with bank2.active(), book.active():
                book.begin_step()
                xd = x.clone().requires_grad_()
                conv.conv_bias_act(xd, w2, b2, 1, 1, 1, 0.01).backward(g * scale_g)
                book.finish()
            before = opt.flat.detach().clone()
            opt.step(bucket.gather(), guard=book.guard)
            torch.cuda.synchronize()
            return before, int(book.guard[0].item())
        before, flag = step(1.0)                                # calibrates
        assert flag == 0 and not torch.equal(before, opt.flat.detach())
        before, flag = step(1e6)                                # 2^20 times larger than the scale expects: 3 * 1e6 > 65504
        assert flag == 1 and torch.equal(before, opt.flat.detach()) and book.skipped_steps() == 1
        assert float(opt.inner.state[opt.flat]["step"]) == 1.0  # the skipped step does not count
        before, flag = step(1e6)                                # the scale has followed: the same data now goes through
        assert flag == 0 and not torch.equal(before, opt.flat.detach()) and book.skipped_steps() == 1
        assert torch.isfinite(opt.flat).all()
    finally:
        conv.set_compute_dtype("fp32")
