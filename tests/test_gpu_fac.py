"""GPU parity: FAC HIP kernels (through the C ABI) vs the CPU oracle on identical seeded inputs.
fp32 tolerance 1e-5 relative (target in BASELINE.json: 1e-3); plus size-independent properties at
the full BASELINE size (B=8, C=64, 128x128, K=5)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ref_ops  # noqa: E402
from ebfi_amd import _native as N  # noqa: E402


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


CASES = [
    # B, C, Ho, Wo, K
    (2, 3, 7, 12, 5),      # tiled path, narrow
    (1, 2, 16, 128, 5),    # exactly one 8x128 tile row group
    (2, 2, 9, 132, 5),     # two x-tiles, ragged edges
    (1, 3, 5, 8, 3),       # K=3: unaligned padded rows
    (2, 4, 8, 16, 1),      # K=1
    (1, 2, 6, 10, 5),      # Wo % 4 != 0 -> generic kernels
    (1, 1, 4, 8, 7),       # K=7 -> generic kernels
    (1, 5, 33, 260, 5),    # backward: 2 chunks of 64 lanes + tail lane
]


@pytest.mark.parametrize("B,C,Ho,Wo,K", CASES)
def test_forward_backward_vs_oracle(B, C, Ho, Wo, K):
    from ebfi_amd.fac import KernelConv2DFunction
    torch.manual_seed(B * 1000 + C * 100 + Ho + Wo + K)
    x = torch.randn(B, C, Ho + K - 1, Wo + K - 1)
    k = torch.randn(B, C * K * K, Ho, Wo)
    g = torch.randn(B, C, Ho, Wo)
    ref = ref_ops.fac_forward(x, k, K)
    gx_ref, gk_ref = ref_ops.fac_backward(x, k, K, g)
    xd, kd = x.cuda().requires_grad_(), k.cuda().requires_grad_()
    out = KernelConv2DFunction.apply(xd, kd, K)
    out.backward(g.cuda())
    assert _rel(out.detach().cpu(), ref) < 1e-5
    assert _rel(xd.grad.cpu(), gx_ref) < 1e-5
    assert _rel(kd.grad.cpu(), gk_ref) < 1e-5


def test_strided_layouts_any_stride():
    """The reference kernels take explicit strides (KernelConv2D_kernel.cu:14-17); so does the C ABI."""
    from ebfi_amd.fac import fac_backward, fac_forward
    torch.manual_seed(5)
    B, C, Ho, Wo, K = 2, 3, 6, 8, 5
    x = torch.randn(B, Ho + K - 1, Wo + K - 1, C).permute(0, 3, 1, 2)      # channels-last storage
    k = torch.randn(B, Ho, Wo, C * K * K).permute(0, 3, 1, 2)
    g = torch.randn(B, C, Ho, Wo + 3)[..., :Wo]                           # padded rows
    ref = ref_ops.fac_forward(x.contiguous(), k.contiguous(), K)
    out = fac_forward(x.cuda(), k.cuda(), K)
    assert _rel(out.cpu(), ref) < 1e-5
    gx_ref, gk_ref = ref_ops.fac_backward(x.contiguous(), k.contiguous(), K, g.contiguous())
    gx, gk = fac_backward(x.cuda(), k.cuda(), K, g.cuda())
    assert _rel(gx.cpu(), gx_ref) < 1e-5 and _rel(gk.cpu(), gk_ref) < 1e-5


def test_module_matches_oracle_module_and_skips_unneeded_grads():
    from ebfi_amd.fac import KernelConv2D
    torch.manual_seed(11)
    x = torch.randn(2, 4, 12, 16)
    k = torch.randn(2, 100, 12, 16)
    ref = ref_ops.fac_module(x, k, 5)
    m = KernelConv2D(5).cuda()
    xd = x.cuda().requires_grad_()
    out = m(xd, k.cuda())                 # kernel does not require grad
    assert _rel(out.detach().cpu(), ref) < 1e-5
    out.sum().backward()
    assert xd.grad is not None and torch.isfinite(xd.grad).all()


def test_reference_asserts():
    from ebfi_amd.fac import KernelConv2DFunction
    x = torch.randn(1, 2, 8, 8).cuda()
    with pytest.raises(AssertionError):
        KernelConv2DFunction.apply(x, torch.randn(1, 18, 5, 6).cuda(), 3)          # H mismatch
    with pytest.raises(AssertionError):
        KernelConv2DFunction.apply(x.transpose(2, 3), torch.randn(1, 18, 6, 6).cuda(), 3)  # not contiguous


def test_full_size_properties():
    """BASELINE size: adjoint identities <FAC(x,k),g> = <x,gx> = <k,gk>, linearity in each argument,
    and one (b, c-slice) checked element-wise against the oracle."""
    from ebfi_amd.fac import fac_backward, fac_forward
    torch.manual_seed(123)
    B, C, H, W, K = 8, 64, 128, 128, 5
    x = torch.randn(B, C, H + 4, W + 4, device="cuda")
    k = torch.randn(B, C * 25, H, W, device="cuda")
    g = torch.randn(B, C, H, W, device="cuda")
    out = fac_forward(x, k, K)
    gx, gk = fac_backward(x, k, K, g)
    dot = (out.double() * g.double()).sum()
    assert abs(((x.double() * gx.double()).sum() - dot) / dot) < 1e-5
    assert abs(((k.double() * gk.double()).sum() - dot) / dot) < 1e-5
    out2 = fac_forward(2 * x, k, K)
    assert _rel(out2, 2 * out) < 1e-6
    b, cs = 5, slice(20, 23)
    ks = k[b:b + 1, 20 * 25:23 * 25].cpu()
    ref = ref_ops.fac_forward(x[b:b + 1, cs].cpu().contiguous(), ks.contiguous(), K)
    assert _rel(out[b:b + 1, cs].cpu(), ref) < 1e-5
    gx_ref, gk_ref = ref_ops.fac_backward(x[b:b + 1, cs].cpu().contiguous(), ks.contiguous(), K,
                                          g[b:b + 1, cs].cpu().contiguous())
    assert _rel(gx[b:b + 1, cs].cpu(), gx_ref) < 1e-5
    assert _rel(gk[b:b + 1, 20 * 25:23 * 25].cpu(), gk_ref) < 1e-5


def test_hd_config5_size_64bit_indexing():
    """BASELINE.json config 5 feature size (B=8, 64 ch, 360x640, K=5): the filter tensor has 2.95e9 elements (11.8 GB),
    past the 2^31 element count where the reference's own launcher overflows (`int n_grad_kernel`,
    KernelConv2D_kernel.cu:166-167).  Adjoint identities over the whole tensors plus element-wise comparison with the
    oracle on (b, c-slices) that lie BEYOND the 2^31-element mark."""
    from ebfi_amd.fac import fac_backward, fac_forward
    torch.manual_seed(321)
    B, C, H, W, K = 8, 64, 360, 640, 5
    x = torch.randn(B, C, H + 4, W + 4, device="cuda")
    k = torch.randn(B, C * 25, H, W, device="cuda")
    g = torch.randn(B, C, H, W, device="cuda")
    assert k.numel() > 2 ** 31
    out = fac_forward(x, k, K)
    gx, gk = fac_backward(x, k, K, g)
    ddot = lambda a, b: sum((a[i].double() * b[i].double()).sum() for i in range(B))     # per sample: bounded fp64 temporaries
    dot = ddot(out, g)
    assert abs((ddot(x, gx) - dot) / dot) < 1e-5
    assert abs((ddot(k, gk) - dot) / dot) < 1e-5
    for b, c0 in ((7, 61), (6, 0), (0, 30)):               # sample 7, channel 61: element offset 2.94e9
        cs, ks = slice(c0, c0 + 2), slice(c0 * 25, (c0 + 2) * 25)
        xs, kk, gs = x[b:b + 1, cs].cpu().contiguous(), k[b:b + 1, ks].cpu().contiguous(), g[b:b + 1, cs].cpu().contiguous()
        assert _rel(out[b:b + 1, cs].cpu(), ref_ops.fac_forward(xs, kk, K)) < 1e-5
        gx_ref, gk_ref = ref_ops.fac_backward(xs, kk, K, gs)
        assert _rel(gx[b:b + 1, cs].cpu(), gx_ref) < 1e-5 and _rel(gk[b:b + 1, ks].cpu(), gk_ref) < 1e-5


def test_backward_with_filter_activation_mask():
    """ebfi_fac_backward_ex: grad_kernel as the gradient of the LeakyReLU pre-activation that produced the filters (fast row
    kernel and generic kernel), against masking the oracle's grad_kernel; grad_input unchanged."""
    from ebfi_amd.fac import fac_backward
    torch.manual_seed(17)
    for (B, C, H, W, K) in [(2, 4, 12, 16, 5), (1, 3, 9, 10, 3), (1, 2, 7, 9, 5)]:      # last two: generic path (W % 4 != 0)
        x, k, g = torch.randn(B, C, H + K - 1, W + K - 1), torch.randn(B, C * K * K, H, W), torch.randn(B, C, H, W)
        gx_ref, gk_ref = ref_ops.fac_backward(x, k, K, g)
        gx, gk = fac_backward(x.cuda(), k.cuda(), K, g.cuda(), kernel_leaky_slope=0.01)
        assert _rel(gx.cpu(), gx_ref) < 1e-5
        assert _rel(gk.cpu(), gk_ref * torch.where(k > 0, torch.ones_like(k), torch.full_like(k, 0.01))) < 1e-5


def test_empty_batch():
    from ebfi_amd.fac import fac_forward
    out = fac_forward(torch.zeros(0, 2, 8, 8).cuda(), torch.zeros(0, 18, 6, 6).cuda(), 3)
    assert out.shape == (0, 2, 6, 6)


FUSED_CASES = [
    # B, C (FAC channels), Cin, H, W
    (2, 8, 16, 20, 36),      # ragged tiles both ways, one 16-channel chunk, 4 row blocks
    (1, 3, 24, 8, 64),       # odd channel count: the last 64-row block holds one channel; Cin not a multiple of 16
    (2, 64, 128, 16, 128),   # the model's widths (128 -> 1600): 8 chunks, 32 row blocks, two x tiles
    (1, 5, 16, 9, 4),        # a feature map smaller than the 5x5 window: every tap clamps
]


@pytest.mark.parametrize("B,C,Cin,H,W", FUSED_CASES)
def test_fused_kernelconv_fac_vs_unfused_pair_and_oracle(B, C, Cin, H, W):
    """filters = LeakyReLU(conv3x3(cat)) -> FAC(feat, filters) as ONE kernel (the filter tensor never exists)
    against (a) the unfused product pair -- the same split-precision conv kernel writing the filters, then the FAC kernel --
    and (b) the CPU oracle: fp32 conv + the FAC restatement with ReplicationPad2d(2) (KernelConv2D.py:82-87)."""
    from ebfi_amd import conv, weightbank
    from ebfi_amd.fac import KernelConv2D, fac_rows_fold_bias, fac_rows_fold_weight, kernelconv_fac_fused
    torch.manual_seed(B * 1000 + C * 10 + H + W)
    K, slope = 5, 0.01
    w = torch.randn(C * K * K, Cin, 3, 3) * (1.0 / (Cin * 9) ** 0.5)
    b = torch.randn(C * K * K) * 0.1
    cat, feat = torch.randn(B, Cin, H, W), torch.randn(B, C, H, W)
    filt = F.leaky_relu(F.conv2d(cat, w, b, 1, 1), slope)
    ref = ref_ops.fac_forward(F.pad(feat, (2, 2, 2, 2), mode="replicate"), filt.contiguous(), K)
    wd, bd = torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda())
    bank = weightbank.WeightBank([wd, bd], inference=True)
    site = bank.register(wd, bd, "facrows", fac_rows_fold_weight, fac_rows_fold_bias, need_tr=False)
    bank.refresh()
    conv.set_compute_dtype("bf16x3")
    try:
        with torch.no_grad():
            out = kernelconv_fac_fused(cat.cuda(), feat.cuda(), site, K, slope)
            filt_d = conv.conv_bias_act(cat.cuda(), wd, bd, 1, 1, conv.ACT_LEAKY, slope)
            pair = KernelConv2D(K)(feat.cuda(), filt_d)
    finally:
        conv.set_compute_dtype("fp32")
    assert _rel(out.cpu(), pair.cpu()) < 2e-6          # same filter bits, another summation order over the 25 taps
    assert _rel(out.cpu(), ref) < 1e-4                 # split-precision conv vs fp32 conv (path tolerance: 1e-3)


@pytest.mark.parametrize("B,C,Cin,H,W", FUSED_CASES)
@pytest.mark.parametrize("magnitude", [1.0, 1e-12, 3e4])
def test_fused_kernelconv_fac_on_fp16_operands(B, C, Cin, H, W, magnitude):
    """Round 6: the same fused kernel with fp16 operands (conv_fwd_f16_ws<.., FAC>: one matrix-core product per tap; inference).
    The weight image carries an exact power-of-two scale from the bank, the input's scale is measured on the device before the
    launch -- so inputs at 1e-12 or 3e4 (outside fp16's range unscaled) give the same relative error as O(1) ones.  Against the
    split-precision fused kernel and the CPU oracle: operands rounded to 11 bits move a 25-tap x (9 x Cin)-term result by ~3e-4."""
    from ebfi_amd import conv, f16scale, weightbank
    from ebfi_amd.fac import fac_rows_fold_bias, fac_rows_fold_weight, kernelconv_fac_fused
    torch.manual_seed(B * 1000 + C * 10 + H + W)
    K, slope = 5, 0.01
    w = torch.randn(C * K * K, Cin, 3, 3) * (1.0 / (Cin * 9) ** 0.5)
    b = torch.randn(C * K * K) * 0.1 * magnitude
    cat, feat = torch.randn(B, Cin, H, W) * magnitude, torch.randn(B, C, H, W)
    filt = F.leaky_relu(F.conv2d(cat.double(), w.double(), b.double(), 1, 1), slope)
    ref = ref_ops.fac_forward(F.pad(feat, (2, 2, 2, 2), mode="replicate").double(), filt.contiguous(), K).float()
    wd, bd = torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda())
    outs = {}
    for tag, book in (("x3", None), ("f16", f16scale.ScaleBook("cuda", capacity=8))):
        bank = weightbank.WeightBank([wd, bd], inference=True)
        if book is not None:
            bank.attach_scale_book(book)
        site = bank.register(wd, bd, "facrows", fac_rows_fold_weight, fac_rows_fold_bias, need_tr=False, fwd16=book is not None)
        bank.refresh()
        assert (site.fwd16_ptr() is not None) == (book is not None)
        N.prof_reset()
        N.prof_enable(True)
        with torch.no_grad():
            outs[tag] = kernelconv_fac_fused(cat.cuda(), feat.cuda(), site, K, slope).cpu()
        torch.cuda.synchronize()
        N.prof_enable(False)
        ran = {k for k, v in N.prof_collect().items() if v[0] > 0}
        # (the fp16 form reads the input as a c16 image when Cin is a multiple of 16, else it converts the fp32 planes while staging)
        f16_label = "conv_fwd_f16_ws/kernelconv_fac_img" if Cin % 16 == 0 else "conv_fwd_f16_ws/kernelconv_fac"
        assert (f16_label in ran) == (book is not None) and ("conv_fwd_bf16x3_ws/kernelconv_fac" in ran) == (book is None), ran
    assert torch.isfinite(outs["f16"]).all()
    assert _rel(outs["x3"], ref) < 1e-4
    assert _rel(outs["f16"], ref) < 1e-3, _rel(outs["f16"], ref)
    assert _rel(outs["f16"], outs["x3"]) < 1e-3
    # the input handed over as the two parts of a channel concatenation (what Modification does): the same scale, the same image,
    # the same bits -- and no concatenated tensor when the image form applies
    if Cin % 16 == 0:
        c0 = Cin // 2
        N.prof_reset()
        N.prof_enable(True)
        with torch.no_grad():
            two = kernelconv_fac_fused((cat[:, :c0].contiguous().cuda(), cat[:, c0:].contiguous().cuda()), feat.cuda(), site, K, slope).cpu()
        torch.cuda.synchronize()
        N.prof_enable(False)
        assert torch.equal(two, outs["f16"])
        assert (N.prof_collect().get("to_c16/cat2", (0,))[0] == 1) == (c0 % 8 == 0)


def test_clip_interpolator_takes_the_fp16_fused_kernel():
    """ClipInterpolator (what infer_ours.py runs) in the default split-precision mode: Modification's KernelConv -> FAC pair is ONE
    launch of the fp16-operand fused kernel per timestamp (filters_f16=False: of the split-precision one); Final within 1e-3 of the
    exact fp32 mode either way."""
    from ebfi_amd.engine import DEFAULT_MODEL_ARGS, ClipInterpolator, synthetic_batch
    from ebfi_amd.model import EVFIAutoEx
    torch.manual_seed(23)
    net = EVFIAutoEx(**dict(DEFAULT_MODEL_ARGS, step=2, channels=[8, 8, 16, 16])).cuda().eval()
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() > 1:
                p.copy_(torch.randn_like(p) * (1.2 / p[0].numel() ** 0.5))
            else:
                p.add_(0.05 * torch.randn_like(p))
    frame, event, _, gtex, _ = synthetic_batch(2, 64, 96, 16, device="cuda", seed=5)
    stamps = [0.25, 0.75]
    exact = ClipInterpolator(net, precision="fp32", graph=False)(frame, event, gtex, stamps)
    for f16, graph in ((True, False), (True, True), (False, False)):
        interp = ClipInterpolator(net, precision="bf16x3", graph=graph, filters_f16=f16, group=1)       # (one fused launch per timestamp)
        N.prof_reset()
        N.prof_enable(not graph)
        got = interp(frame, event, gtex, stamps)
        torch.cuda.synchronize()
        N.prof_enable(False)
        if not graph:
            ran = {k: v[0] for k, v in N.prof_collect().items() if v[0] > 0}
            want, other = ("conv_fwd_f16_ws/kernelconv_fac_img", "conv_fwd_bf16x3_ws/kernelconv_fac")[::1 if f16 else -1]
            assert ran.get(want) == len(stamps) and other not in ran and not any(k.startswith("fac_fwd") for k in ran), ran
        assert exact.std() > 1e-3 and _rel(got.cpu(), exact.cpu()) < 1e-3, (f16, graph, _rel(got.cpu(), exact.cpu()))


def test_inference_bank_runs_modification_fused():
    """An inference weight bank (Engine(train=False), infer_ours.py) makes Modification.forward take the fused kernel: one
    launch labelled .../kernelconv_fac, no FAC launch, no filter tensor -- and the module output still matches the unfused
    module and the CPU oracle (oracle.model_ref.modification) at the path's 1e-3."""
    from oracle import model_ref
    from ebfi_amd import _native as N
    from ebfi_amd import conv, weightbank
    from ebfi_amd.model import Modification
    torch.manual_seed(17)
    mod = Modification(FrameBasech=64, EventBasech=64)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn_like(p) * (1.2 / p[0].numel() ** 0.5) if p.dim() > 1 else 0.05 * torch.randn_like(p))
    frame_feat, event_feat = torch.randn(2, 64, 24, 40), torch.randn(2, 64, 24, 40)
    sd = {"Modification." + k: v.detach().clone() for k, v in mod.state_dict().items()}
    ref = model_ref.modification(sd, "Modification", frame_feat, event_feat)
    mod = mod.cuda().eval()
    conv.set_compute_dtype("bf16x3")
    try:
        def run(bank):
            N.prof_reset()
            N.prof_enable(True)
            with torch.no_grad():
                if bank is None:
                    y = mod(frame_feat.cuda(), event_feat.cuda())
                else:
                    bank.ensure_fresh()
                    with bank.active():
                        y = mod(frame_feat.cuda(), event_feat.cuda())
            torch.cuda.synchronize()
            N.prof_enable(False)
            return y, {k: v[0] for k, v in N.prof_collect().items() if v[0] > 0}
        y0, prof0 = run(None)
        y1, prof1 = run(weightbank.build_for(mod, inference=True))
        y2, prof2 = run(weightbank.build_for(mod))                      # a training bank: packed weights, unfused pair
    finally:
        conv.set_compute_dtype("fp32")
    assert prof1.get("conv_fwd_bf16x3_ws/kernelconv_fac") == 1 and not any(k.startswith("fac_fwd") for k in prof1), prof1
    assert any(k.startswith("fac_fwd") for k in prof0) and any(k.startswith("fac_fwd") for k in prof2)
    assert "conv_fwd_bf16x3_ws/kernelconv_fac" not in prof0 and "conv_fwd_bf16x3_ws/kernelconv_fac" not in prof2
    assert _rel(y1.cpu(), y0.cpu()) < 1e-5 and _rel(y2.cpu(), y0.cpu()) < 1e-5
    assert _rel(y1.cpu(), ref) < 1e-3


@pytest.mark.parametrize("forward_f16", [None, "filters"])
def test_kernelconv_fac_training_form_on_fp16_planes_vs_unfused_pair(forward_f16):
    """ebfi_amd.fac.KernelConvFacTrain (training half): the filters and their gradient exist only as planar fp16
    tensors, and -- with the fp16-operand forward, the training step's default -- the convolution's input cat([ev, frame], 1) only
    as the fp16 image written from its two parts (ebfi_to_c16_cat2).  Against the unfused fp32 pair on the same weights and inputs
    (model_singleframe.py:159-163): the output within the fp16 rounding of the filters (a few 1e-4 of its scale; 2e-3 with fp16
    operands), every gradient -- wrt the frame features, the event features (both of their paths), the weight and the bias --
    within 3e-3 of its norm; the node saves the filters as fp16 planes, never an fp32 [B, C*25, h, w] tensor, and no fp32
    concatenation."""
    from ebfi_amd import _native as N
    from ebfi_amd import conv, f16scale, weightbank
    from ebfi_amd.fac import KernelConv2D, KernelConvFacTrain, kernelconv_fac_train_usable
    torch.manual_seed(7)
    B, C, H, W, K = 2, 64, 24, 64, 5
    Cin = 2 * C
    w = torch.nn.Parameter((torch.randn(C * K * K, Cin, 3, 3) * (1.0 / (Cin * 9) ** 0.5)).cuda())
    b = torch.nn.Parameter((torch.randn(C * K * K) * 0.1).cuda())
    frame = torch.randn(B, C, H, W).cuda()
    ev = torch.randn(B, C, H, W).cuda()
    gout = torch.randn(B, C, H, W).cuda() * 1e-2
    bank = weightbank.WeightBank([w, b])
    site = bank.register(w, b, "id", fwd16=forward_f16 is not None)
    book = f16scale.ScaleBook("cuda")
    book.forward_f16 = forward_f16
    bank.attach_scale_book(book)
    bank.refresh()
    conv.set_compute_dtype("bf16x3")
    try:
        # reference: conv (split precision) + LeakyReLU -> fp32 filters -> FAC, autograd through both ops and the concatenation
        with bank.active(), book.active():
            f1, e1 = frame.clone().requires_grad_(), ev.clone().requires_grad_()
            filt = conv.conv_bias_act(torch.cat([e1, f1], 1), w, b, 1, 1, conv.ACT_LEAKY, 0.01, grad_preact=True)
            ref = KernelConv2D(K)(e1, filt, kernel_leaky_slope=0.01)
            book.operand((site.key, "f"), filt)            # (what the model's calibration pass does)
            ref.backward(gout)
            book.finish()
        ref_g = (f1.grad.clone(), e1.grad.clone(), w.grad.clone(), b.grad.clone())
        w.grad = b.grad = None
        assert kernelconv_fac_train_usable(site, book, frame, ev, K)
        with bank.active(), book.active():
            f2, e2 = frame.clone().requires_grad_(), ev.clone().requires_grad_()
            N.prof_reset()
            N.prof_enable(True)
            out = KernelConvFacTrain.apply(f2, e2, site, 0.01, K, w, b)
            saved = [(tuple(t.shape), t.dtype) for t in out.grad_fn.saved_tensors]
            out.backward(gout)
            book.finish()
        torch.cuda.synchronize()
        N.prof_enable(False)
        prof = N.prof_collect()
    finally:
        conv.set_compute_dtype("fp32")
    assert int(book.guard[0].item()) == 0
    assert ("to_c16/cat2" in prof) == (forward_f16 is not None)
    assert _rel(out.detach(), ref.detach()) < (2e-3 if forward_f16 else 1e-3)
    for got, want, name in zip((f2.grad, e2.grad, w.grad, b.grad), ref_g, ("frame", "ev", "weight", "bias")):
        err = ((got - want).norm() / want.norm()).item()
        # (fp16 operands move ~1e-3 of the filters across LeakyReLU's kink against the split-precision reference: each flips the slope
        #  of its gradient, ~3 % of the gradient's norm at these random weights.  What the two-part image changes is pinned bit for bit
        #  by test_gpu_c16.py::test_image_of_a_concatenation_from_its_two_parts; here the fp16-forward arm is a sanity bound.)
        assert err < (5e-2 if forward_f16 else 3e-3), (name, err)
    # what the node keeps for its backward: the fp16 image of the conv input, the padded feature map, the fp16 filter planes
    assert ((B, C * K * K, H, W), torch.float16) in saved and not any(dt == torch.float32 and len(sh) == 4 and sh[1] == C * K * K for sh, dt in saved)
    assert not any(dt == torch.float32 and len(sh) == 4 and sh[1] == Cin for sh, dt in saved)


@pytest.mark.parametrize("mode,p", [("reflect", 3), ("replicate", 2), ("reflect", 1), ("replicate", 4)])
def test_deterministic_pad_adjoints_match_torch(mode, p):
    """csrc/imgops.hip pad2d_bwd (the adjoints of ReflectionPad2d / ReplicationPad2d as gathers in a fixed order) against torch's
    backward of F.pad in float64 on the CPU, on shapes with short and long sides; twice the same bits."""
    from ebfi_amd import fused
    torch.manual_seed(p)
    for shape in [(2, 3, 9, 14), (1, 16, 64, 40), (3, 2, 5, 5)]:
        x = torch.randn(*shape, dtype=torch.float64, requires_grad=True)
        g = torch.randn(shape[0], shape[1], shape[2] + 2 * p, shape[3] + 2 * p, dtype=torch.float64)
        torch.nn.functional.pad(x, (p, p, p, p), mode=mode).backward(g)
        fn = fused.reflect_pad2d if mode == "reflect" else fused.replicate_pad2d
        outs = []
        for _ in range(2):
            xd = x.detach().float().cuda().requires_grad_()
            y = fn(xd, p)
            assert torch.equal(y, torch.nn.functional.pad(xd.detach(), (p, p, p, p), mode=mode))
            y.backward(g.float().cuda())
            outs.append(xd.grad.clone())
        assert torch.equal(outs[0], outs[1])
        assert torch.allclose(outs[0].cpu().double(), x.grad, rtol=1e-6, atol=1e-6), (mode, p, shape)
