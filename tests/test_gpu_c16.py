"""fp16 operand storage (csrc/c16.hpp): every writer and reader of the c16 images against plain torch and against the fp32-input
forms of the same kernels.  Same fp16 operands in the same MFMA order: the image-reading kernels must reproduce the
fp32-reading ones BIT FOR BIT; the images themselves must equal fp16(value * scale) of the fp32 tensors exactly."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ebfi_amd import _native as N  # noqa: E402


def _ref_image(x, scale):
    B, C, H, W = x.shape
    return (x * scale).half().reshape(B, C // 16, 2, 8, H, W).permute(0, 1, 4, 2, 5, 3).contiguous()


def _banked(cin, cout, groups=1):
    from ebfi_amd import f16scale, weightbank
    w = torch.nn.Parameter((torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5).cuda())
    b = torch.nn.Parameter((torch.randn(cout) * 0.1).cuda())
    bank = weightbank.WeightBank([w, b])
    site = bank.register(w, b, "id", groups=groups)
    book = f16scale.ScaleBook("cuda")
    bank.attach_scale_book(book)
    bank.refresh()
    return w, b, bank, book, site


@pytest.mark.parametrize("B,C0,C1,H,W", [(2, 64, 64, 16, 64), (1, 8, 24, 13, 36), (3, 40, 8, 8, 4)])
def test_image_of_a_concatenation_from_its_two_parts(B, C0, C1, H, W):
    """ebfi_to_c16_cat2 == ebfi_to_c16 of torch.cat, bit for bit, image and recorded |max|."""
    from ebfi_amd import c16, f16scale
    torch.manual_seed(C0 + C1)
    a, b = torch.randn(B, C0, H, W).cuda(), torch.randn(B, C1, H, W).cuda() * 3.0
    book = f16scale.ScaleBook("cuda")
    i, j = book.slot("one"), book.slot("two")
    book.calibrate(i, a, b)
    book.calibrate(j, a, b)
    ref = c16.to_c16(torch.cat([a, b], 1), book.ptr(i))
    got = c16.to_c16_cat2(a, b, book.ptr(j))
    assert torch.equal(got, ref) and book.amax(i) == book.amax(j) == max(a.abs().max().item(), b.abs().max().item())


@pytest.mark.parametrize("B,C,H,W,masked", [(2, 64, 16, 64, False), (1, 128, 13, 36, True), (3, 16, 8, 4, False)])
def test_to_c16_matches_torch(B, C, H, W, masked):
    from ebfi_amd import c16, f16scale
    torch.manual_seed(B + C)
    book = f16scale.ScaleBook("cuda")
    i = book.slot("t")
    book.slots[f16scale.SLOT_STRIDE * i] = 0.25
    x = (torch.randn(B, C, H, W) * 3).cuda()
    y = torch.randn(B, C, H, W).cuda()
    img = c16.to_c16(x, book.ptr(i), y if masked else None, 0.01)
    ref = x * torch.where(y > 0, 1.0, 0.01) if masked else x
    assert torch.equal(img, _ref_image(ref, 0.25))
    assert book.amax(i) == ref.abs().max().item()
    assert torch.equal(c16.from_c16(img, 0.25), (ref * 0.25).half().float() / 0.25)


@pytest.mark.parametrize("B,Cin,H,W,Cout,groups", [(2, 64, 16, 64, 128, 1), (1, 64, 13, 36, 128, 2), (3, 128, 20, 132, 64, 1)])
def test_forward_conv_writes_the_image_of_its_output(B, Cin, H, W, Cout, groups):
    from ebfi_amd import c16, f16scale
    torch.manual_seed(1)
    w, b, bank, book, site = _banked(Cin, Cout, groups)
    i = book.slot("out")
    book.slots[f16scale.SLOT_STRIDE * i] = 2.0
    x = torch.randn(B, groups * Cin, H, W).cuda()
    out = torch.empty(B, Cout, H, W, device="cuda")
    img = c16.empty(B, Cout, H, W, "cuda").fill_(7.0)
    plain = torch.empty_like(out)
    lib = N.lib()
    st = N.stream_ptr(x.device)
    args = (N.ptr(x), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()))
    geo = (B, Cin, H, W, Cout, 3, 1, groups, 1, 0.01, N.ptr(None), N.ptr(None), 0, 0.0)
    N.check(lib.ebfi_conv2d_packed_x3(*args, N.ptr(plain), *geo, st), "x3")
    N.check(lib.ebfi_conv2d_packed_x3_c16(*args, N.ptr(out), *geo, N.ptr(img), book.ptr(i), 0, st), "x3_c16")
    assert torch.equal(out, plain)                               # the fp32 output is unchanged by the side image
    assert torch.equal(img, _ref_image(out, 2.0))
    assert book.amax(i) == out.abs().max().item()


@pytest.mark.parametrize("B,Cin,H,W,Cout,groups,extras", [
    (2, 64, 16, 64, 128, 1, False), (1, 64, 13, 36, 128, 2, True), (2, 128, 24, 68, 64, 1, True), (3, 64, 342, 160, 64, 1, False)])
def test_data_gradient_reads_and_writes_images(B, Cin, H, W, Cout, groups, extras):
    """ebfi_conv2d_packed_f16_c16 (the data gradient as a convolution with the transposed fp16 images): fp16-image input vs
    fp32 input bit for bit; fp16-image output == fp16(fp32 output * scale); a NULL fp32 output writes only the image."""
    from ebfi_amd import c16, f16scale
    torch.manual_seed(2)
    # `site` as the TRANSPOSED images of a layer Cout <- Cin: the data gradient maps Cout-channel gradients to Cin channels
    w, b, bank, book, site = _banked(Cout // groups if groups > 1 else Cout, Cin * groups if groups > 1 else Cin, groups)
    # (the registered layer maps (groups * site.K) -> site.M; its data gradient reads site.M channels and writes groups * site.K)
    gch, och = site.M, site.K * groups
    g = torch.randn(B, gch, H, W).cuda() * 3e-3
    si, so = book.slot("g"), book.slot("o")
    book.calibrate(si, g)
    book.slots[f16scale.SLOT_STRIDE * so] = 64.0
    g16 = c16.to_c16(g, book.ptr(si))
    addend = torch.randn(B, och, H, W).cuda() * 1e-3 if extras else None
    mask = torch.randn(B, och, H, W).cuda() if extras else None
    lib, st = N.lib(), N.stream_ptr(g.device)

    def run(inp, is16, out, out16):
        rc = lib.ebfi_conv2d_packed_f16_c16(N.ptr(inp), is16, site.tr16_ptr(), site.tr16_bytes, N.ptr(None), N.ptr(out), B,
                                            gch // groups, H, W, och, 3, 1, groups, 0, 0.0, N.ptr(addend), N.ptr(mask),
                                            1 if extras else 0, 0.01 if extras else 0.0, book.ptr(si), site.w_slot_ptr(),
                                            N.ptr(out16), book.ptr(so) if out16 is not None else N.ptr(None), 0, 0, st)
        N.check(rc, "ebfi_conv2d_packed_f16_c16")
    ref32 = torch.empty(B, och, H, W, device="cuda")
    run(g, 0, ref32, None)
    a32 = torch.empty_like(ref32)
    run(g16, 1, a32, None)
    assert torch.equal(a32, ref32)
    b32, b16 = torch.empty_like(ref32), c16.empty(B, och, H, W, "cuda")
    run(g16, 1, b32, b16)
    assert torch.equal(b32, ref32) and torch.equal(b16, _ref_image(ref32, 64.0))
    only16 = c16.empty(B, och, H, W, "cuda")
    run(g16, 1, None, only16)
    assert torch.equal(only16, b16)
    assert book.amax(so) == ref32.abs().max().item()
    if extras:
        # the mask as the c16 IMAGE of the mask tensor (only its signs are read): the same result bit for bit
        sm = book.slot("m")
        book.calibrate(sm, mask)
        mask16 = c16.to_c16(mask, book.ptr(sm))
        m32, m16 = torch.empty_like(ref32), c16.empty(B, och, H, W, "cuda")
        rc = lib.ebfi_conv2d_packed_f16_c16(N.ptr(g16), 1, site.tr16_ptr(), site.tr16_bytes, N.ptr(None), N.ptr(m32), B, gch // groups, H, W,
                                            och, 3, 1, groups, 0, 0.0, N.ptr(addend), N.ptr(mask16), 1, 0.01, book.ptr(si),
                                            site.w_slot_ptr(), N.ptr(m16), book.ptr(so), 0, 1, st)
        N.check(rc, "ebfi_conv2d_packed_f16_c16 (image mask)")
        assert torch.equal(m32, ref32) and torch.equal(m16, b16)
    # against fp32 math (loose: fp16 operands)
    wt = w.detach()
    ref = torch.nn.functional.conv_transpose2d(g.cpu(), wt.cpu(), None, 1, 1, groups=groups) if True else None
    if extras:
        ref = (ref + addend.cpu()) * torch.where(mask.cpu() > 0, 1.0, 0.01)
    assert ((ref32.cpu() - ref).abs().max() / ref.abs().max()).item() < 2e-3


@pytest.mark.parametrize("B,Cin,H,W,Cout,groups", [(2, 64, 16, 64, 64, 1), (1, 64, 13, 36, 128, 2), (2, 128, 20, 68, 200 // 8 * 8, 1),
                                                   (2, 48, 16, 32, 32, 1), (3, 64, 342, 160, 64, 1)])
def test_weight_gradient_from_images(B, Cin, H, W, Cout, groups):
    """ebfi_conv2d_backward_weight_f16c vs ebfi_conv2d_backward_weight_f16g on the same tensors: the same fp16 operands in the
    same order -> grad_weight bit for bit; grad_bias from the fp16-rounded gradient (1e-3)."""
    from ebfi_amd import c16, f16scale
    torch.manual_seed(3)
    Cout = (Cout + 15) // 16 * 16
    x = torch.randn(B, groups * Cin, H, W).cuda() * 0.3
    g = torch.randn(B, Cout, H, W).cuda() * 2e-2
    book = f16scale.ScaleBook("cuda")
    sx, sg = book.slot("x"), book.slot("g")
    book.calibrate(sx, x)
    book.calibrate(sg, g)
    lib, st = N.lib(), N.stream_ptr(x.device)
    need = int(lib.ebfi_conv2d_backward_weight_workspace(B, Cin, H, W, Cout, 3, 1, 1, N.EBFI_F32))
    ws = torch.empty(max(need, 4), dtype=torch.uint8, device="cuda")
    gw_a, gb_a = torch.empty(Cout, Cin, 3, 3, device="cuda"), torch.empty(Cout, device="cuda")
    rc = lib.ebfi_conv2d_backward_weight_f16g(N.ptr(x), N.ptr(g), N.ptr(None), N.ptr(gw_a), N.ptr(gb_a), N.ptr(None), B, Cin, H, W, Cout,
                                              3, 1, groups, 0, 0.0, book.ptr(sx), book.ptr(sg), N.ptr(ws), need, st)
    N.check(rc, "f16g")
    x16, g16 = c16.to_c16(x, book.ptr(sx)), c16.to_c16(g, book.ptr(sg))
    gw_b, gb_b = torch.empty_like(gw_a), torch.empty_like(gb_a)
    rc = lib.ebfi_conv2d_backward_weight_f16c(N.ptr(x16), N.ptr(g16), 0, N.ptr(gw_b), N.ptr(gb_b), B, Cin, H, W, Cout, groups,
                                              book.ptr(sx), book.ptr(sg), N.ptr(ws), need, st)
    N.check(rc, "f16c")
    assert torch.equal(gw_a, gw_b)
    assert ((gb_a - gb_b).abs().max() / gb_a.abs().max()).item() < 1e-3
    ref_b = g.sum((0, 2, 3))
    assert ((gb_b - ref_b).abs().max() / ref_b.abs().max()).item() < 1e-3


@pytest.mark.parametrize("B,H,W,n", [(8, 128, 128, 3), (2, 52, 72, 3), (2, 37, 100, 2), (4, 64, 64, 1)])
def test_batched_weight_gradients_equal_the_single_launches(B, H, W, n):
    """ebfi_conv2d_backward_weight_f16c_batch (the three layers of a ResidualControl round in one launch: 64 -> 128, 2 x (64 -> 64)
    grouped, 128 -> 64; pixel splits laid out per XCD, ragged tile counts) against ebfi_conv2d_backward_weight_f16c per layer: the
    same fp16 products, summed over a different number of pixel splits -> equal to fp32 summation order (2e-6 of the tensor's
    scale), and against the exact fp32 weight gradient of the de-quantised operands."""
    import ctypes
    from ebfi_amd import c16, f16scale
    torch.manual_seed(5)
    layers = [(64, 128, 1), (64, 128, 2), (128, 64, 1)][:n]
    book = f16scale.ScaleBook("cuda")
    lib, st = N.lib(), N.stream_ptr(torch.device("cuda"))
    xs, gs, imgs, slots, single = [], [], [], [], []
    for k, (cin, cout, groups) in enumerate(layers):
        x = torch.randn(B, groups * cin, H, W).cuda() * 0.4
        g = torch.randn(B, cout, H, W).cuda() * 3e-2
        sx, sg = book.slot(("x", k)), book.slot(("g", k))
        book.calibrate(sx, x)
        book.calibrate(sg, g)
        x16, g16 = c16.to_c16(x, book.ptr(sx)), c16.to_c16(g, book.ptr(sg))
        need = int(lib.ebfi_conv2d_backward_weight_workspace(B, cin, H, W, cout, 3, 1, 1, N.EBFI_F32))
        ws = torch.empty(max(need, 4), dtype=torch.uint8, device="cuda")
        gw, gb = torch.empty(cout, cin, 3, 3, device="cuda"), torch.empty(cout, device="cuda")
        N.check(lib.ebfi_conv2d_backward_weight_f16c(N.ptr(x16), N.ptr(g16), 0, N.ptr(gw), N.ptr(gb), B, cin, H, W, cout, groups,
                                                     book.ptr(sx), book.ptr(sg), N.ptr(ws), need, st), "f16c")
        xs.append(x); gs.append(g); imgs.append((x16, g16)); slots.append((sx, sg)); single.append((gw, gb))
    ints = lambda v: (ctypes.c_int * n)(*v)
    ptrs = lambda v: (ctypes.c_void_p * n)(*v)
    cin_a, cout_a, gr_a = ints([l[0] for l in layers]), ints([l[1] for l in layers]), ints([l[2] for l in layers])
    need = int(lib.ebfi_conv2d_backward_weight_f16c_batch_workspace(n, cin_a, cout_a))
    assert need == sum(8 * (32 // (2 * n)) * (l[0] * l[1] * 9 + l[1]) * 4 for l in layers)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    out = [(torch.full((l[1], l[0], 3, 3), float("nan"), device="cuda"), torch.full((l[1],), float("nan"), device="cuda")) for l in layers]
    rc = lib.ebfi_conv2d_backward_weight_f16c_batch(
        n, ptrs([a.data_ptr() for a, _ in imgs]), ptrs([b.data_ptr() for _, b in imgs]), ptrs([w.data_ptr() for w, _ in out]),
        ptrs([b.data_ptr() for _, b in out]), cin_a, cout_a, gr_a, ptrs([book.ptr(a).value for a, _ in slots]),
        ptrs([book.ptr(b).value for _, b in slots]), B, H, W, N.ptr(ws), need, st)
    N.check(rc, "f16c_batch")
    for k, (cin, cout, groups) in enumerate(layers):
        (gw, gb), (rw, rb) = out[k], single[k]
        assert torch.isfinite(gw).all() and torch.isfinite(gb).all()
        assert ((gw - rw).abs().max() / rw.abs().max()).item() < 2e-6, k
        assert ((gb - rb).abs().max() / rb.abs().max()).item() < 2e-6, k
        # exact fp32 reference on the operands the kernels saw (fp16-rounded at the slot scales)
        xq = c16.from_c16(imgs[k][0], book.scale(slots[k][0]))
        gq = c16.from_c16(imgs[k][1], book.scale(slots[k][1]))
        ref = torch.nn.grad.conv2d_weight(xq.double(), (cout, cin, 3, 3), gq.double(), padding=1, groups=groups).float()
        assert ((gw - ref).abs().max() / ref.abs().max()).item() < 1e-5, k
    # a layer that is not two 64 x 64 blocks is refused (callers fall back to the per-layer entry)
    bad = ints([64] * n)
    rc = lib.ebfi_conv2d_backward_weight_f16c_batch(
        n, ptrs([a.data_ptr() for a, _ in imgs]), ptrs([b.data_ptr() for _, b in imgs]), ptrs([w.data_ptr() for w, _ in out]),
        ptrs([b.data_ptr() for _, b in out]), bad, bad, ints([1] * n), ptrs([book.ptr(a).value for a, _ in slots]),
        ptrs([book.ptr(b).value for _, b in slots]), B, H, W, N.ptr(ws), need, st)
    assert rc != 0 and b"two 64 x 64 blocks" in lib.ebfi_last_error()


@pytest.mark.parametrize("B,Cin,H,W,Cout", [(2, 64, 16, 64, 64), (1, 48, 13, 36, 32), (2, 128, 24, 68, 80), (1, 64, 260, 256, 64)])
def test_weight_gradient_writes_the_image_of_the_preactivation_gradient(B, Cin, H, W, Cout):
    """ebfi_conv2d_backward_weight_f16g_ex with grad_preact_is_c16: grad_out * LeakyReLU'(saved_output) leaves as the c16 image the
    data gradient stages -- exactly fp16(value * scale) of the fp32 tensor the plain form writes; grad_weight / grad_bias unchanged;
    the data gradient from the image equals the data gradient from the fp32 tensor bit for bit."""
    from ebfi_amd import c16
    torch.manual_seed(7)
    w, b, bank, book, site = _banked(Cin, Cout)
    x = torch.randn(B, Cin, H, W).cuda() * 0.5
    y = torch.randn(B, Cout, H, W).cuda()
    g = torch.randn(B, Cout, H, W).cuda() * 1e-2
    sx, sg = book.slot("x"), book.slot("g")
    book.calibrate(sx, x)
    book.calibrate(sg, g)
    lib, st = N.lib(), N.stream_ptr(x.device)
    need = int(lib.ebfi_conv2d_backward_weight_workspace(B, Cin, H, W, Cout, 3, 1, 1, N.EBFI_F32))
    ws = torch.empty(max(need, 4), dtype=torch.uint8, device="cuda")

    def wgrad(gpre, is16):
        gw, gb = torch.empty(Cout, Cin, 3, 3, device="cuda"), torch.empty(Cout, device="cuda")
        rc = lib.ebfi_conv2d_backward_weight_f16g_ex(N.ptr(x), N.ptr(g), N.ptr(y), N.ptr(gw), N.ptr(gb), N.ptr(gpre), is16, B, Cin, H, W,
                                                     Cout, 3, 1, 1, 1, 0.01, book.ptr(sx), book.ptr(sg), N.ptr(ws), need, st)
        N.check(rc, "ebfi_conv2d_backward_weight_f16g_ex")
        return gw, gb
    gpre32 = torch.empty_like(g)
    gw0, gb0 = wgrad(gpre32, 0)
    gpre16 = torch.full_like(c16.empty(B, Cout, H, W, "cuda"), float("nan"))
    gw1, gb1 = wgrad(gpre16, 1)
    assert torch.equal(gw0, gw1) and torch.equal(gb0, gb1)
    ref = g * torch.where(y > 0, 1.0, 0.01)
    assert torch.equal(gpre32, ref)
    assert torch.equal(gpre16, _ref_image(ref, book.scale(sg)))
    if Cin >= 48:                       # the data gradient reads either form of the same values
        a, r = torch.empty(B, Cin, H, W, device="cuda"), torch.empty(B, Cin, H, W, device="cuda")
        for inp, mode, out in ((gpre32, 0, r), (gpre16, 1, a)):
            N.check(lib.ebfi_conv2d_packed_f16_c16(N.ptr(inp), mode, site.tr16_ptr(), site.tr16_bytes, N.ptr(None), N.ptr(out), B, Cout, H, W,
                                                   Cin, 3, 1, 1, 0, 0.0, N.ptr(None), N.ptr(None), 0, 0.0, book.ptr(sg), site.w_slot_ptr(),
                                                   N.ptr(None), N.ptr(None), 0, 0, st), "dgrad")
        assert torch.equal(a, r)


def test_fused_residual_control_stages_write_images():
    from ebfi_amd import c16, f16scale
    torch.manual_seed(4)
    B, C, H, W = 2, 64, 16, 32
    HW = H * W
    a = torch.randn(B, 2 * C, H, W).cuda()
    s0, s1 = torch.randn(B, C).cuda(), torch.randn(B, C).cuda()
    x = torch.randn(B, C, H, W).cuda()
    book = f16scale.ScaleBook("cuda")
    so, sg = book.slot("c"), book.slot("ga")
    book.slots[f16scale.SLOT_STRIDE * so] = 4.0
    book.slots[f16scale.SLOT_STRIDE * sg] = 512.0
    lib, st = N.lib(), N.stream_ptr(a.device)
    a1p = N._vp(a.data_ptr() + 4 * C * HW)
    out, out16 = torch.empty(B, 2 * C, H, W, device="cuda"), c16.empty(B, 2 * C, H, W, "cuda")
    N.check(lib.ebfi_scale_residual_cat_forward_c16(N.ptr(a), N.ptr(s0), a1p, N.ptr(s1), N.ptr(x), N.ptr(out), N.ptr(out16), book.ptr(so),
                                                    B, C, H, W, 2 * C * HW, st), "fwd_c16")
    ref = torch.empty_like(out)
    N.check(lib.ebfi_scale_residual_cat_forward_ex(N.ptr(a), N.ptr(s0), a1p, N.ptr(s1), N.ptr(x), N.ptr(ref), B, C, HW, 2 * C * HW, st), "fwd")
    assert torch.equal(out, ref) and torch.equal(out16, _ref_image(ref, 4.0)) and book.amax(so) == ref.abs().max().item()
    # backward
    gc = torch.randn(B, 2 * C, H, W).cuda() * 1e-2
    ga = torch.empty(B, 2 * C, H, W, device="cuda")
    gx, gs0, gs1 = torch.empty(B, C, H, W, device="cuda"), torch.empty(B, C, device="cuda"), torch.empty(B, C, device="cuda")
    N.check(lib.ebfi_scale_residual_cat_backward_ex(N.ptr(gc), N.ptr(a), N.ptr(s0), a1p, N.ptr(s1), N.ptr(ga), N._vp(ga.data_ptr() + 4 * C * HW),
                                                    N.ptr(gx), N.ptr(gs0), N.ptr(gs1), B, C, HW, 2 * C * HW, 2 * C * HW, 1, 0.01, st), "bwd")
    S = int(lib.ebfi_scale_residual_cat_backward_slices())
    ga16 = c16.empty(B, 2 * C, H, W, "cuda")
    gx2, p0, p1 = torch.empty_like(gx), torch.empty(S, B, C, device="cuda"), torch.empty(S, B, C, device="cuda")
    N.check(lib.ebfi_scale_residual_cat_backward_c16(N.ptr(gc), N.ptr(a), N.ptr(s0), a1p, N.ptr(s1), N.ptr(ga16), book.ptr(sg), N.ptr(gx2),
                                                     N.ptr(p0), N.ptr(p1), B, C, H, W, 2 * C * HW, 0.01, st), "bwd_c16")
    assert torch.equal(gx2, gx) and torch.equal(ga16, _ref_image(ga, 512.0)) and book.amax(sg) == ga.abs().max().item()
    assert torch.allclose(p0.sum(0), gs0, rtol=1e-5, atol=1e-7) and torch.allclose(p1.sum(0), gs1, rtol=1e-5, atol=1e-7)


# ---------------------------------------------------------------------------------------------------------------------------
# planar fp16 storage of the KernelConv -> FAC pair's 1600-channel tensors (filters, grad_kernel)
def _p16(x, scale):
    return (x * scale).half()


@pytest.mark.parametrize("B,C,H,W", [(2, 8, 16, 64), (1, 64, 24, 132), (2, 3, 9, 36)])
def test_fac_on_fp16_filter_planes(B, C, H, W):
    """ebfi_fac_forward_p16 / _backward_p16 against the fp32 FAC kernels fed the de-quantised filters: same arithmetic, bit for
    bit (output, grad_input); grad_kernel16 == fp16(grad_kernel32 * scale) exactly, its |max| recorded."""
    from ebfi_amd import f16scale
    from ebfi_amd.fac import fac_backward, fac_forward
    torch.manual_seed(C + W)
    K = 5
    xp = torch.randn(B, C, H + 4, W + 4).cuda()
    filt = torch.randn(B, C * K * K, H, W).cuda() * 0.2
    go = torch.randn(B, C, H, W).cuda() * 1e-2
    book = f16scale.ScaleBook("cuda")
    sf, sg = book.slot("f"), book.slot("g")
    book.calibrate(sf, filt)
    book.slots[f16scale.SLOT_STRIDE * sg] = 256.0
    f16 = _p16(filt, book.scale(sf))
    fq = f16.float() / book.scale(sf)                      # what the kernels see
    lib, st = N.lib(), N.stream_ptr(xp.device)
    out = torch.empty(B, C, H, W, device="cuda")
    N.check(lib.ebfi_fac_forward_p16(N.ptr(xp), 0, N.ptr(f16), book.ptr(sf), N.ptr(out), B, C, H, W, K, st), "fac_forward_p16")
    assert torch.equal(out, fac_forward(xp, fq, K))
    gin = torch.empty_like(xp)
    gk16 = torch.empty(B, C * K * K, H, W, dtype=torch.float16, device="cuda")
    N.check(lib.ebfi_fac_backward_p16(N.ptr(xp), 0, N.ptr(f16), book.ptr(sf), N.ptr(go), N.ptr(gin), N.ptr(gk16), book.ptr(sg), 0.01,
                                      B, C, H, W, K, st), "fac_backward_p16")
    rin, rk = fac_backward(xp, fq, K, go, kernel_leaky_slope=0.01)
    assert torch.equal(gin, rin)
    assert torch.equal(gk16, _p16(rk, 256.0)) and book.amax(sg) == rk.abs().max().item()
    # ---- the replicate padding INSIDE the kernels (round 5): the unpadded tensor in, the gradient of the unpadded tensor out
    ev = torch.randn(B, C, H, W).cuda()
    evp = torch.nn.functional.pad(ev, (2, 2, 2, 2), mode="replicate")
    ref_out, out_u = torch.empty_like(out), torch.empty_like(out)
    N.check(lib.ebfi_fac_forward_p16(N.ptr(evp), 0, N.ptr(f16), book.ptr(sf), N.ptr(ref_out), B, C, H, W, K, st), "padded")
    N.check(lib.ebfi_fac_forward_p16(N.ptr(ev), 1, N.ptr(f16), book.ptr(sf), N.ptr(out_u), B, C, H, W, K, st), "unpadded")
    assert torch.equal(out_u, ref_out)
    gp, gk_p = torch.empty_like(evp), torch.empty_like(gk16)
    N.check(lib.ebfi_fac_backward_p16(N.ptr(evp), 0, N.ptr(f16), book.ptr(sf), N.ptr(go), N.ptr(gp), N.ptr(gk_p), book.ptr(sg), 0.01,
                                      B, C, H, W, K, st), "padded bwd")
    gu, gk_u = torch.full_like(ev, float("nan")), torch.empty_like(gk16)
    N.check(lib.ebfi_fac_backward_p16(N.ptr(ev), 1, N.ptr(f16), book.ptr(sf), N.ptr(go), N.ptr(gu), N.ptr(gk_u), book.ptr(sg), 0.01,
                                      B, C, H, W, K, st), "unpadded bwd")
    # grad_kernel: where the filter is > 0 the two kernels form the same product (bit-equal); where it is <= 0 the 8-pixel kernel of
    # the unpadded path multiplies in * (grad_out * slope) and the 4-pixel one (in * grad_out) * slope -- one fp32 rounding apart
    # before the fp16 rounding: a few elements in a thousand differ by one fp16 ulp
    pos = (f16 > 0)
    assert torch.equal(gk_u[pos], gk_p[pos])
    dk = (gk_u.float() - gk_p.float()).abs()
    assert (dk > 0).float().mean().item() < 5e-3 and (dk <= gk_p.float().abs() * 2.0 ** -10 + 1e-12).all()
    # the padding's adjoint in float64 from the padded gradient: interior elements are copies, border elements sums of <= 9 terms
    yy = torch.arange(H + 4, device="cuda").sub(2).clamp(0, H - 1)
    xx = torch.arange(W + 4, device="cuda").sub(2).clamp(0, W - 1)
    rows = torch.zeros(B, C, H, W + 4, dtype=torch.float64, device="cuda").index_add_(2, yy, gp.double())
    ref_g = torch.zeros(B, C, H, W, dtype=torch.float64, device="cuda").index_add_(3, xx, rows)
    assert torch.isfinite(gu).all()
    # (interior elements are the same sums; the 8-pixel kernel adds a column's terms in another association than the 4-pixel one --
    #  neighbouring threads' partial sums meet at other places -- so the comparison is to fp32 rounding, not bit for bit)
    assert ((gu[:, :, 1:-1, 1:-1] - gp[:, :, 3:-3, 3:-3]).abs().max() / gp.abs().max()).item() < 1e-6
    assert ((gu.double() - ref_g).abs().max() / ref_g.abs().max()).item() < 1e-6
    gu2 = torch.empty_like(ev)                          # fixed summation order: bit-identical from run to run
    N.check(lib.ebfi_fac_backward_p16(N.ptr(ev), 1, N.ptr(f16), book.ptr(sf), N.ptr(go), N.ptr(gu2), N.ptr(gk_u), book.ptr(sg), 0.01,
                                      B, C, H, W, K, st), "unpadded bwd again")
    assert torch.equal(gu, gu2)


@pytest.mark.parametrize("B,Cin,H,W,Cout", [(2, 128, 16, 64, 200), (1, 64, 13, 36, 64)])
def test_conv_writes_planar_fp16_filters_only(B, Cin, H, W, Cout):
    from ebfi_amd import f16scale
    torch.manual_seed(5)
    w, b, bank, book, site = _banked(Cin, Cout)
    i = book.slot("f")
    book.slots[f16scale.SLOT_STRIDE * i] = 8.0
    x = torch.randn(B, Cin, H, W).cuda()
    ref = torch.empty(B, Cout, H, W, device="cuda")
    lib, st = N.lib(), N.stream_ptr(x.device)
    args = (N.ptr(x), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()))
    geo = (B, Cin, H, W, Cout, 3, 1, 1, 1, 0.01, N.ptr(None), N.ptr(None), 0, 0.0)
    N.check(lib.ebfi_conv2d_packed_x3(*args, N.ptr(ref), *geo, st), "x3")
    f16 = torch.full((B, Cout, H, W), 7.0, dtype=torch.float16, device="cuda")
    N.check(lib.ebfi_conv2d_packed_x3_c16(*args, N.ptr(None), *geo, N.ptr(f16), book.ptr(i), 1, st), "x3 planar16")
    assert torch.equal(f16, _p16(ref, 8.0)) and book.amax(i) == ref.abs().max().item()


@pytest.mark.parametrize("B,Cin,H,W,Cout", [(2, 128, 16, 64, 200), (1, 64, 24, 68, 64)])
def test_backward_convs_stage_planar_fp16_gradients(B, Cin, H, W, Cout):
    """The data gradient reading a planar fp16 gradient == the fp32-input kernel fed the de-quantised tensor (bit for bit); the
    weight gradient reading it == the image-reading kernel fed the image of the de-quantised tensor."""
    from ebfi_amd import c16, f16scale
    torch.manual_seed(6)
    w, b, bank, book, site = _banked(Cin, Cout)          # layer Cin -> Cout; its gradients: Cout-channel g, Cin-channel x
    x = torch.randn(B, Cin, H, W).cuda() * 0.5
    g = torch.randn(B, Cout, H, W).cuda() * 1e-2
    sx, sg = book.slot("x"), book.slot("g")
    book.calibrate(sx, x)
    book.calibrate(sg, g)
    g16 = _p16(g, book.scale(sg))
    gq = g16.float() / book.scale(sg)
    lib, st = N.lib(), N.stream_ptr(x.device)

    def dgrad(inp, mode, out):
        N.check(lib.ebfi_conv2d_packed_f16_c16(N.ptr(inp), mode, site.tr16_ptr(), site.tr16_bytes, N.ptr(None), N.ptr(out), B, Cout, H, W, Cin,
                                               3, 1, 1, 0, 0.0, N.ptr(None), N.ptr(None), 0, 0.0, book.ptr(sg), site.w_slot_ptr(),
                                               N.ptr(None), N.ptr(None), 0, 0, st), "dgrad")
    a, r = torch.empty(B, Cin, H, W, device="cuda"), torch.empty(B, Cin, H, W, device="cuda")
    dgrad(gq, 0, r)
    dgrad(g16, 2, a)
    assert torch.equal(a, r)
    need = int(lib.ebfi_conv2d_backward_weight_workspace(B, Cin, H, W, Cout, 3, 1, 1, N.EBFI_F32))
    ws = torch.empty(max(need, 4), dtype=torch.uint8, device="cuda")
    x16 = c16.to_c16(x, book.ptr(sx))
    gw_a, gb_a = torch.empty(Cout, Cin, 3, 3, device="cuda"), torch.empty(Cout, device="cuda")
    gw_b, gb_b = torch.empty_like(gw_a), torch.empty_like(gb_a)
    N.check(lib.ebfi_conv2d_backward_weight_f16c(N.ptr(x16), N.ptr(g16), 1, N.ptr(gw_a), N.ptr(gb_a), B, Cin, H, W, Cout, 1, book.ptr(sx),
                                                 book.ptr(sg), N.ptr(ws), need, st), "wgrad planar")
    if Cout % 16 == 0:
        gimg = c16.to_c16(gq, book.ptr(sg))
        N.check(lib.ebfi_conv2d_backward_weight_f16c(N.ptr(x16), N.ptr(gimg), 0, N.ptr(gw_b), N.ptr(gb_b), B, Cin, H, W, Cout, 1, book.ptr(sx),
                                                     book.ptr(sg), N.ptr(ws), need, st), "wgrad image")
        assert torch.equal(gw_a, gw_b)
        assert ((gb_a - gb_b).abs().max() / gb_b.abs().max()).item() < 1e-4
    xq = c16.from_c16(x16, book.scale(sx))
    ref_w = torch.nn.grad.conv2d_weight(xq.cpu().double(), (Cout, Cin, 3, 3), gq.cpu().double(), padding=1).float()
    assert ((gw_a.cpu() - ref_w).abs().max() / ref_w.abs().max()).item() < 1e-4
    ref_b = gq.sum((0, 2, 3))
    assert ((gb_a - ref_b).abs().max() / ref_b.abs().max()).item() < 1e-4


# ---------------------------------------------------------------------------------------------------------------------------
# The storage forms against INDEPENDENT fp32 math (CPU torch conv2d + autograd on the un-rounded tensors), not against the HIP
# fp32-operand forms of the same kernels: a bug shared by both forms of a kernel would pass every test above.
@pytest.mark.parametrize("B,Cin,H,W,Cout", [(2, 128, 16, 64, 208), (1, 64, 24, 68, 64)])
def test_storage_kernels_against_cpu_fp32_autograd(B, Cin, H, W, Cout):
    """conv_fwd_f16_ws/img_p16 (fp16 image in, fp16 weight image, planar fp16 out: the training step's KernelConv forward),
    conv_fwd_f16_ws/p16_f32 (its data gradient from planar fp16 gradients) and conv_wgrad_f16_tr<0, IN16, GP16> (its weight
    gradient from the image and the planes) against torch's CPU conv2d forward / autograd in fp32 on the SAME tensors before any
    rounding.  Tolerance = what rounding both operands of a product to fp16 (2^-11 each) and, for the forward, the output to fp16
    allows on sums of Cin * 9 terms: 2e-3 of the result's maximum (measured 3-6e-4)."""
    from ebfi_amd import c16, f16scale, weightbank
    torch.manual_seed(Cin + Cout)
    w = torch.nn.Parameter((torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5).cuda())
    b = torch.nn.Parameter((torch.randn(Cout) * 0.1).cuda())
    bank = weightbank.WeightBank([w, b])
    site = bank.register(w, b, "id", fwd16="filters")
    book = f16scale.ScaleBook("cuda")
    bank.attach_scale_book(book)
    bank.refresh()
    assert site.fwd16_ptr() is not None and site.tr16_ptr() is not None
    x = torch.randn(B, Cin, H, W) * 0.7
    g = torch.randn(B, Cout, H, W) * 2e-2
    # ---- the oracle side: plain fp32 on the CPU
    xc = x.clone().requires_grad_()
    wc, bc = w.detach().cpu().clone().requires_grad_(), b.detach().cpu().clone().requires_grad_()
    pre = torch.nn.functional.conv2d(xc, wc, bc, padding=1)
    yc = torch.nn.functional.leaky_relu(pre, 0.01)
    gpre = g * torch.where(pre.detach() > 0, 1.0, 0.01)          # what the FAC backward hands over: grad * act'
    pre.backward(gpre)
    rel = lambda a, r: ((a.float().cpu() - r).abs().max() / r.abs().max()).item()
    # ---- device side
    xd, gd = x.cuda(), gpre.cuda()
    sx, sg, sf = book.slot("x"), book.slot("g"), book.slot("f")
    book.calibrate(sx, xd)
    book.calibrate(sg, gd)
    book.calibrate(sf, yc.cuda())
    lib, st = N.lib(), N.stream_ptr(xd.device)
    x16 = c16.to_c16(xd, book.ptr(sx))
    # forward: image -> planar fp16 filters (conv_fwd_f16_ws/img_p16)
    f16 = torch.empty(B, Cout, H, W, dtype=torch.float16, device="cuda")
    N.check(lib.ebfi_conv2d_packed_f16_c16(N.ptr(x16), 1, site.fwd16_ptr(), site.fwd16_bytes, N.ptr(site.bias()), N.ptr(None), B, Cin, H, W,
                                           Cout, 3, 1, 1, 1, 0.01, N.ptr(None), N.ptr(None), 0, 0.0, book.ptr(sx), site.w_slot_ptr(),
                                           N.ptr(f16), book.ptr(sf), 1, 0, st), "img_p16")
    assert rel(f16.float() / book.scale(sf), yc.detach()) < 2e-3
    # data gradient: planar fp16 gradient -> fp32 (conv_fwd_f16_ws/p16_f32)
    g16 = (gd * book.scale(sg)).half()
    gx = torch.empty(B, Cin, H, W, device="cuda")
    N.check(lib.ebfi_conv2d_packed_f16_c16(N.ptr(g16), 2, site.tr16_ptr(), site.tr16_bytes, N.ptr(None), N.ptr(gx), B, Cout, H, W, Cin,
                                           3, 1, 1, 0, 0.0, N.ptr(None), N.ptr(None), 0, 0.0, book.ptr(sg), site.w_slot_ptr(),
                                           N.ptr(None), N.ptr(None), 0, 0, st), "p16_f32")
    assert rel(gx, xc.grad) < 2e-3
    # weight gradient: image x planes (conv_wgrad_f16_tr<0, IN16, GP16>)
    need = int(lib.ebfi_conv2d_backward_weight_workspace(B, Cin, H, W, Cout, 3, 1, 1, N.EBFI_F32))
    ws = torch.empty(max(need, 4), dtype=torch.uint8, device="cuda")
    gw, gb = torch.empty(Cout, Cin, 3, 3, device="cuda"), torch.empty(Cout, device="cuda")
    N.check(lib.ebfi_conv2d_backward_weight_f16c(N.ptr(x16), N.ptr(g16), 1, N.ptr(gw), N.ptr(gb), B, Cin, H, W, Cout, 1, book.ptr(sx),
                                                 book.ptr(sg), N.ptr(ws), need, st), "wgrad IN16 GP16")
    assert rel(gw, wc.grad) < 2e-3 and rel(gb, bc.grad) < 2e-3


def test_one_nan_element_reaches_the_slot_and_raises_the_guard():
    """The running |max| of every fp16 writer is taken on the float bits (c16.hpp amax_acc), so a tensor that is only PARTLY NaN
    still leaves a NaN pattern in its slot and ebfi_f16_scales_finish raises guard[0] -- fmaxf would have dropped the NaN and
    the saturating conversions would have hidden it (round-4 advisory).  Writers covered: to_c16, the fused ResidualControl
    backward stage (src_bwd_c16) and the forward epilogue's side image."""
    from ebfi_amd import c16, f16scale
    torch.manual_seed(8)
    lib, st = N.lib(), N.stream_ptr(torch.device("cuda"))

    def guard_after(fill):
        book = f16scale.ScaleBook("cuda")
        i = book.slot("t")
        fill(book, i)
        book.finish()
        torch.cuda.synchronize()
        return int(book.guard[0].item())
    x = torch.randn(2, 64, 16, 64).cuda()
    bad = x.clone()
    bad[1, 37, 5, 11] = float("nan")
    assert guard_after(lambda bk, i: c16.to_c16(x, bk.ptr(i))) == 0
    assert guard_after(lambda bk, i: c16.to_c16(bad, bk.ptr(i))) == 1
    # src_bwd_c16: one NaN in the incoming gradient
    B, C, H, W = 2, 64, 16, 32
    HW = H * W
    a = torch.randn(B, 2 * C, H, W).cuda()
    s0, s1 = torch.randn(B, C).cuda(), torch.randn(B, C).cuda()
    a1p = N._vp(a.data_ptr() + 4 * C * HW)
    S = int(lib.ebfi_scale_residual_cat_backward_slices())

    def src_bwd(gc):
        def fill(bk, i):
            ga16 = c16.empty(B, 2 * C, H, W, "cuda")
            gx, p0, p1 = torch.empty(B, C, H, W, device="cuda"), torch.empty(S, B, C, device="cuda"), torch.empty(S, B, C, device="cuda")
            N.check(lib.ebfi_scale_residual_cat_backward_c16(N.ptr(gc), N.ptr(a), N.ptr(s0), a1p, N.ptr(s1), N.ptr(ga16), bk.ptr(i), N.ptr(gx),
                                                             N.ptr(p0), N.ptr(p1), B, C, H, W, 2 * C * HW, 0.01, st), "bwd_c16")
        return fill
    gc = torch.randn(B, 2 * C, H, W).cuda() * 1e-2
    gbad = gc.clone()
    gbad[0, 100, 3, 7] = float("nan")
    assert guard_after(src_bwd(gc)) == 0 and guard_after(src_bwd(gbad)) == 1
    # the forward epilogue's side image (store_out_tile): ONE NaN in the convolution's INPUT must come out of the kernel -- in the
    # fp32 output, in the fp16 image and in the slot's |max| (-> guard).  Round 5 planted the NaN in the bias instead: a NaN in the
    # input did not come out of the wave-specialised kernel.  Cause (round 6, tools/mfma_nan_probe.hip): its matrix-issuing waves
    # ran with MODE.FP16_OVFL set for the epilogue's fp16 stores, and under that bit the matrix cores drop non-finite operands;
    # the bit is now set around the epilogue only (csrc/c16.hpp).
    w, b, bank, book0, site = _banked(64, 64)
    seen = {}

    def fwd_img(xin):
        def fill(bk, i):
            out, img = torch.empty(2, 64, 16, 64, device="cuda"), c16.empty(2, 64, 16, 64, "cuda")
            N.check(lib.ebfi_conv2d_packed_x3_c16(N.ptr(xin), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(out), 2, 64, 16, 64, 64,
                                                  3, 1, 1, 1, 0.01, N.ptr(None), N.ptr(None), 0, 0.0, N.ptr(img), bk.ptr(i), 0, st), "x3_c16")
            torch.cuda.synchronize()
            seen.update(out_nan=int(torch.isnan(out).sum()), img_nan=int(torch.isnan(img).sum()),
                        amax_bits=hex(int(bk.slots[f16scale.SLOT_STRIDE * i + f16scale.SLOT_AMAX].view(torch.int32).item()) & 0xffffffff))
        return fill
    assert guard_after(fwd_img(x)) == 0 and seen["out_nan"] == 0 and seen["img_nan"] == 0, seen
    assert guard_after(fwd_img(bad)) == 1, seen
    assert seen["out_nan"] == 64 * 9 and seen["img_nan"] == 64 * 9, seen      # the 3x3 neighbourhood of (5, 11) in every output channel


@pytest.mark.parametrize("special", [float("nan"), float("inf"), float("-inf")])
@pytest.mark.parametrize("B,Cin,Cout,H,W,form", [(2, 64, 64, 16, 64, "image"), (4, 64, 64, 128, 256, "plain"), (4, 64, 128, 128, 128, "plain"),
                                                 (2, 64, 64, 8, 8, "plain")])
def test_non_finite_input_comes_out_where_the_fp32_kernel_puts_it(B, Cin, Cout, H, W, form, special):
    """NaN and +-Inf in the INPUT of the split-precision forward come out exactly at the output positions where the exact fp32
    kernel (conv_fwd_f32) has non-finite values -- in the wave-specialised kernel (large shapes; pinned at any size by the
    image-writing form), in the double-buffered one (small shapes) -- and nowhere else.  (Kind: a NaN stays NaN; an Inf becomes
    NaN in split precision, because its low part Inf - bf16(Inf) is NaN -- non-finite either way, reference ConvLayer
    submodules.py:159-200 on fp32 propagates Inf.)"""
    from ebfi_amd import c16, f16scale
    torch.manual_seed(Cout + H)
    w, b, bank, book, site = _banked(Cin, Cout)
    lib, st = N.lib(), N.stream_ptr(torch.device("cuda"))
    x = torch.randn(B, Cin, H, W).cuda()
    for pos in ((B - 1, 37, 5, min(11, W - 1)), (0, 0, 0, 0), (B - 1, Cin - 1, H - 1, W - 1)):
        xb = x.clone()
        xb[pos] = special
        ref = torch.zeros(B, Cout, H, W, device="cuda")
        N.check(lib.ebfi_conv2d_forward(N.ptr(xb), N.ptr(w.detach()), N.ptr(b.detach()), N.ptr(ref), B, Cin, H, W, Cout, 3, 1, 1, 1, 0.01,
                                        N.EBFI_F32, st), "f32")
        out = torch.zeros_like(ref)
        N.prof_reset()
        N.prof_enable(True)
        if form == "image":
            img = c16.empty(B, Cout, H, W, "cuda")
            N.check(lib.ebfi_conv2d_packed_x3_c16(N.ptr(xb), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(out), B, Cin, H, W, Cout,
                                                  3, 1, 1, 1, 0.01, N.ptr(None), N.ptr(None), 0, 0.0, N.ptr(img), book.ptr(book.slot("t")), 0, st), "x3_c16")
        else:
            N.check(lib.ebfi_conv2d_packed_x3(N.ptr(xb), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(out), B, Cin, H, W, Cout,
                                              3, 1, 1, 1, 0.01, N.ptr(None), N.ptr(None), 0, 0.0, st), "x3")
        torch.cuda.synchronize()
        N.prof_enable(False)
        ran = {k.split("/")[0] for k, v in N.prof_collect().items() if v[0] > 0}
        assert ("conv_fwd_bf16x3_ws" in ran) == (form == "image" or H * W >= 16384), ran        # (which kernel the shape selected)
        bad_ref, bad_out = ~torch.isfinite(ref), ~torch.isfinite(out)
        assert int(bad_ref.sum()) >= 4 * Cout and torch.equal(bad_ref, bad_out), (pos, int(bad_ref.sum()), int(bad_out.sum()))
        if special != special:
            assert torch.equal(torch.isnan(ref), torch.isnan(out))
        if form == "image":
            # (image layout [B][C/16][H][2][W][8]: the non-finite pixels of the image are those of the output)
            bad_img = ~torch.isfinite(c16.from_c16(img, 1.0))
            assert torch.equal(bad_img, bad_out)


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 16, 64), (1, 64, 13, 36), (3, 64, 40, 132)])
def test_residual_control_tail_in_the_convolution_epilogue(B, C, H, W):
    """ebfi_conv2d_packed_x3_rc (round 5: scale + residual + concat of a ResidualControl round folded into the epilogue of its
    grouped second-layer convolution) against the two launches it replaces -- the grouped convolution, then
    ebfi_scale_residual_cat_forward_c16 -- bit for bit on the fp32 output and its image; the extra image of the activation output
    `a` equals fp16(a * scale); the backward stage reading that image agrees with the one reading fp32 `a` to fp16 rounding."""
    from ebfi_amd import c16, f16scale
    torch.manual_seed(B * 7 + W)
    w, b, bank, book, site = _banked(C, 2 * C, groups=2)
    HW = H * W
    ya = torch.randn(B, 2 * C, H, W).cuda()
    x = torch.randn(B, C, H, W).cuda()
    s_ex, s_t = torch.randn(B, C).cuda(), torch.randn(B, C).cuda()
    lib, st = N.lib(), N.stream_ptr(x.device)
    sc_, sa_ = book.slot("c"), book.slot("a")
    book.slots[f16scale.SLOT_STRIDE * sc_] = 4.0
    book.slots[f16scale.SLOT_STRIDE * sa_] = 8.0
    # reference: convolution, then the fused stage
    a = torch.empty(B, 2 * C, H, W, device="cuda")
    N.check(lib.ebfi_conv2d_packed_x3(N.ptr(ya), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(a), B, C, H, W, 2 * C, 3, 1, 2, 1,
                                      0.01, N.ptr(None), N.ptr(None), 0, 0.0, st), "x3")
    c_ref, c16_ref = torch.empty_like(a), c16.empty(B, 2 * C, H, W, "cuda")
    a1p = N._vp(a.data_ptr() + 4 * C * HW)
    N.check(lib.ebfi_scale_residual_cat_forward_c16(N.ptr(a), N.ptr(s_ex), a1p, N.ptr(s_t), N.ptr(x), N.ptr(c_ref), N.ptr(c16_ref),
                                                    book.ptr(sc_), B, C, H, W, 2 * C * HW, st), "src_fwd_c16")
    amax_c = book.amax(sc_)
    book.slots[f16scale.SLOT_STRIDE * sc_ + f16scale.SLOT_AMAX] = 0.0
    # the fused epilogue
    s_cat = torch.cat([s_ex, s_t], 1).contiguous()
    c_out = torch.full_like(a, float("nan"))
    c16_out, a16 = c16.empty(B, 2 * C, H, W, "cuda").fill_(7.0), c16.empty(B, 2 * C, H, W, "cuda").fill_(7.0)
    N.check(lib.ebfi_conv2d_packed_x3_rc(N.ptr(ya), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(c_out), B, C, H, W, 2 * C, 2,
                                         0.01, N.ptr(s_cat), N.ptr(x), C, N.ptr(a16), book.ptr(sa_), N.ptr(c16_out), book.ptr(sc_), st), "x3_rc")
    # (a * s + x: the stage multiplies and adds in separate roundings or one fma depending on the compiler's contraction -- allow 1 ulp)
    assert torch.allclose(c_out, c_ref, rtol=3e-7, atol=1e-7)
    assert ((c16_out.float() - c16_ref.float()).abs().max() <= 4.0 * 2e-3 * c_ref.abs().max()).item()
    assert torch.equal(c16_out, _ref_image(c_out, 4.0))
    assert torch.equal(a16, _ref_image(a, 8.0))
    assert book.amax(sa_) == a.abs().max().item() and abs(book.amax(sc_) - amax_c) <= 1e-6 * amax_c
    # inference form (no images): the same fp32 result, nothing else written; a partial set of image arguments is refused
    c_inf = torch.full_like(a, float("nan"))
    amax_before = (book.amax(sa_), book.amax(sc_))
    N.check(lib.ebfi_conv2d_packed_x3_rc(N.ptr(ya), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(c_inf), B, C, H, W, 2 * C, 2,
                                         0.01, N.ptr(s_cat), N.ptr(x), C, N.ptr(None), N.ptr(None), N.ptr(None), N.ptr(None), st), "x3_rc (inference)")
    assert torch.equal(c_inf, c_out) and (book.amax(sa_), book.amax(sc_)) == amax_before
    assert lib.ebfi_conv2d_packed_x3_rc(N.ptr(ya), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(c_inf), B, C, H, W, 2 * C, 2,
                                        0.01, N.ptr(s_cat), N.ptr(x), C, N.ptr(a16), book.ptr(sa_), N.ptr(None), N.ptr(None), st) == -1      # EBFI_ERR_ARG
    # backward stage on the image of a
    gc = torch.randn(B, 2 * C, H, W).cuda() * 1e-2
    sg = book.slot("g")
    book.slots[f16scale.SLOT_STRIDE * sg] = 512.0
    S = int(lib.ebfi_scale_residual_cat_backward_slices())
    outs = []
    for mode in (0, 1):
        ga16 = c16.empty(B, 2 * C, H, W, "cuda")
        gx, p0, p1 = torch.empty(B, C, H, W, device="cuda"), torch.empty(S, B, C, device="cuda"), torch.empty(S, B, C, device="cuda")
        if mode == 0:
            N.check(lib.ebfi_scale_residual_cat_backward_c16(N.ptr(gc), N.ptr(a), N.ptr(s_ex), a1p, N.ptr(s_t), N.ptr(ga16), book.ptr(sg),
                                                             N.ptr(gx), N.ptr(p0), N.ptr(p1), B, C, H, W, 2 * C * HW, 0.01, st), "bwd_c16")
        else:
            N.check(lib.ebfi_scale_residual_cat_backward_c16a(N.ptr(gc), N.ptr(a16), book.ptr(sa_), N.ptr(s_ex), N.ptr(s_t), N.ptr(ga16),
                                                              book.ptr(sg), N.ptr(gx), N.ptr(p0), N.ptr(p1), B, C, H, W, 0.01, st), "bwd_c16a")
        outs.append((ga16.clone(), gx.clone(), p0.sum(0), p1.sum(0)))
    (g0, x0, q0, r0), (g1, x1, q1, r1) = outs
    assert torch.equal(x0, x1)
    # the mask only needs the SIGN of a (an element below the image's denormal range reads as 0: 1e-8 of |max|)
    assert (g0 != g1).float().mean().item() < 1e-4
    assert ((q0 - q1).abs().max() / q0.abs().max()).item() < 1e-3 and ((r0 - r1).abs().max() / r0.abs().max()).item() < 1e-3
