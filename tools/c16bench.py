#!/usr/bin/env python3
"""Device time of the fp16 backward kernels and of the image writers, fp32 operands vs c16 images (development tool).
   python tools/c16bench.py            # B=8, 128x128, ResidualControl layer shapes"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd import _native as N, c16, f16scale, weightbank  # noqa: E402

B, H, W, C = 8, 128, 128, 64
lib = N.lib()
dev = torch.device("cuda")


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    N.prof_reset()
    N.prof_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    N.prof_enable(False)
    return {k: 1e3 * v[1] / v[0] for k, v in N.prof_collect().items() if v[0]}


def layer(cin, cout, groups=1):
    w = torch.nn.Parameter((torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5).cuda())
    b = torch.nn.Parameter((torch.randn(cout) * 0.1).cuda())
    bank = weightbank.WeightBank([w, b])
    site = bank.register(w, b, "id", groups=groups)
    book = f16scale.ScaleBook("cuda")
    bank.attach_scale_book(book)
    bank.refresh()
    return site, book, bank


st = N.stream_ptr(dev)
print("shape B=%d %dx%d" % (B, H, W))
for name, cin_g, cout, groups in (("A 64->128", 64, 128, 1), ("B 2x(64->64)", 64, 128, 2), ("5 128->64", 128, 64, 1)):
    site, book, bank = layer(cin_g, cout, groups)
    cin = cin_g * groups
    x = torch.randn(B, cin, H, W, device=dev)
    g = torch.randn(B, cout, H, W, device=dev) * 1e-2
    sx, sg, so = book.slot("x"), book.slot("g"), book.slot("o")
    book.calibrate(sx, x); book.calibrate(sg, g)
    x16, g16 = c16.to_c16(x, book.ptr(sx)), c16.to_c16(g, book.ptr(sg))
    need = int(lib.ebfi_conv2d_backward_weight_workspace(B, cin_g, H, W, cout, 3, 1, 1, 0))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    gw, gb = torch.empty(cout, cin_g, 3, 3, device=dev), torch.empty(cout, device=dev)
    t = timed(lambda: lib.ebfi_conv2d_backward_weight_f16g(N.ptr(x), N.ptr(g), None, N.ptr(gw), N.ptr(gb), None, B, cin_g, H, W, cout, 3, 1,
                                                           groups, 0, 0.0, book.ptr(sx), book.ptr(sg), N.ptr(ws), need, st))
    t2 = timed(lambda: lib.ebfi_conv2d_backward_weight_f16c(N.ptr(x16), N.ptr(g16), 0, N.ptr(gw), N.ptr(gb), B, cin_g, H, W, cout, groups,
                                                            book.ptr(sx), book.ptr(sg), N.ptr(ws), need, st))
    print("%-14s wgrad fp32-in %6.1f us   images %6.1f us   (reduce %4.1f)" % (name, t["conv_wgrad_f16_tr/f32"], t2["conv_wgrad_f16_tr/img"], t2["conv_wgrad_reduce_f32"]))
    # data gradient: cout-channel gradient -> cin channels
    gin, gin16 = torch.empty(B, cin, H, W, device=dev), c16.empty(B, cin, H, W, dev)
    mask = torch.randn(B, cin, H, W, device=dev)

    def dg(inp, is16, out, out16, m=None):
        return lib.ebfi_conv2d_packed_f16_c16(N.ptr(inp), is16, site.tr16_ptr(), site.tr16_bytes, None, N.ptr(out), B, cout // groups, H, W,
                                              cin, 3, 1, groups, 0, 0.0, None, N.ptr(m), 1 if m is not None else 0, 0.01,
                                              book.ptr(sg), site.w_slot_ptr(), N.ptr(out16), book.ptr(so) if out16 is not None else None, 0, 0, st)
    r = []
    for label, args in (("fp32->fp32", (g, 0, gin, None)), ("img->fp32", (g16, 1, gin, None)), ("img->img", (g16, 1, None, gin16)),
                        ("img->img+mask", (g16, 1, None, gin16, mask)), ("fp32->fp32+mask", (g, 0, gin, None, mask))):
        r.append("%s %5.1f" % (label, [v for k, v in timed(lambda: dg(*args)).items() if k.startswith("conv_fwd_f16_ws")][0]))
    print("%-14s dgrad  %s" % (name, "   ".join(r)))
    # forward with / without the side image
    out, out16 = torch.empty(B, cout, H, W, device=dev), c16.empty(B, cout, H, W, dev)

    def fw(o16):
        return lib.ebfi_conv2d_packed_x3_c16(N.ptr(x), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(out), B, cin_g, H, W, cout, 3, 1,
                                             groups, 1, 0.01, None, None, 0, 0.0, N.ptr(o16), book.ptr(so) if o16 is not None else None, 0, st)
    ta, tb = timed(lambda: fw(None)), timed(lambda: fw(out16))
    print("%-14s fwd    plain %5.1f   with image %5.1f" % (name, list(ta.values())[0], list(tb.values())[0]))

a = torch.randn(B, 2 * C, H, W, device=dev)
s0, s1, x = torch.randn(B, C, device=dev), torch.randn(B, C, device=dev), torch.randn(B, C, H, W, device=dev)
book = f16scale.ScaleBook("cuda")
so = book.slot("o")
HW = H * W
a1p = N._vp(a.data_ptr() + 4 * C * HW)
out, out16 = torch.empty(B, 2 * C, H, W, device=dev), c16.empty(B, 2 * C, H, W, dev)
t1 = timed(lambda: lib.ebfi_scale_residual_cat_forward_ex(N.ptr(a), N.ptr(s0), a1p, N.ptr(s1), N.ptr(x), N.ptr(out), B, C, HW, 2 * C * HW, st))
t2 = timed(lambda: lib.ebfi_scale_residual_cat_forward_c16(N.ptr(a), N.ptr(s0), a1p, N.ptr(s1), N.ptr(x), N.ptr(out), N.ptr(out16), book.ptr(so), B, C, H, W, 2 * C * HW, st))
print("src fwd  plain %5.1f  with image %5.1f" % (list(t1.values())[0], list(t2.values())[0]))
gc = torch.randn(B, 2 * C, H, W, device=dev)
ga = torch.empty(B, 2 * C, H, W, device=dev)
gx, gs0, gs1 = torch.empty(B, C, H, W, device=dev), torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)
S = int(lib.ebfi_scale_residual_cat_backward_slices())
p0, p1 = torch.empty(S, B, C, device=dev), torch.empty(S, B, C, device=dev)
t1 = timed(lambda: lib.ebfi_scale_residual_cat_backward_ex(N.ptr(gc), N.ptr(a), N.ptr(s0), a1p, N.ptr(s1), N.ptr(ga), N._vp(ga.data_ptr() + 4 * C * HW),
                                                           N.ptr(gx), N.ptr(gs0), N.ptr(gs1), B, C, HW, 2 * C * HW, 2 * C * HW, 1, 0.01, st))
t2 = timed(lambda: lib.ebfi_scale_residual_cat_backward_c16(N.ptr(gc), N.ptr(a), N.ptr(s0), a1p, N.ptr(s1), N.ptr(out16), book.ptr(so), N.ptr(gx),
                                                            N.ptr(p0), N.ptr(p1), B, C, H, W, 2 * C * HW, 0.01, st))
print("src bwd  plain %5.1f  images %5.1f" % (list(t1.values())[0], list(t2.values())[0]))
t3 = timed(lambda: c16.to_c16(x, book.ptr(so)))
print("to_c16 [8,64,128,128] %5.1f us" % list(t3.values())[0])

# ---- the same image-writing launches with the scale book's finish() between iterations (as inside a training step: the running
# maximum restarts from its floor every step)
print("with book.finish() between iterations (amax restarts from the floor):")
site, book, bank = layer(64, 128, 1)
x = torch.randn(B, 64, H, W, device=dev)
so = book.slot("o")
out, out16 = torch.empty(B, 128, H, W, device=dev), c16.empty(B, 128, H, W, dev)


def fw2(o16, fin):
    lib.ebfi_conv2d_packed_x3_c16(N.ptr(x), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(out), B, 64, H, W, 128, 3, 1,
                                  1, 1, 0.01, None, None, 0, 0.0, N.ptr(o16), book.ptr(so) if o16 is not None else None, 0, st)
    if fin:
        book.finish()


for fin in (False, True):
    t = timed(lambda: fw2(out16, fin))
    print("  fwd 64->128 with image, finish=%s: %s" % (fin, {k: round(v, 1) for k, v in t.items()}))
    print("    slot scale %.4g amax %.4g floor %.4g" % (book.scale(so), book.amax(so), float(book.slots[f16scale.SLOT_STRIDE * so + 1])))

# ---- cold caches: rotate over input / output buffers whose footprint exceeds the 256 MB Infinity Cache
print("rotating over 12 buffer sets (cold caches):")
NB = 12
xs = [torch.randn(B, 64, H, W, device=dev) for _ in range(NB)]
outs = [torch.empty(B, 128, H, W, device=dev) for _ in range(NB)]
o16s = [c16.empty(B, 128, H, W, dev) for _ in range(NB)]
it = [0]


def fw3(img, fin):
    k = it[0] % NB
    it[0] += 1
    lib.ebfi_conv2d_packed_x3_c16(N.ptr(xs[k]), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(outs[k]), B, 64, H, W, 128, 3, 1,
                                  1, 1, 0.01, None, None, 0, 0.0, N.ptr(o16s[k]) if img else None, book.ptr(so) if img else None, 0, st)
    if fin:
        book.finish()


for img in (False, True):
    t = timed(lambda: fw3(img, True), iters=36)
    print("  fwd 64->128 image=%s: %s" % (img, {k: round(v, 1) for k, v in t.items() if k.startswith("conv")}))
