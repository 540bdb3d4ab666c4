#!/usr/bin/env python3
"""Per-shape timing of the backward kernels of one 3x3 layer in both operand formats (development tool, GPU box):
split-precision bf16x3 (three MFMAs per product) vs fp16 single-product with operand scales (csrc/conv2d_f16.inc.hpp).
usage: python tools/f16bench.py [Cin Cout [H W B]] ...   (default: the step's shapes at 128x128, B=8)"""
import os
import sys

os.environ.setdefault("EBFI_DEV", "1")      # this tool flips the library's development switches
import torch

LAST = {}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd import conv, f16scale, weightbank  # noqa: E402


def bench(fn, iters=20):
    """Average DEVICE time (us) of the conv kernels `fn` launches: the library's hipEvent pairs (a Python-level autograd call
    is host-bound below ~100 us)."""
    from ebfi_amd import _native as N
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    N.prof_reset()
    N.prof_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    N.prof_enable(False)
    prof = N.prof_collect()
    global LAST
    LAST = {k: round(v[1] / max(v[0], 1) * 1e3, 1) for k, v in prof.items() if k.startswith("conv_") and v[0] > 0}
    tot = sum(v[1] for k, v in prof.items() if k.startswith("conv_") and "reduce" not in k)
    return tot / iters * 1e3


def run(Cin, Cout, H=128, W=128, B=8, gmag=1.0, sparse=0.0, cold=0, dirty=0, xmag=1.0):
    torch.manual_seed(0)
    w = torch.nn.Parameter((torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5).cuda())
    b = torch.nn.Parameter(torch.zeros(Cout).cuda())
    bank = weightbank.WeightBank([w, b])
    bank.register(w, b, "id")
    book = f16scale.ScaleBook("cuda")
    bank.attach_scale_book(book)
    bank.refresh()
    x = torch.randn(B, Cin, H, W, device="cuda") * xmag
    g = torch.randn(B, Cout, H, W, device="cuda") * gmag
    if sparse > 0:                       # a few large elements, the rest `sparse` times smaller (fp16 subnormals after scaling)
        g = torch.where(torch.rand_like(g) < 1e-4, g, g * sparse)
    conv.set_compute_dtype("bf16x3")
    res = {}
    # cold > 0: rotate over `cold` independent (input, gradient) pairs so that no launch finds its operands in the 256 MB
    # Infinity Cache (what the kernels see inside a training step)
    for mode in ("x3", "f16"):
        sets = []
        for _ in range(max(cold, 1)):
            xd = (x + 0).requires_grad_()
            with bank.active():
                y = conv.conv_bias_act(xd, w, b, 1, 1, 0, 0.0)
            sets.append((xd, y, g + 0))
        state = {"i": 0}

        scratch = [torch.empty(dirty * 1024 * 1024 // 4, device="cuda") for _ in range(2)] if dirty else None

        def nxt():
            state["i"] = (state["i"] + 1) % len(sets)
            if scratch is not None:      # what precedes a backward kernel inside a step: another kernel's freshly written output
                scratch[0].copy_(scratch[1])
            return sets[state["i"]]

        def both():
            xd, y, gg = nxt()
            y.backward(gg, retain_graph=True)

        def only_w():
            xd, y, gg = nxt()
            torch.autograd.grad(y, w, gg, retain_graph=True)

        def only_x():
            xd, y, gg = nxt()
            torch.autograd.grad(y, xd, gg, retain_graph=True)
        if mode == "f16":
            with bank.active(), book.active():
                both()
                res[mode] = (bench(only_x), bench(only_w))
                per = dict(LAST)
        else:
            with bank.active():
                both()
                res[mode] = (bench(only_x), bench(only_w))
    gf = 2.0 * B * H * W * Cin * Cout * 9 / 1e9
    print("cold %d dirty %d MB " % (cold, dirty), end="")
    print("%4d -> %4d  %dx%d B=%d  %7.1f GFLOP | dgrad x3 %8.1f us  f16 %8.1f us (%.0f TF/s) | wgrad x3 %8.1f us  f16 %8.1f us (%.0f TF/s)"
          % (Cin, Cout, H, W, B, gf, res["x3"][0], res["f16"][0], gf / res["f16"][0] * 1e3, res["x3"][1], res["f16"][1],
             gf / res["f16"][1] * 1e3), flush=True)
    print("      per kernel (us):", per, flush=True)


def run_direct():
    """ebfi_conv2d_packed_f16 called directly: plain / with epilogue extras / grouped (the forms rc_fused launches)."""
    from ebfi_amd import _native as N
    lib = N.lib()
    B, C, H, W = 8, 64, 128, 128
    for (cin_g, cout, groups, extra) in ((128, 64, 1, False), (128, 64, 1, True), (64, 128, 2, False), (64, 128, 2, True), (64, 128, 1, True)):
        torch.manual_seed(0)
        w = torch.nn.Parameter((torch.randn(cin_g * groups, cout // groups, 3, 3) / 30).cuda())      # forward weight [M=cin_g*groups.., K=cout/groups]
        bank = weightbank.WeightBank([w])
        site = bank.register(w, None, "id", groups=groups)
        book = f16scale.ScaleBook("cuda")
        bank.attach_scale_book(book)
        bank.refresh()
        g = torch.randn(B, cin_g * groups, H, W, device="cuda")
        out = torch.empty(B, cout, H, W, device="cuda")
        add = torch.randn_like(out) if extra else None
        msk = torch.randn_like(out) if extra else None
        slot = book.operand(("t", "g"), g)

        def call():
            rc = lib.ebfi_conv2d_packed_f16(N.ptr(g), site.tr16_ptr(), site.tr16_bytes, N.ptr(None), N.ptr(out), B, cin_g, H, W, cout, 3, 1,
                                            groups, 0, 0.01, N.ptr(add), N.ptr(msk), 1 if extra else 0, 0.01 if extra else 0.0, slot,
                                            site.w_slot_ptr(), N.stream_ptr(g.device))
            N.check(rc, "packed_f16")
        print("packed_f16 cin_g %d cout %d groups %d extra %d : %.1f us" % (cin_g, cout, groups, extra, bench(call)), flush=True)


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    if a and a[0] == -2:
        run_direct()
    elif a and a[0] == -5:
        for (xm, gm) in ((1.0, 1.0), (1e-19, 1e-22), (1e-30, 1e-30), (1e-36, 1e-36)):
            print("xmag %.0e gmag %.0e" % (xm, gm))
            run(64, 128, cold=4, xmag=xm, gmag=gm)
    elif a and a[0] == -4:
        for shape in ((64, 128), (128, 64)):
            run(*shape, cold=12, dirty=128)
    elif a and a[0] == -3:
        for shape in ((64, 64), (64, 128), (128, 64)):
            run(*shape, cold=12)
    elif a and a[0] == -1:
        for gm, sp in ((1.0, 0.0), (1e-20, 0.0), (1.0, 1e-6), (1.0, 1e-9)):
            run(64, 128, gmag=gm, sparse=sp)
    elif a:
        run(*a)
    else:
        for shape in ((64, 64), (64, 128), (128, 64), (128, 1600), (64, 256)):
            run(*shape)
