"""Which entry point ignores its stream?  On a side stream a slow producer (20 ms of GEMMs) feeds each launch: a launch that
lands on another stream runs before its input exists."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd import _native as N, c16, f16scale, weightbank
lib = N.lib()
B, Cin, H, W, Cout = 2, 64, 64, 64, 128
w = torch.nn.Parameter((torch.randn(Cout, Cin, 3, 3) / 24).cuda()); b = torch.nn.Parameter(torch.randn(Cout).cuda() * 0.1)
bank = weightbank.WeightBank([w, b]); site = bank.register(w, b, "id")
book = f16scale.ScaleBook("cuda"); bank.attach_scale_book(book); bank.refresh()
g0 = torch.randn(B, Cout, H, W).cuda() * 1e-2
x0 = torch.randn(B, Cin, H, W).cuda()
sg, sx, so = book.slot("g"), book.slot("x"), book.slot("o")
book.calibrate(sg, g0); book.calibrate(sx, x0)
big = torch.randn(6144, 6144, device="cuda") / 78.0
torch.cuda.synchronize()
side = torch.cuda.Stream()
def slow_zero():
    t = big
    for _ in range(12):
        t = t @ big
    return (t.sum() * 0.0)          # a device scalar 0 that exists only after ~20 ms of work
def run(on_side, which):
    with (torch.cuda.stream(side) if on_side else torch.cuda.stream(torch.cuda.current_stream())):
        st = N.stream_ptr(g0.device)
        z = slow_zero()
        if which == "to_c16":
            g = g0 + z
            r = c16.to_c16(g, book.ptr(sg)).float().abs().sum()
        elif which == "dgrad":
            g16 = c16.to_c16(g0, book.ptr(sg))
            torch.cuda.current_stream().synchronize()
            z = slow_zero()
            g16 = g16 + z.half()
            out = torch.zeros(B, Cin, H, W, device="cuda")
            N.check(lib.ebfi_conv2d_packed_f16_c16(N.ptr(g16), 1, site.tr16_ptr(), site.tr16_bytes, None, N.ptr(out), B, Cout, H, W, Cin, 3, 1, 1, 0, 0.0,
                                                   None, None, 0, 0.0, book.ptr(sg), site.w_slot_ptr(), None, None, st), "dgrad")
            r = out.abs().sum()
        elif which == "wgrad":
            g16, x16 = c16.to_c16(g0, book.ptr(sg)), c16.to_c16(x0, book.ptr(sx))
            torch.cuda.current_stream().synchronize()
            z = slow_zero()
            g16 = g16 + z.half()
            need = int(lib.ebfi_conv2d_backward_weight_workspace(B, Cin, H, W, Cout, 3, 1, 1, 0))
            ws = torch.empty(need, dtype=torch.uint8, device="cuda")
            gw, gb = torch.zeros(Cout, Cin, 3, 3, device="cuda"), torch.zeros(Cout, device="cuda")
            N.check(lib.ebfi_conv2d_backward_weight_f16c(N.ptr(x16), N.ptr(g16), 0, N.ptr(gw), N.ptr(gb), B, Cin, H, W, Cout, 1, book.ptr(sx), book.ptr(sg), N.ptr(ws), need, st), "wgrad")
            r = gw.abs().sum()
        elif which == "srcb":
            C = 64
            a = torch.randn(B, 2 * C, H, W, device="cuda"); s0 = torch.randn(B, C, device="cuda"); s1 = torch.randn(B, C, device="cuda")
            torch.cuda.current_stream().synchronize()
            gc = torch.randn(B, 2 * C, H, W, device="cuda") + slow_zero()
            ga16 = c16.empty(B, 2 * C, H, W, "cuda"); ga16.zero_()
            gx = torch.zeros(B, C, H, W, device="cuda")
            S = int(lib.ebfi_scale_residual_cat_backward_slices())
            p0, p1 = torch.zeros(S, B, C, device="cuda"), torch.zeros(S, B, C, device="cuda")
            N.check(lib.ebfi_scale_residual_cat_backward_c16(N.ptr(gc), N.ptr(a), N.ptr(s0), N._vp(a.data_ptr() + 4 * C * H * W), N.ptr(s1), N.ptr(ga16), book.ptr(so), N.ptr(gx),
                                                             N.ptr(p0), N.ptr(p1), B, C, H, W, 2 * C * H * W, 0.01, st), "srcb")
            r = gx.abs().sum()
        elif which == "fwd":
            x = x0 + slow_zero()
            y, y16 = torch.zeros(B, Cout, H, W, device="cuda"), c16.empty(B, Cout, H, W, "cuda")
            N.check(lib.ebfi_conv2d_packed_x3_c16(N.ptr(x), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(y), B, Cin, H, W, Cout, 3, 1, 1, 1, 0.01,
                                                  None, None, 0, 0.0, N.ptr(y16), book.ptr(so), 0, st), "fwd")
            r = y.abs().sum()
    torch.cuda.synchronize()
    return float(r)
torch.manual_seed(0)
for which in ("to_c16", "dgrad", "wgrad", "srcb", "fwd"):
    torch.manual_seed(1); a = run(False, which)
    torch.manual_seed(1); s = run(True, which)
    print("%-8s default %.6g  side %.6g  %s" % (which, a, s, "OK" if a == s else "MISMATCH"))
