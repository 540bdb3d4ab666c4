"""Each c16 entry point on a side stream: the result must be visible to a torch op queued right behind it on that stream."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd import _native as N, c16, f16scale, weightbank
lib = N.lib()
B, Cin, H, W, Cout = (2, 64, 64, 64, 128) if len(sys.argv) > 1 else (8, 64, 128, 128, 128)
GS = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-2
w = torch.nn.Parameter((torch.randn(Cout, Cin, 3, 3) / 24).cuda()); b = torch.nn.Parameter(torch.randn(Cout).cuda() * 0.1)
bank = weightbank.WeightBank([w, b]); site = bank.register(w, b, "id")
book = f16scale.ScaleBook("cuda"); bank.attach_scale_book(book); bank.refresh()
g = torch.randn(B, Cout, H, W).cuda() * GS
x = torch.randn(B, Cin, H, W).cuda()
sg, sx, so = book.slot("g"), book.slot("x"), book.slot("o")
book.calibrate(sg, g); book.calibrate(sx, x)
torch.cuda.synchronize()
side = torch.cuda.Stream()
def run(on_side):
    ctx = torch.cuda.stream(side) if on_side else torch.cuda.stream(torch.cuda.current_stream())
    res = {}
    with ctx:
        st = N.stream_ptr(g.device)
        g16 = c16.to_c16(g, book.ptr(sg))
        res["to_c16"] = g16.float().abs().sum()
        out = torch.full((B, Cin, H, W), 7.0, device="cuda")
        rc = lib.ebfi_conv2d_packed_f16_c16(N.ptr(g16), 1, site.tr16_ptr(), site.tr16_bytes, None, N.ptr(out), B, Cout, H, W, Cin, 3, 1, 1, 0, 0.0,
                                            None, None, 0, 0.0, book.ptr(sg), site.w_slot_ptr(), None, None, st)
        N.check(rc, "dgrad")
        res["dgrad img->f32"] = out.abs().sum()
        o16 = c16.empty(B, Cin, H, W, "cuda"); o16.fill_(3.0)
        rc = lib.ebfi_conv2d_packed_f16_c16(N.ptr(g16), 1, site.tr16_ptr(), site.tr16_bytes, None, None, B, Cout, H, W, Cin, 3, 1, 1, 0, 0.0,
                                            None, None, 0, 0.0, book.ptr(sg), site.w_slot_ptr(), N.ptr(o16), book.ptr(so), st)
        N.check(rc, "dgrad img")
        res["dgrad img->img"] = o16.float().abs().sum()
        x16 = c16.to_c16(x, book.ptr(sx))
        need = int(lib.ebfi_conv2d_backward_weight_workspace(B, Cin, H, W, Cout, 3, 1, 1, 0))
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        gw, gb = torch.full((Cout, Cin, 3, 3), 5.0, device="cuda"), torch.full((Cout,), 5.0, device="cuda")
        rc = lib.ebfi_conv2d_backward_weight_f16c(N.ptr(x16), N.ptr(g16), 0, N.ptr(gw), N.ptr(gb), B, Cin, H, W, Cout, 1, book.ptr(sx), book.ptr(sg), N.ptr(ws), need, st)
        N.check(rc, "wgrad")
        res["wgrad"] = gw.abs().sum() + gb.abs().sum()
        y, y16 = torch.full((B, Cout, H, W), 2.0, device="cuda"), c16.empty(B, Cout, H, W, "cuda")
        rc = lib.ebfi_conv2d_packed_x3_c16(N.ptr(x), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(y), B, Cin, H, W, Cout, 3, 1, 1, 1, 0.01,
                                           None, None, 0, 0.0, N.ptr(y16), book.ptr(so), 0, st)
        N.check(rc, "fwd")
        res["fwd+img"] = y.abs().sum() + y16.float().abs().sum()
    torch.cuda.synchronize()
    return {k: float(v) for k, v in res.items()}
a = run(False); b_ = run(True); c_ = run(True)
for k in a:
    print("%-16s default %.6g   side %.6g   side again %.6g   %s" % (k, a[k], b_[k], c_[k], "OK" if a[k] == b_[k] == c_[k] else "MISMATCH"))

import threading
out = {}
def worker():
    out["t"] = run(True)
th = threading.Thread(target=worker); th.start(); th.join()
for k in a:
    print("%-16s default %.6g   side-in-thread %.6g   %s" % (k, a[k], out["t"][k], "OK" if a[k] == out["t"][k] else "MISMATCH"))
