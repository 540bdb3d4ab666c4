import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/ebfi-be_amd')
from ebfi_amd import _native as N, c16, f16scale
lib = N.lib()
for (B, Cin, H, W, Cout, groups) in [(2, 64, 64, 64, 128, 1), (2, 64, 64, 64, 128, 2), (2, 128, 64, 64, 64, 1), (8, 64, 128, 128, 128, 1)]:
    torch.manual_seed(3)
    x = torch.randn(B, groups * Cin, H, W).cuda() * 0.3
    g = torch.randn(B, Cout, H, W).cuda() * 2e-2
    book = f16scale.ScaleBook("cuda")
    sx, sg = book.slot("x"), book.slot("g")
    book.calibrate(sx, x); book.calibrate(sg, g)
    st = N.stream_ptr(x.device)
    need = int(lib.ebfi_conv2d_backward_weight_workspace(B, Cin, H, W, Cout, 3, 1, 1, N.EBFI_F32))
    x16, g16 = c16.to_c16(x, book.ptr(sx)), c16.to_c16(g, book.ptr(sg))
    res = []
    for mode in ("f16g", "f16c"):
        ws = torch.full((need // 4 + 1,), float("nan"), device="cuda")
        gw, gb = torch.empty(Cout, Cin, 3, 3, device="cuda"), torch.empty(Cout, device="cuda")
        if mode == "f16g":
            rc = lib.ebfi_conv2d_backward_weight_f16g(N.ptr(x), N.ptr(g), N.ptr(None), N.ptr(gw), N.ptr(gb), N.ptr(None), B, Cin, H, W, Cout, 3, 1, groups, 0, 0.0, book.ptr(sx), book.ptr(sg), N.ptr(ws), need, st)
        else:
            rc = lib.ebfi_conv2d_backward_weight_f16c(N.ptr(x16), N.ptr(g16), 0, N.ptr(gw), N.ptr(gb), B, Cin, H, W, Cout, groups, book.ptr(sx), book.ptr(sg), N.ptr(ws), need, st)
        N.check(rc, mode)
        torch.cuda.synchronize()
        res.append((gw, gb))
        print((B, Cin, H, W, Cout, groups), mode, "non-finite gw", int((~torch.isfinite(gw)).sum()), "gb", int((~torch.isfinite(gb)).sum()), (~torch.isfinite(gb)).nonzero().flatten()[:8].tolist())
    print("   equal gw:", torch.equal(res[0][0], res[1][0]))
