#!/usr/bin/env python3
"""Where does a non-finite INPUT element of the split-precision forward go?  (round-5 observation, DESIGN.md: a NaN planted in
the input of conv_fwd_bf16x3_ws did not come out.)  Plants NaN / +Inf / -Inf at several (sample, channel, row, column)
positions of the input of a 64 -> 64 and a 64 -> 128 3x3 layer and compares the set of non-finite OUTPUT elements of

    ebfi_conv2d_forward (exact fp32 matrix cores)          -- the yardstick
    ebfi_conv2d_packed_x3 (wave-specialised kernel)        -- the kernel in question
    ebfi_conv2d_packed_x3 on a shape that selects _db      -- the double-buffered kernel

and prints, per case, the counts and whether the positions agree.  GPU box only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd import _native as N  # noqa: E402
from ebfi_amd import f16scale, weightbank  # noqa: E402


def banked(cin, cout):
    w = torch.nn.Parameter((torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5).cuda())
    b = torch.nn.Parameter((torch.randn(cout) * 0.1).cuda())
    bank = weightbank.WeightBank([w, b])
    site = bank.register(w, b, "id")
    bank.attach_scale_book(f16scale.ScaleBook("cuda"))
    bank.refresh()
    return w, b, site


def run_x3(x, site, cout, act=1):
    B, C, H, W = x.shape
    out = torch.zeros(B, cout, H, W, device="cuda")
    st = N.stream_ptr(x.device)
    N.check(N.lib().ebfi_conv2d_packed_x3(N.ptr(x), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(out), B, C, H, W, cout,
                                          3, 1, 1, act, 0.01, N.ptr(None), N.ptr(None), 0, 0.0, st), "x3")
    torch.cuda.synchronize()
    return out


def run_x3_img(x, site, cout, act=1):
    """the image-writing form (ebfi_conv2d_packed_x3_c16): pins the wave-specialised kernel at any size"""
    from ebfi_amd import c16
    B, C, H, W = x.shape
    out, img = torch.zeros(B, cout, H, W, device="cuda"), c16.empty(B, cout, H, W, "cuda")
    book = f16scale.ScaleBook("cuda")
    i = book.slot("t")
    st = N.stream_ptr(x.device)
    N.check(N.lib().ebfi_conv2d_packed_x3_c16(N.ptr(x), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(out), B, C, H, W, cout,
                                              3, 1, 1, act, 0.01, N.ptr(None), N.ptr(None), 0, 0.0, N.ptr(img), book.ptr(i), 0, st), "x3_c16")
    torch.cuda.synchronize()
    return out, img


def run_f32(x, w, b, act=1):
    B, C, H, W = x.shape
    cout = w.shape[0]
    out = torch.zeros(B, cout, H, W, device="cuda")
    st = N.stream_ptr(x.device)
    N.check(N.lib().ebfi_conv2d_forward(N.ptr(x), N.ptr(w.detach()), N.ptr(b.detach()), N.ptr(out), B, C, H, W, cout, 3, 1, 1, act, 0.01,
                                        N.EBFI_F32, st), "f32")
    torch.cuda.synchronize()
    return out


def main():
    torch.manual_seed(0)
    N.prof_reset()
    N.prof_enable(True)
    for (Bn, cin, cout, H, W) in ((2, 64, 64, 16, 64), (4, 64, 64, 128, 256), (4, 64, 128, 128, 128), (2, 64, 64, 8, 8)):
        w, b, site = banked(cin, cout)
        x0 = torch.randn(Bn, cin, H, W).cuda()
        for val in (float("nan"), float("inf"), float("-inf")):
            for pos in ((1, 37, 5, 11), (0, 0, 0, 0), (1, 63, H - 1, W - 1), (0, 8, 3, 4)):
                pos = tuple(min(p, s - 1) for p, s in zip(pos, x0.shape))
                x = x0.clone()
                x[pos] = val
                a, c = run_f32(x, w, b), run_x3(x, site, cout)
                if W % 4 == 0 and H * W >= 1024:
                    c2, img = run_x3_img(x, site, cout)
                    print("      image-writing form: %4d non-finite in out (%s the plain form), %4d in the fp16 image"
                          % (int((~torch.isfinite(c2)).sum()), "==" if torch.equal(~torch.isfinite(c2), ~torch.isfinite(c)) else "!=",
                             int((~torch.isfinite(img.float())).sum())))
                na, nc = ~torch.isfinite(a), ~torch.isfinite(c)
                same_pos = bool(torch.equal(na, nc))
                same_kind = bool(torch.equal(torch.isnan(a), torch.isnan(c)))
                print("%3d->%3d %2dx%2d  %5s at %-16s  fp32: %4d non-finite (%4d NaN)   x3: %4d non-finite (%4d NaN)   positions %s  kinds %s"
                      % (cin, cout, H, W, val, pos, int(na.sum()), int(torch.isnan(a).sum()), int(nc.sum()), int(torch.isnan(c).sum()),
                         "EQUAL" if same_pos else "DIFFER", "equal" if same_kind else "differ"), flush=True)
        clean = (run_f32(x0, w, b) - run_x3(x0, site, cout)).abs().max().item()
        print("   finite input: max |fp32 - x3| = %.3e" % clean)
    torch.cuda.synchronize()
    N.prof_enable(False)
    print("kernels that ran:", sorted(k for k, v in N.prof_collect().items() if v[0] and not k.startswith("__")))


if __name__ == "__main__":
    main()
