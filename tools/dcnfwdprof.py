#!/usr/bin/env python3
"""Phase times of dcn_fwd_win for one workgroup (development aid; needs a library built with -DDCN_STAMPS -DEBFI_ABLATE into a
separate file and selected with EBFI_DEV=1 EBFI_LIB_PATH=...)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd import _native as N  # noqa: E402
from ebfi_amd.dcn import dcn_v2_forward  # noqa: E402

B, C, h, w, dg = 8, 64, 128, 128, 8
torch.manual_seed(0)
x = torch.randn(B, C, h, w, device="cuda")
off = torch.randn(B, dg * 18, h, w, device="cuda") * 2
msk = torch.sigmoid(torch.randn(B, dg * 9, h, w, device="cuda"))
wt = torch.randn(C, C, 3, 3, device="cuda") / 24
bias = torch.randn(C, device="cuda")
cfg = ((1, 1), (1, 1), (1, 1), dg)
N.lib()
import ctypes as C_  # noqa: E402
raw = C_.CDLL(N.LIB_PATH)
f = raw.ebfi_dcn_debug_stamps
f.argtypes = [C_.c_void_p, C_.c_int]
for prod in ("fp32", "bf16x3"):
    for _ in range(3):
        dcn_v2_forward(x, wt, bias, off, msk, *cfg, product=prod)
    torch.cuda.synchronize()
    f(None, 1)
    dcn_v2_forward(x, wt, bias, off, msk, *cfg, product=prod)
    torch.cuda.synchronize()
    buf = (C_.c_ulonglong * 16)()
    f(buf, 0)
    v = list(buf)[:8]
    names = ["first prefetch issue", "barrier A (prev MFMA done)", "commit to LDS (waits for the loads)", "barrier B", "sampling", "prefetch issue", "barrier C", "MFMA"]
    tot = sum(v)
    print(prod, "total cycles", tot)
    for n, c in zip(names, v):
        print("   %-40s %8d  %5.1f %%" % (n, c, 100.0 * c / max(tot, 1)))
