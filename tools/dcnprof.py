#!/usr/bin/env python3
"""Phase times of dcn_bwd_data_f32 for one workgroup (development aid; needs a library built with -DDCN_STAMPS:
EBFI_EXTRA_FLAGS=-DDCN_STAMPS, selected with EBFI_LIB_PATH)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd import _native as N  # noqa: E402
from ebfi_amd.dcn import dcn_v2_backward  # noqa: E402

B, C, h, w, dg = 8, 64, 128, 128, 8
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(B, C, h, w, device=dev)
off = torch.randn(B, dg * 18, h, w, device=dev) * float(os.environ.get("OFFSCALE", "2"))
msk = torch.sigmoid(torch.randn(B, dg * 9, h, w, device=dev))
wt = torch.randn(C, C, 3, 3, device=dev) / 24
bias = torch.randn(C, device=dev)
g = torch.randn(B, C, h, w, device=dev)
cfg = ((1, 1), (1, 1), (1, 1), dg)
lib = N.lib()
f = getattr(lib, "ebfi_dcn_debug_stamps", None)
for _ in range(3):
    dcn_v2_backward(x, wt, bias, off, msk, g, *cfg)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
if f is not None:
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    f(None, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N.prof_reset()
N.prof_enable(True)
e0.record()
dcn_v2_backward(x, wt, bias, off, msk, g, *cfg)
e1.record()
torch.cuda.synchronize()
N.prof_enable(False)
print({k: (v[0], round(v[1], 4)) for k, v in N.prof_collect().items() if v[0]})
if f is not None:
    f(buf, 0)
    names = ["zero box", "gout issue + weight slice", "mfma", "colgrad -> LDS", "walk: loop exit", "barrier before flush", "flush",
             "walk: loop head", "walk: offsets arrive", "walk: corners arrive", "walk: channels (math + LDS atomics)", "walk: stores"]
    tot = sum(buf[:12])
    for n, v in zip(names, buf[:12]):
        print("%-36s %9.1f us  %5.1f %%" % (n, v / 100.0, 100.0 * v / max(tot, 1)))
    print("total %.1f us (s_memtime at 100 MHz)" % (tot / 100.0))
