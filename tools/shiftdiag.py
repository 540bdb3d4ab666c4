#!/usr/bin/env python3
"""Per-launch time of the thin-layer weight gradients (csrc/conv2d_shift.inc.hpp) at the model's shapes (development aid; with
EBFI_DEV=1 EBFI_LIB_PATH=<diagnostic build> it times a variant with the products or the thick loads compiled out)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd import _native as N  # noqa: E402
from ebfi_amd import conv  # noqa: E402

conv.set_compute_dtype("bf16x3")
for (B, Cin, H, W, Cout, k, p, act) in ((8, 64, 256, 256, 3, 3, 1, 2), (8, 64, 256, 256, 1, 3, 1, 0), (8, 4, 256, 256, 64, 3, 1, 1), (8, 16, 262, 262, 3, 7, 0, 0)):
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, W, device="cuda")
    w = (torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5).requires_grad_()
    b = torch.zeros(Cout, device="cuda", requires_grad=True)
    out = conv.conv_bias_act(x, w, b, 1, p, act, 0.01)
    g = torch.randn_like(out)
    for it in range(6):
        if it == 1:
            torch.cuda.synchronize()
            N.prof_reset()
            N.prof_enable(True)
        w.grad = None
        out = conv.conv_bias_act(x, w, b, 1, p, act, 0.01)
        out.backward(g)
    torch.cuda.synchronize()
    N.prof_enable(False)
    prof = N.prof_collect()
    print((B, Cin, H, W, Cout, k), {n: round(v[1] / v[0] * 1e3, 1) for n, v in prof.items() if ("wgrad" in n or "conv7" in n) and v[0]})
