#!/usr/bin/env python3
"""Device time of every conv launch of one eager training step by entry point and geometry (development aid):
wraps the C-ABI conv entry points with event pairs.  usage (GPU box): python tools/convshapes.py"""
import collections
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd import _native as N  # noqa: E402
from ebfi_amd.engine import Engine, synthetic_batch  # noqa: E402

eng = Engine(device="cuda", seed=1, precision="bf16x3", graph=False)
batch = synthetic_batch(8, 256, 256)
for _ in range(2):
    eng.train_step(*batch)
torch.cuda.synchronize()
lib = N.lib()
records = []
GEO = {"ebfi_conv2d_forward_bf16x3": slice(4, 12), "ebfi_conv2d_backward_data_bf16x3": slice(4, 12),
       "ebfi_conv2d_packed_x3": slice(5, 13), "ebfi_conv2d_backward_weight_ex": slice(6, 14),
       "ebfi_conv2d_backward_weight_x3g": slice(4, 12), "ebfi_conv2d_forward": slice(4, 12), "ebfi_conv2d_backward_data": slice(4, 12),
       # fp16 backward of the generic (non-ResidualControl) layers: (x, gout, y, gw, gb, gpre, gpre_is_c16, B, Cin, H, W, M, ks, pad, ..)
       "ebfi_conv2d_backward_weight_f16g_ex": slice(7, 15), "ebfi_conv2d_packed_f16": slice(5, 13),
       "ebfi_conv2d_packed_f16_c16": slice(6, 14), "ebfi_conv2d_thin_forward": slice(4, 12),
       # (B, Cin, H, W, Cout) only: printed with k = 3
       "ebfi_conv2d_packed_x3_shuffled": slice(5, 10), "ebfi_conv2d_packed_f16_shuffled": slice(6, 11),
       "ebfi_conv2d_packed_x3_c16": slice(5, 13), "ebfi_conv2d_packed_x3_rc": slice(5, 10),
       "ebfi_conv2d_backward_weight_f16c": slice(5, 10)}


def wrap(name):
    fn = getattr(lib, name)

    def f(*a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*a)
        e1.record()
        geo = tuple(int(v.value) if hasattr(v, "value") else int(v) for v in a[GEO[name]])
        records.append((name, geo, e0, e1))
        return rc
    setattr(lib, name, f)


for n in GEO:
    wrap(n)
eng.train_step(*batch)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for name, geo, e0, e1 in records:
    agg[(name.replace("ebfi_conv2d_", ""), geo)][0] += 1
    agg[(name.replace("ebfi_conv2d_", ""), geo)][1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in agg.values())
print("total %.1f us in %d launches" % (tot, len(records)))
for (name, geo), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    B, C, H, W, Co, k = (tuple(geo) + (3,))[:6]
    fl = 2.0 * B * H * W * C * Co * k * k * n
    print("%-26s x%2d %8.1f us (%5.1f each) %5.1f TF/s  %s" % (name, n, us, us / n, fl / us / 1e6, geo))
