#!/usr/bin/env python3
"""Which PyTorch-native ops (by input shapes) one eager training step still launches (development aid).
usage (GPU box): python tools/opshapes.py [add copy_ fill_ mul cat sum ...]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd.engine import Engine, synthetic_batch  # noqa: E402

want = sys.argv[1:] or ["add", "add_", "copy_", "fill_", "zero_", "mul", "cat", "sum", "contiguous", "clone"]
eng = Engine(device="cuda", seed=1, precision="bf16x3", graph=False)
batch = synthetic_batch(8, 256, 256)
for _ in range(2):
    eng.train_step(*batch)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True) as prof:
    eng.train_step(*batch)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    name = ev.name.replace("aten::", "")
    if (name in want or want == ["all"]) and ev.device_time_total > 0 and not name.startswith("ebfi"):
        key = (name, str(ev.input_shapes)[:110])
        agg[key][0] += 1
        agg[key][1] += ev.device_time_total
for (name, shp), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
    print("%-10s x%3d %8.1f us  %s" % (name, n, us, shp))
