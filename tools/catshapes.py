#!/usr/bin/env python3
"""Which torch.cat / torch.stack / .contiguous() calls (shapes, call site) one eager training step makes (development aid)."""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd.engine import Engine, synthetic_batch  # noqa: E402

eng = Engine(device="cuda", seed=1, precision="bf16x3", graph=False)
batch = synthetic_batch(8, 256, 256)
eng.train_step(*batch)
log = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "ebfi_amd" in fr.filename:
            return "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
    return "?"


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*a, **k):
        ts = a[0] if isinstance(a[0], (list, tuple)) else [a[0]]
        log[(name, tuple(tuple(t.shape) for t in ts if hasattr(t, "shape")), site())] += 1
        return orig(*a, **k)
    setattr(mod, name, f)


wrap(torch, "cat")
wrap(torch, "stack")
wrap(torch.nn.functional, "pixel_shuffle")
wrap(torch.nn.functional, "pad")
eng.train_step(*batch)
torch.cuda.synchronize()
for (name, shp, where), n in sorted(log.items(), key=lambda kv: -sum(torch.Size(s).numel() for s in kv[0][1])):
    print("%-14s x%d %-28s %s" % (name, n, where, shp))
