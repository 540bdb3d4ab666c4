#!/usr/bin/env python3
"""Development: forward+backward passes on the default stream and on a side stream (what a graph capture's warm-up does)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd.engine import DEFAULT_MODEL_ARGS, Engine, synthetic_batch
from ebfi_amd import rc_fused
eng = Engine(DEFAULT_MODEL_ARGS, device="cuda", precision="bf16x3", graph=False, seed=9, lr=1e-4)
for it in range(2):
    eng.train_step(*synthetic_batch(2, 128, 128, device="cuda", seed=500 + it, on_device=True))
batch = synthetic_batch(2, 128, 128, device="cuda", seed=502, on_device=True)
def one(tag):
    rc_fused.TRACE.clear()
    eng.bucket.zero()
    eng._fwd_bwd(*batch)
    torch.cuda.synchronize()
    d = {n: (float(mx), int(c)) for n, c, mx, t in rc_fused.TRACE}
    print(tag, {k: d[k] for k in ("f11.xn", "b.gout", "b.xlast", "b.g5", "b11.gc", "b11.ga16") if k in d}, flush=True)
one("default stream pass 0")
one("default stream pass 1")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    one("side stream pass 0")
    one("side stream pass 1")
torch.cuda.current_stream().wait_stream(side)
one("default stream again")
