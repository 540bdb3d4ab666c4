#!/usr/bin/env python3
"""Device time of the detail branch (UNet3d_18 folded to 2-D), forward + backward, by kernel, in the configuration the training
step runs it in (weight bank, fp16 backward with calibrated scales).  torch profiler: includes the PyTorch-native glue kernels.
Usage (GPU box): python tools/detailprof.py [--trace]    (--trace: every launch in order, for a per-layer table)"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd.engine import DEFAULT_MODEL_ARGS, Engine, synthetic_batch  # noqa: E402


def main():
    eng = Engine(DEFAULT_MODEL_ARGS, device="cuda", precision="bf16x3", graph=False, seed=1)
    batch = synthetic_batch(8, 256, 256)
    for _ in range(3):
        eng.train_step(*batch)
    a = torch.rand(8, 3, 256, 256, device="cuda")
    b = torch.rand(8, 3, 256, 256, device="cuda", requires_grad=True)

    def run():
        eng.model.zero_grad(set_to_none=True)
        with eng._autocast(), eng._bank(), eng._book():
            eng.model.Detail(img0=a, img1=b).sum().backward()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run()
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    rows = sorted(prof.key_averages(), key=lambda r: -r.self_device_time_total)
    tot = sum(r.self_device_time_total for r in rows)
    print("detail fwd+bwd device time %.3f ms in %d launches" % (tot / 1e3, sum(r.count for r in rows)))
    for r in rows[:40]:
        print("%9.1f us %4d x avg %7.1f  %s" % (r.self_device_time_total, r.count, r.self_device_time_total / r.count, r.key[:110]))
    if "--trace" in sys.argv:
        print("---- launches in order")
        for e in sorted(evs, key=lambda e: e.time_range.start):
            print("%8.1f us  %s" % (e.time_range.elapsed_us(), e.name[:100]))


if __name__ == "__main__":
    main()
