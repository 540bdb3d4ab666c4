#!/usr/bin/env python3
"""Device time of one sub-module of EVFIAutoEx, forward + backward, by kernel (torch profiler).
Usage (GPU box): python tools/partprof.py exposure|detail"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd import conv  # noqa: E402
from ebfi_amd.engine import DEFAULT_MODEL_ARGS  # noqa: E402
from ebfi_amd.model import EVFIAutoEx  # noqa: E402


def main():
    part = sys.argv[1] if len(sys.argv) > 1 else "exposure"
    torch.manual_seed(0)
    conv.set_compute_dtype("bf16x3")
    net = EVFIAutoEx(**DEFAULT_MODEL_ARGS).cuda().train()
    if part == "detail":
        a = torch.rand(8, 3, 256, 256, device="cuda")
        b = torch.rand(8, 3, 256, 256, device="cuda", requires_grad=True)
        run = lambda: net.Detail(img0=a, img1=b).sum().backward()
    else:
        ev = torch.rand(8, 32, 256, 256, device="cuda")
        bl = torch.rand(8, 4, 256, 256, device="cuda")
        run = lambda: net.ExposureDecision(ev, bl).sum().backward()
    for _ in range(3):
        net.zero_grad(set_to_none=True)
        run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        net.zero_grad(set_to_none=True)
        run()
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda r: -r.self_device_time_total)
    tot = sum(r.self_device_time_total for r in rows)
    print("%s fwd+bwd device time %.3f ms in %d launches" % (part, tot / 1e3, sum(r.count for r in rows)))
    for r in rows[:32]:
        print("%9.1f us %4d x avg %7.1f  %s" % (r.self_device_time_total, r.count, r.self_device_time_total / r.count, r.key[:120]))


if __name__ == "__main__":
    main()
