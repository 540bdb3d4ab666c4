#!/usr/bin/env python3
"""Device time of the detail branch (UNet3d_18) alone, forward + backward, by kernel (library event pairs + torch profiler)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd import _native as N  # noqa: E402
from ebfi_amd import conv  # noqa: E402
from ebfi_amd.engine import DEFAULT_MODEL_ARGS  # noqa: E402
from ebfi_amd.model import EVFIAutoEx  # noqa: E402


def main():
    torch.manual_seed(0)
    conv.set_compute_dtype("bf16x3")
    net = EVFIAutoEx(**DEFAULT_MODEL_ARGS).cuda().train()
    a = torch.rand(8, 3, 256, 256, device="cuda")
    b = torch.rand(8, 3, 256, 256, device="cuda", requires_grad=True)
    for _ in range(3):
        net.zero_grad(set_to_none=True)
        net.Detail(img0=a, img1=b).sum().backward()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        net.zero_grad(set_to_none=True)
        net.Detail(img0=a, img1=b).sum().backward()
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda r: -r.self_device_time_total)
    tot = sum(r.self_device_time_total for r in rows)
    print("detail branch fwd+bwd device time %.3f ms in %d launches" % (tot / 1e3, sum(r.count for r in rows)))
    for r in rows[:40]:
        print("%9.1f us %4d x avg %7.1f  %s" % (r.self_device_time_total, r.count, r.self_device_time_total / r.count, r.key[:110]))


if __name__ == "__main__":
    main()
