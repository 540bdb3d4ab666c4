#!/usr/bin/env python3
"""Attribute the PyTorch-side (non-library) device time of one training step to aten ops and source lines.
Usage (GPU box): python tools/stepprof.py [--top 40]"""
import argparse
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd.engine import Engine, synthetic_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    a = ap.parse_args()
    eng = Engine(device="cuda", seed=1)
    batch = synthetic_batch(a.batch, a.size, a.size)
    for _ in range(3):
        eng.train_step(*batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        eng.train_step(*batch)
        torch.cuda.synchronize()
    print("== by op ==")
    print(prof.key_averages().table(sort_by="self_device_time_total", row_limit=a.top, max_name_column_width=60))
    print("== by op + stack ==")
    rows = prof.key_averages(group_by_stack_n=6)
    rows = sorted(rows, key=lambda r: -r.self_device_time_total)
    for r in rows[:a.top * 2]:
        if r.self_device_time_total <= 0:
            continue
        stack = [s for s in r.stack if "ebfi" in s or "bench" in s or "tools/" in s][:3]
        print("%9.1f us %5d x  %-40s %s" % (r.self_device_time_total, r.count, r.key[:40], " <- ".join(s.split("/")[-1] for s in stack)))


if __name__ == "__main__":
    main()
