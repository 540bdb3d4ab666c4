#!/usr/bin/env python3
"""Which Python line launches each PyTorch-native kernel of one EAGER training step (development aid: what is left outside the
library, by call site).  torch.profiler with stacks; prints, per (kernel, call site in ebfi_amd), launches and device time.
usage (GPU box): python tools/gluestack.py [--no-detail]"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd.engine import Engine, synthetic_batch  # noqa: E402


def site_of(stack):
    """first frame inside ebfi_amd (innermost), as file:line function"""
    for fr in stack:
        if "ebfi_amd" in fr and "_native.py" not in fr:
            return fr.split("ebfi-be_amd/")[-1]
    return stack[0] if stack else "?"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-detail", action="store_true")
    a = ap.parse_args()
    eng = Engine(dict(DetailEnabled=False) if a.no_detail else None, device="cuda", seed=1, precision="bf16x3", graph=False)
    batch = synthetic_batch(8, 256, 256)
    for _ in range(3):
        eng.train_step(*batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        eng.train_step(*batch)
        torch.cuda.synchronize()
    # CPU ops with their stacks; device kernels are linked to the launching op by correlation: use the op events' device time
    rows = collections.defaultdict(lambda: [0, 0.0])
    total = 0.0
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        if ev.cpu_parent is not None and ev.cpu_parent.kernels and any(k in ev.cpu_parent.kernels for k in ev.kernels):
            pass                                     # (kernels are attached to the innermost op only)
        for k in ev.kernels:
            name = k.name
            if "at::native" not in name and "rocclr" not in name and "Memcpy" not in name and "Memset" not in name:
                continue
            short = name.replace("at::native::", "").replace("(anonymous namespace)::", "")[:70]
            shapes = str([list(x) for x in (ev.input_shapes or []) if x])[:58]
            key = (short, ev.name, site_of(ev.stack or []) + " " + shapes)
            rows[key][0] += 1
            rows[key][1] += k.duration
            total += k.duration
    for (short, op, site), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("%8.1f us %3d x  %-24s %-62s %s" % (us, n, op[:24], site[:62], short[:48]))
    print("total %.1f us in PyTorch-native kernels of one eager step" % total)


if __name__ == "__main__":
    main()
