#!/usr/bin/env python3
"""N graph-replayed training steps and nothing else: run under `rocprofv3 --kernel-trace --stats` to see the device time
of a replayed step by kernel (the 5 eager warm-up / capture steps are <15 % of the trace at --steps 40).
Usage (GPU box): rocprofv3 --kernel-trace --stats --output-format csv -d out -o g -- python3 tools/graphprof.py --steps 40"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd.engine import Engine, synthetic_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--no-detail", action="store_true", help="DetailEnabled=False (what the detail branch costs)")
    ap.add_argument("--gtex", action="store_true", help="UseGTEx=True (what the exposure-decision head costs)")
    ap.add_argument("--rc-steps", type=int, default=12, help="ResidualControl rounds")
    a = ap.parse_args()
    over = dict(step=a.rc_steps)
    if a.no_detail:
        over["DetailEnabled"] = False
    if a.gtex:
        over["UseGTEx"] = True
    eng = Engine(over, device="cuda", seed=1, precision=a.precision, graph=True)
    batch = synthetic_batch(8, 256, 256)
    eng.train_step(*batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.train_step(*batch)
    torch.cuda.synchronize()
    print("%.3f ms/step over %d replayed steps" % (1e3 * (time.perf_counter() - t0) / a.steps, a.steps))


if __name__ == "__main__":
    main()
