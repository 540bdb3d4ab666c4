#!/usr/bin/env python3
"""Which fp16 operand slots trip the overflow guard?  Runs train_ours' loop (fresh device batch per step, eager) and, before
every ScaleBook.finish(), snapshots {scale, |max|} of all slots: prints the slots whose |max| * scale left [1, 60000] or whose
|max| moved by more than 8x since the previous step.

  python tools/scaletrace.py [--steps 40] [--seed 123]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
import yaml  # noqa: E402
from ebfi_amd.engine import Engine, synthetic_batch  # noqa: E402
from ebfi_amd.f16scale import SLOT_AMAX, SLOT_STRIDE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--seed", type=int, default=123)
    ap.add_argument("--same-batch", action="store_true")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--cal", type=int, default=None, help="Engine.calibration_steps (default: the engine's)")
    ap.add_argument("--rand-init", action="store_true", help="re-randomised weights (tests / infer_ours --rand-init) instead of the x0.1 init")
    args = ap.parse_args()
    with open(os.path.join(ROOT, "ebfi-be_amd", "config", "train_ours.yml")) as fh:
        config = yaml.safe_load(fh)
    dev = torch.device("cuda", 0)
    eng = Engine(config["model"]["args"], device=dev, precision="bf16x3", lr=1e-4, seed=args.seed, graph=False)
    book = eng.book
    if args.cal is not None:
        eng.calibration_steps = args.cal
    if args.rand_init:
        gen = torch.Generator(device="cpu").manual_seed(11)
        with torch.no_grad():
            for p in eng.model.parameters():
                if p.dim() > 1:
                    p.copy_((torch.randn(p.shape, generator=gen) * (1.2 / p[0].numel() ** 0.5)).cuda())
                else:
                    p.add_((0.05 * torch.randn(p.shape, generator=gen)).cuda())
    TB = int(config["model"]["args"]["TB"])
    snaps = []
    orig = book.finish

    def finish():
        n = len(book.index)
        snaps.append(book.slots.reshape(-1, SLOT_STRIDE)[:n, [0, SLOT_AMAX]].clone())
        orig()
    book.finish = finish
    for it in range(args.steps):
        seed = args.seed + (0 if args.same_batch else 1000 * it)
        batch = synthetic_batch(args.batch, 256, 256, TB, device=dev, seed=seed, rank=0, on_device=True)
        loss = eng.train_step(*batch)
        print("step %3d loss %.5e skipped so far %d" % (it, loss.item(), book.skipped_steps()), flush=True)
    keys = {i: k for k, i in book.index.items()}
    prev = None
    for it, s in enumerate(snaps):
        s = s.cpu()
        for i in range(s.shape[0]):
            sc, a = float(s[i, 0]), float(s[i, 1])
            prod = a * sc
            jump = (a / float(prev[i, 1])) if prev is not None and i < prev.shape[0] and float(prev[i, 1]) > 0 else 1.0
            bad = a > 0 and (prod > 60000 or prod < 1.0 or a != a)
            if bad or jump > 8 or (0 < jump < 1 / 8):
                print("step %3d slot %4d %-60s scale 2^%-4d |max| %.3e  scaled %.3e  jump x%.3g%s"
                      % (it, i, str(keys.get(i)), int(torch.log2(torch.tensor(sc)).item()) if sc > 0 else -999, a, prod, jump,
                         "  <-- GUARD" if (prod > 60000 or a != a) else ""))
        prev = s


if __name__ == "__main__":
    main()
