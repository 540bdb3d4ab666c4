#!/usr/bin/env python3
"""Does the fp16 backward TRAIN?  (round-5 verdict, weak #2: the headline step rests on one-step gradient errors only.)

N optimiser steps from one seed on fresh device-side batches, three arms on the same initial weights and the same batches:

    fp32      every convolution on the exact fp32 matrix cores                       (the reference's arithmetic)
    x3        split-precision forward AND backward (Engine(backward_f16=False))       (round-2 mode)
    default   split-precision forward, fp16-operand backward with delayed scales     (what bench.py times)

The loss is logged every `--log` steps (mean over the window, and the last step's value).  Two tasks:

    random    the synthetic batch of SURVEY.md 8(d): the target frame is independent noise -- nothing to learn beyond its mean;
              shows that the arms stay together on the benchmark's own data
    copy      the target is the (blurry) input frame itself: a learnable identity task -- the loss falls by an order of
              magnitude within the run, so a backward pass that does not train shows as a curve that stays up

    python tools/traincurves.py --steps 300 --batch 8 --size 256 --out profiles/r06/train_curves.json

Every arm must take every step (the fp16 overflow guard skipping one is reported and fails the run with --strict)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))

ARMS = (("fp32", dict(precision="fp32")), ("x3", dict(precision="bf16x3", backward_f16=False)),
        ("default", dict(precision="bf16x3")))


def run_arm(kw, model_args, task, steps, log, B, size, seed, graph=True, lr=1e-4, init_state=None):
    from ebfi_amd.engine import Engine, synthetic_batch
    eng = Engine(model_args, device="cuda", seed=seed, graph=graph, lr=lr, **kw)
    if init_state is not None:
        eng.model.load_state_dict(init_state)
    losses, curve = [], []
    t0 = time.perf_counter()
    for it in range(steps):
        frame, event, t, gtex, target = synthetic_batch(B, size, size, model_args.get("TB", 16), device="cuda", seed=seed + 1000 * it, on_device=True)
        if task == "copy":
            target = frame
        losses.append(eng.train_step(frame, event, t, gtex, target))
        if (it + 1) % log == 0:
            window = torch.stack(losses[-log:]).double().cpu()
            curve.append({"step": it + 1, "loss_mean": float(window.mean()), "loss_last": float(window[-1])})
    torch.cuda.synchronize()
    skipped = eng.book.skipped_steps() if eng.book is not None else 0
    return {"curve": curve, "skipped_steps": skipped, "seconds": round(time.perf_counter() - t0, 2), "optimiser_steps": eng.iteration,
            "finite": bool(torch.isfinite(eng.optimizer.flat).all().item())}


def compare(result, ref="fp32"):
    """max over logged windows of |loss_arm / loss_ref - 1| (window means)"""
    out = {}
    for arm, r in result.items():
        if arm == ref:
            continue
        out[arm] = max(abs(a["loss_mean"] / b["loss_mean"] - 1.0) for a, b in zip(r["curve"], result[ref]["curve"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--log", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--seed", type=int, default=123)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--tasks", default="random,copy")
    ap.add_argument("--small", action="store_true", help="reduced-width model (the test's size)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--strict", action="store_true")
    a = ap.parse_args()
    from ebfi_amd.engine import DEFAULT_MODEL_ARGS
    margs = dict(DEFAULT_MODEL_ARGS)
    if a.small:
        margs.update(step=2, channels=[8, 8, 16, 16])
    report = {"config": {"steps": a.steps, "log_every": a.log, "batch": a.batch, "size": a.size, "seed": a.seed, "lr": a.lr,
                         "model": "reduced (step=2, channels 8/8/16/16)" if a.small else "config/train_ours.yml defaults",
                         "init": "reference initialisation (kaiming x0.1, model_util.py:16-36)", "optimizer": "Adam"}, "tasks": {}}
    bad = False
    for task in a.tasks.split(","):
        res = {}
        for name, kw in ARMS:
            res[name] = run_arm(kw, margs, task, a.steps, a.log, a.batch, a.size, a.seed, lr=a.lr)
            print("[%s/%s] %d steps in %.1f s, skipped %d, last window loss %.6g" % (task, name, a.steps, res[name]["seconds"],
                                                                                    res[name]["skipped_steps"], res[name]["curve"][-1]["loss_mean"]), flush=True)
            bad |= res[name]["skipped_steps"] != 0 or not res[name]["finite"]
        dev = compare(res)
        first, last = res["fp32"]["curve"][0]["loss_mean"], res["fp32"]["curve"][-1]["loss_mean"]
        print("[%s] fp32 loss %.6g -> %.6g (x%.3f); max relative deviation of the window means from the fp32 arm: %s"
              % (task, first, last, last / first, ", ".join("%s %.3e" % kv for kv in dev.items())), flush=True)
        report["tasks"][task] = {"arms": res, "max_rel_deviation_from_fp32": dev, "fp32_loss_first_window": first, "fp32_loss_last_window": last}
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            json.dump(report, fh, indent=1)
        print("wrote", a.out)
    if a.strict and bad:
        raise SystemExit(2)


if __name__ == "__main__":
    main()
