"""The fp16-operand backward must TRAIN like the exact fp32 arithmetic, not just agree on one step's gradient (round-5 verdict).

tools/traincurves.py at reduced size: the same initial weights, the same fresh device-side batches, Adam, four arms --

    fp32      exact fp32 matrix cores (the reference's arithmetic)
    fp32p     the SAME arithmetic from initial weights perturbed by 1e-6 relative: the yardstick -- how far two runs drift apart
              from a last-digit difference alone
    default   split-precision forward, fp16 backward with delayed operand scales, hipGraph replay (what bench.py times)

Measured at full size (B=8 256x256, 300 steps; profiles/r06/train_curves.json): from O(1) weights the default arm's 10-step loss
means stay within 1.3 % of the fp32 arm's (the split-precision arm 1.5 %, the yardstick arm 2.4 %).  From the reference's x0.1
initialisation every arm sits on a plateau for ~50 steps and leaves it at a step that depends on the last digit (yardstick
12 %, split precision 360 %, default 68 % at the worst window; all arms then fall along the same curve): a fixed-step comparison
from that initialisation measures the plateau's exit time, not the backward pass -- so the test runs from O(1) weights and bounds
the default arm by the LARGER of 5 % and 1.5x the yardstick arm's own deviation; no optimiser step may be skipped, and the
learnable task must actually be learnt."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_default_arm_tracks_the_fp32_arm():
    import traincurves
    from ebfi_amd.engine import DEFAULT_MODEL_ARGS
    margs = dict(DEFAULT_MODEL_ARGS, step=2, channels=[8, 8, 16, 16])
    steps, log = 80, 10
    res = {}
    for name, kw in traincurves.ARMS:
        if name == "x3":
            continue
        res[name] = traincurves.run_arm(kw, margs, "copy", steps, log, 2, 64, 123, lr=1e-3, init="o1",
                                        perturb=traincurves.PERTURB if name == "fp32p" else 0.0)
        assert res[name]["skipped_steps"] == 0 and res[name]["finite"] and res[name]["optimiser_steps"] == steps, (name, res[name])
    dev = traincurves.compare(res)
    curve = res["fp32"]["curve"]
    assert len(curve) == steps // log
    assert curve[-1]["loss_mean"] < 0.7 * curve[0]["loss_mean"], curve         # the task is learnable and is being learnt
    assert dev["default"] < max(0.05, 1.5 * dev["fp32p"]), (dev, [(a["loss_mean"], b["loss_mean"]) for a, b in zip(res["default"]["curve"], curve)])


def test_benchmark_data_arms_agree():
    """On the benchmark's own synthetic batches (the target is independent noise: nothing to learn beyond its mean) the default arm's
    loss stays within 1e-4 of the fp32 arm's at every window, from the reference initialisation, with no step skipped."""
    import traincurves
    from ebfi_amd.engine import DEFAULT_MODEL_ARGS
    margs = dict(DEFAULT_MODEL_ARGS, step=2, channels=[8, 8, 16, 16])
    res = {name: traincurves.run_arm(kw, margs, "random", 40, 10, 2, 64, 123, lr=1e-4, init="reference")
           for name, kw in traincurves.ARMS if name in ("fp32", "default")}
    assert all(r["skipped_steps"] == 0 and r["finite"] for r in res.values())
    assert traincurves.compare(res)["default"] < 1e-4
