#!/usr/bin/env python3
"""infer_ours.py -- MI355X counterpart of the reference inference entry point (infer_ours.py:156-172,
:113-118): loads a checkpoint in the reference layout (`cpt['config']['model']`, `cpt['model']['states']`),
builds the model by name and runs `model(Frame, Event, T, GTEx)[-1]` under no_grad.  Inputs are
synthetic clips (BASELINE.json configs 1/2/5); metrics / PNG dumps / HDF5 lists are out of scope.

    python infer_ours.py --model_path output/models/Ours/run/checkpoint-iteration99.pth --batch 4 --height 256 --width 256
    (train_ours.py names a checkpoint after the LAST COMPLETED iteration, counted from 0 like the reference: a 100-iteration
    run writes checkpoint-iteration99.pth and is resumed at iteration 100)
    python infer_ours.py --batch 1 --height 128 --width 128          # random-init weights
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ebfi_amd.engine import DEFAULT_MODEL_ARGS, synthetic_batch  # noqa: E402
from models.Ours.model_singleframe import EVFIAutoEx  # noqa: E402,F401  (resolved by name, like the reference's eval())


def load_model(model_path, device):
    if model_path is None:
        name, margs, states = "EVFIAutoEx", dict(DEFAULT_MODEL_ARGS), None
    else:
        cpt = torch.load(model_path, map_location="cpu")
        name, margs, states = cpt["config"]["model"]["name"], cpt["config"]["model"]["args"], cpt["model"]["states"]
    model = globals()[name](**margs)
    if states is not None:
        model.load_state_dict(states)
    return model.to(device).eval(), margs


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_path", default=None)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--num_ts", type=int, default=16, help="latent timestamps per clip (NumI of the reference loop)")
    ap.add_argument("--seed", type=int, default=123)
    ap.add_argument("--precision", default="bf16x3", choices=["fp32", "bf16x3", "bf16"],
                    help="matrix-core operands of the convs: bf16x3 = split bf16 pairs, fp32-grade accuracy (default); fp32 = exact")
    ap.add_argument("--rand-init", action="store_true",
                    help="without --model_path: draw O(1)-gain random weights instead of the reference's x0.1 initialisation, "
                         "whose output is the constant 0.5 (benchmark / profile runs: the printed mean then depends on the data)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-hoist", action="store_true",
                    help="recompute the timestamp-independent prefix (feature extractors, exposure decision) for every timestamp")
    a = ap.parse_args()
    torch.manual_seed(a.seed)
    device = torch.device("cuda", 0)
    model, margs = load_model(a.model_path, device)
    if a.rand_init and a.model_path is None:
        with torch.no_grad():
            for p in model.parameters():
                if p.dim() > 1:
                    p.copy_(torch.randn_like(p) * (1.2 / p[0].numel() ** 0.5))
                else:
                    p.add_(0.05 * torch.randn_like(p))
    from ebfi_amd.engine import ClipInterpolator
    # Frame / Event are the same for every latent timestamp of a clip (reference loop infer_ours.py:113-118): the part of the
    # forward that does not depend on T -- padding, both feature extractors, Frame2Lap + ExposureDecision (6.2 of 91 GMAC) --
    # runs ONCE per clip, the per-timestamp part is replayed from a captured hipGraph (ebfi_amd.engine.ClipInterpolator).
    # --no-hoist keeps the plain model(Frame, Event, T, GTEx) call per timestamp (bit-identical outputs).
    interp = ClipInterpolator(model, precision=a.precision, graph=not a.no_graph, hoist=not a.no_hoist)
    frame, event, _, gtex, _ = synthetic_batch(a.batch, a.height, a.width, margs["TB"], device=device, seed=a.seed)
    stamps = [i / float(a.num_ts) for i in range(a.num_ts)]
    interp(frame, event, gtex, stamps[:1])        # untimed: module load, allocator, graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = interp(frame, event, gtex, stamps)      # one clip: the prefix once + num_ts replays
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("interpolated %d frames of %dx%d in %.3f s: %.1f frames/s; output %s, mean %.4f std %.4f, peak memory %.1f GB"
          % (a.batch * a.num_ts, a.height, a.width, dt, a.batch * a.num_ts / dt, tuple(out.shape), out.mean().item(),
             out.std().item(), torch.cuda.max_memory_allocated(device) / 1e9))


if __name__ == "__main__":
    main()
