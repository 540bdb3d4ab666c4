#!/usr/bin/env python3
"""Per-LAYER table of the detail branch (UNet3d_18 folded to 2-D): device time of every stage's forward and backward, its launch
count, output shape, algorithmic bytes / matrix work of its convolution, in the configuration the training step runs it in (weight
bank, fp16 backward with calibrated scales).  torch profiler: a kernel belongs to the stage whose forward call (a record_function
range opened by module hooks) launched it; backward kernels are attributed through the sequence number autograd gives a forward op
and its backward node.
Usage (GPU box): python tools/detail_layers.py > table.txt"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile, record_function

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd.engine import DEFAULT_MODEL_ARGS, Engine, synthetic_batch  # noqa: E402


def stages(detail):
    """(name, module) of the stages a reader of the reference would name: stem, the 8 residual blocks, the 5 decoder stages, the
    feature fusion and the output convolution."""
    out = [("encoder.stem", detail.encoder.stem)]
    for li in range(1, 5):
        for bi, blk in enumerate(getattr(detail.encoder, "layer%d" % li)):
            out.append(("encoder.layer%d.%d" % (li, bi), blk))
    for di, st in enumerate(detail.decoder):
        out.append(("decoder.%d" % di, st))
    out.append(("feature_fuse", detail.feature_fuse))
    out.append(("outconv", detail.outconv))
    return out


def conv_work(mod):
    """[(Cin, Cout, taps, kind)] of the stage's convolutions."""
    macs = []
    for m in mod.modules():
        if isinstance(m, (torch.nn.Conv3d, torch.nn.Conv2d, torch.nn.ConvTranspose3d)):
            k = 1
            for v in m.kernel_size:
                k *= v
            macs.append((m.in_channels, m.out_channels, k, type(m).__name__))
    return macs


def main():
    eng = Engine(DEFAULT_MODEL_ARGS, device="cuda", precision="bf16x3", graph=False, seed=1)
    batch = synthetic_batch(8, 256, 256)
    for _ in range(3):
        eng.train_step(*batch)
    a = torch.rand(8, 3, 256, 256, device="cuda")
    b = torch.rand(8, 3, 256, 256, device="cuda", requires_grad=True)
    detail = eng.model.Detail
    sts = stages(detail)
    names = {id(m): n for n, m in sts}
    rec = {n: {} for n, _ in sts}
    hooks = []

    def pre(mod, inp):
        rf = record_function("FWD " + names[id(mod)])
        rf.__enter__()
        rec[names[id(mod)]]["rf"] = rf

    def post(mod, inp, out):
        r = rec[names[id(mod)]]
        r.pop("rf").__exit__(None, None, None)
        o = out[0] if isinstance(out, (tuple, list)) else out
        r["shape"] = tuple(o.shape)
        r["in_elems"] = sum(t.numel() for t in inp if torch.is_tensor(t))
        r["out_elems"] = o.numel()

    for n, m in sts:
        hooks.append(m.register_forward_pre_hook(pre))
        hooks.append(m.register_forward_hook(post))

    def run():
        eng.model.zero_grad(set_to_none=True)
        with eng._autocast(), eng._bank(), eng._book():
            detail(img0=a, img1=b).sum().backward()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        run()
        torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    # forward: kernels launched inside a stage's record_function range (by CPU-side launch time); backward: autograd runs the
    # stages in reverse, its kernels are attributed through the sequence numbers torch links forward and backward ops with
    evs = prof.events()
    ranges = [(e.name[4:], e.time_range.start, e.time_range.end) for e in evs if e.name.startswith("FWD ")]
    fwd_seq = {}
    fwd_us = {n: [0.0, 0] for n, _ in sts}
    bwd_us = {n: [0.0, 0] for n, _ in sts}
    other = [0.0, 0]
    for e in evs:
        if e.device_type != torch.autograd.DeviceType.CPU or e.name.startswith("FWD "):
            continue
        dev = sum(k.duration for k in e.kernels) if e.kernels else 0.0
        nk = len(e.kernels) if e.kernels else 0
        if nk == 0:
            continue
        owner = None
        for n, s, t in ranges:
            if s <= e.time_range.start <= t:
                owner = n          # (innermost wins: ranges do not nest here)
        if owner is not None:
            fwd_us[owner][0] += dev
            fwd_us[owner][1] += nk
            if e.sequence_nr >= 0:
                fwd_seq[e.sequence_nr] = owner
        elif e.sequence_nr in fwd_seq:
            o = fwd_seq[e.sequence_nr]
            bwd_us[o][0] += dev
            bwd_us[o][1] += nk
        else:
            other[0] += dev
            other[1] += nk
    print("detail branch per stage, B=8 256x256 (device time of one forward + backward; launches in brackets)")
    print("%-20s %-24s %10s %10s   %s" % ("stage", "output", "fwd us", "bwd us", "convolutions (Cin -> Cout, taps)"))
    tf = tb = 0.0
    for n, m in sts:
        r = rec[n]
        f, bq = fwd_us[n], bwd_us[n]
        tf += f[0]
        tb += bq[0]
        convs = ", ".join("%d->%d x%d %s" % c for c in conv_work(m))
        print("%-20s %-24s %7.1f [%2d] %7.1f [%2d]   %s" % (n, "x".join(str(v) for v in r.get("shape", ())), f[0], f[1], bq[0], bq[1], convs))
    print("%-20s %-24s %10.1f %10.1f" % ("sum of stages", "", tf, tb))
    print("%-20s %-24s %10.1f [%d launches: cats / unbinds between stages, gradient accumulation, the weight-bank refresh]" % (
        "not attributed", "", other[0], other[1]))


if __name__ == "__main__":
    main()
