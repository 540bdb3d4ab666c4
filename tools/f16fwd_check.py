#!/usr/bin/env python3
"""fp16-operand FORWARD (Engine(forward_f16="all")) against the exact fp32 mode and the split-precision mode on one GPU:
outputs (max-norm relative, the measure of tests/test_gpu_model.py), loss, packed gradient (relative in norm), and the time of
a forward+backward pass per mode.  B=8 256x256, default widths, re-randomised weights (the x0.1 initialisation gives
Sharp == 0.5 everywhere), several seeds.

    python tools/f16fwd_check.py [seeds...]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))

from ebfi_amd.engine import DEFAULT_MODEL_ARGS, Engine, synthetic_batch      # noqa: E402


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def main():
    seeds = [int(v) for v in sys.argv[1:]] or [31, 77, 5]
    args = dict(DEFAULT_MODEL_ARGS, UseGTEx=False)
    for seed in seeds:
        eng = Engine(args, device="cuda", precision="bf16x3", seed=4, lr=1e-4, forward_f16="all")
        gen = torch.Generator(device="cpu").manual_seed(11 + seed)
        with torch.no_grad():
            for p in eng.model.parameters():
                if p.dim() > 1:
                    p.copy_((torch.randn(p.shape, generator=gen) * (1.2 / p[0].numel() ** 0.5)).cuda())
                else:
                    p.add_((0.05 * torch.randn(p.shape, generator=gen)).cuda())
        batch = synthetic_batch(8, 256, 256, device="cuda", seed=seed)
        res = {}

        def run(tag, n=1):
            out = None
            for _ in range(n):
                eng.bucket.zero()
                eng.book.begin_step()
                with eng._autocast(), eng._bank(), eng._book():
                    s, f = eng.model(*batch[:4])
                    loss = eng.loss(s.float(), f.float(), batch[4], 0, 1)
                    loss.backward()
                    if eng.precision == "bf16x3":
                        eng.book.finish()
                out = (s.detach().clone(), f.detach().clone(), eng.bucket.gather().clone(), loss.detach().clone())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                eng.bucket.zero()
                with eng._autocast(), eng._bank(), eng._book():
                    s, f = eng.model(*batch[:4])
                    loss = eng.loss(s.float(), f.float(), batch[4], 0, 1)
                    loss.backward()
            torch.cuda.synchronize()
            res[tag] = out + ((time.perf_counter() - t0) / 3 * 1e3,)

        eng.precision = "fp32"
        run("fp32")
        eng.precision = "bf16x3"
        eng.book.forward_f16 = None
        run("bf16x3", 3)            # (first passes calibrate the operand slots just in time, later ones use delayed scales)
        eng.book.forward_f16 = "all"
        run("f16", 3)
        eng.book.forward_f16 = "filters"
        run("filters", 3)
        # emulation: ONLY the 128 -> 1600 KernelConv with fp16 operands (input and weight rounded to fp16 at the book's scaling)
        eng.book.forward_f16 = None
        from ebfi_amd import fac as facmod

        def r16(x):
            a = x.detach().abs().amax().clamp_min(1e-30)
            sc = torch.exp2(1.0 - torch.floor(torch.log2(a)))         # |max| * sc in [2, 4)
            return ((x * sc).half().float() / sc)
        kcw = eng.model.Modification.KernelConv.conv2d.weight
        keep = kcw.detach().clone()
        orig_apply = facmod.KernelConvFacTrain.apply
        facmod.KernelConvFacTrain.apply = staticmethod(lambda frame, *rest: orig_apply(r16(frame), *rest))   # (round 6: the node takes the concatenation's parts)
        with torch.no_grad():
            kcw.copy_(r16(kcw))
        try:
            run("kc16emu", 1)
        finally:
            facmod.KernelConvFacTrain.apply = orig_apply
            with torch.no_grad():
                kcw.copy_(keep)
        a = res["fp32"]
        assert a[0].std() > 0.01
        for tag in ("bf16x3", "f16", "filters", "kc16emu"):
            b = res[tag]
            print("seed %d %-7s sharp %.2e final %.2e loss %.2e grad(norm) %.2e  guard %s  eager fwd+bwd %.2f ms (fp32 %.2f)" % (
                seed, tag, rel(b[0], a[0]), rel(b[1], a[1]), abs(b[3].item() - a[3].item()) / abs(a[3].item()),
                ((b[2] - a[2]).norm() / a[2].norm()).item(), eng.book.guard.tolist(), b[4], a[4]), flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
