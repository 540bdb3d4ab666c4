#!/usr/bin/env python3
"""Where do two identical runs of the training step part?  Two fresh engines from one seed take the same eager step; backward
hooks on the model's top-level modules record grad_output / grad_input, and the report names, in backward order, the first
module whose grad_input differs although its grad_output is identical (development tool behind
tests/test_gpu_model.py::test_training_step_is_bit_reproducible).  usage: python tools/detcheck.py [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
import torch  # noqa: E402

from ebfi_amd.engine import Engine, synthetic_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
runs = []
for r in range(2):
    eng = Engine(dict(step=3), device="cuda", seed=21, graph=False, precision="bf16x3")
    rec = {}
    order = []

    def hook(name):
        def fn(mod, gin, gout):
            rec[name] = ([None if g is None else g.detach().clone() for g in gout], [None if g is None else g.detach().clone() for g in gin])
            order.append(name)
        return fn
    handles = [m.register_full_backward_hook(hook(n)) for n, m in eng.model.named_children()]
    fwd = {}
    fh = [m.register_forward_hook(lambda mod, a, out, n=n: fwd.__setitem__(n, [o.detach().clone() for o in (out if isinstance(out, (tuple, list)) else [out]) if torch.is_tensor(o)]))
          for n, m in eng.model.named_children()]
    losses = []
    for k in range(steps):
        rec.clear(); order.clear(); fwd.clear()
        losses.append(eng.train_step(*synthetic_batch(2, 128, 128, device="cuda", seed=500 + k)).item())
    torch.cuda.synchronize()
    names = [n for n, p in eng.model.named_parameters() if p.requires_grad]
    grads = {n: p.grad.detach().clone() for n, p in eng.model.named_parameters() if p.grad is not None}
    runs.append((losses, dict(rec), list(order), dict(fwd), grads))
    for h in handles + fh:
        h.remove()
    del eng

(la, ra, oa, fa, ga), (lb, rb, ob, fb, gb) = runs
print("losses", la, lb, "equal" if la == lb else "DIFFERENT")


def same(x, y):
    return (x is None and y is None) or (x is not None and y is not None and torch.equal(x, y))


for n in fa:
    eq = all(same(x, y) for x, y in zip(fa[n], fb[n]))
    print("forward  %-22s outputs %s" % (n, "identical" if eq else "DIFFERENT"))
for n in oa:
    go_eq = all(same(x, y) for x, y in zip(ra[n][0], rb[n][0]))
    gi_eq = all(same(x, y) for x, y in zip(ra[n][1], rb[n][1]))
    print("backward %-22s grad_output %-10s grad_input %s" % (n, "identical" if go_eq else "DIFFERENT", "identical" if gi_eq else "DIFFERENT"))
bad = [n for n in ga if not torch.equal(ga[n], gb[n])]
print("%d of %d parameter gradients differ:" % (len(bad), len(ga)), " ".join(sorted({".".join(n.split(".")[:2]) for n in bad})))
