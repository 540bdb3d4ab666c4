"""ResidualControl with fp16 image storage against its layer-wise fp16 form: packed gradients of one training step (development)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd.engine import Engine, synthetic_batch, DEFAULT_MODEL_ARGS
from ebfi_amd import rc_fused
# image path vs layer-wise fp16 path: same engine weights, compare packed gradients
def run(images):
    orig = rc_fused.images_usable
    if not images:
        rc_fused.images_usable = lambda *a, **k: False
    try:
        eng = Engine(DEFAULT_MODEL_ARGS, device="cuda", precision="bf16x3", graph=False, seed=4, lr=1e-4)
        g = torch.Generator(device="cpu").manual_seed(11)
        with torch.no_grad():
            for p in eng.model.parameters():
                if p.dim() > 1: p.copy_((torch.randn(p.shape, generator=g) * (1.2 / p[0].numel() ** 0.5)).cuda())
                else: p.add_((0.05 * torch.randn(p.shape, generator=g)).cuda())
        out=[]
        for it in range(4):
            batch = synthetic_batch(2, 128, 128, device="cuda", seed=50+it)
            loss = eng.train_step(*batch)
            out.append((loss.item(), eng.bucket.flat.detach().clone()))
        print("images", images, "skipped", eng.book.skipped_steps(), [o[0] for o in out])
        return out
    finally:
        rc_fused.images_usable = orig
a = run(False); b = run(True)
for (la, ga), (lb, gb) in zip(a, b):
    print("loss diff", abs(la-lb)/abs(la), "grad rel", ((ga-gb).norm()/ga.norm()).item())
