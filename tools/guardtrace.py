#!/usr/bin/env python3
"""Which scale slots trip the fp16 overflow guard (development): snapshots every slot right before each finish launch."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
from ebfi_amd.engine import DEFAULT_MODEL_ARGS, Engine, synthetic_batch
from ebfi_amd import f16scale
GRAPH = "--graph" in sys.argv
eng = Engine(DEFAULT_MODEL_ARGS, device="cuda", precision="bf16x3", graph=GRAPH, seed=9, lr=1e-4)
book = eng.book
orig = book.finish
step = [0]
snap = torch.zeros_like(book.slots)
def report():
    n = len(book.index)
    sl = snap[: n * f16scale.SLOT_STRIDE].view(n, f16scale.SLOT_STRIDE).cpu()
    inv = {v: k for k, v in book.index.items()}
    bad = [(i, inv[i][1] if isinstance(inv[i], tuple) else inv[i], float(sl[i, 0]), float(sl[i, 32]), float(sl[i, 1])) for i in range(n)
           if not (sl[i, 32] <= 3e38) or sl[i, 32] * sl[i, 0] > 60000]
    print("pass %d: %d slots, %d tripping: %s" % (step[0], n, len(bad), bad[:6]), flush=True)
    step[0] += 1
def finish():
    snap.copy_(book.slots)          # (captured with the graph: the snapshot of every replay)
    orig()
book.finish = finish
for it in range(6):
    batch = synthetic_batch(2, 128, 128, device="cuda", seed=500 + it, on_device=True)
    from ebfi_amd import rc_fused
    if it == 2:
        rc_fused.TRACE.clear()
    eng.train_step(*batch)
    torch.cuda.synchronize()
    if it == 2 and rc_fused.TRACE and not GRAPH:
        for nm, cnt, mx, t in rc_fused.TRACE:
            if nm.startswith("b11") or nm.startswith("b10") or nm.startswith("f11"):
                print("   %-12s eager step max %.4g (bad %d)" % (nm, float(mx), int(cnt)), flush=True)
    if it == 2 and rc_fused.TRACE and GRAPH:
        n = len(rc_fused.TRACE) // 3          # two warm-up passes + the capture: the last third are the graph's tensors
        for name, cnt, mx, t in rc_fused.TRACE[2 * n:]:
            if int(cnt) or not (float(mx) < 1e30):
                print("   TRACE", name, "non-finite", int(cnt), "max", float(mx), flush=True)
            if t is not None:
                bad = (~torch.isfinite(t.float())).nonzero()
                print("   ", name, "shape", tuple(t.shape), "bad", len(bad), flush=True)
                if len(bad):
                    for d in range(bad.shape[1]):
                        u, c = torch.unique(bad[:, d], return_counts=True)
                        print("      dim", d, "values", u[:24].tolist(), "counts", c[:24].tolist(), flush=True)
                    vals = t[tuple(bad[:8].T)]
                    print("      samples", vals.tolist(), "finite max", float(t.float()[torch.isfinite(t.float())].abs().max()), flush=True)
        print("   TRACE entries", len(rc_fused.TRACE), flush=True)
        for k in range(n):
            nm = rc_fused.TRACE[k][0]
            if nm.startswith("b11") or nm.startswith("b10") or nm.startswith("f11"):
                print("   %-12s passA max %.4g (bad %d)  passB max %.4g (bad %d)   replay max %.4g (bad %d)" % (
                    nm, float(rc_fused.TRACE[k][2]), int(rc_fused.TRACE[k][1]), float(rc_fused.TRACE[n + k][2]), int(rc_fused.TRACE[n + k][1]),
                    float(rc_fused.TRACE[2 * n + k][2]), int(rc_fused.TRACE[2 * n + k][1])), flush=True)
    report()
    print(" step", it, "skipped so far", book.skipped_steps(), "guard", book.guard.tolist(), flush=True)
    flat = eng.bucket.flat
    if flat is not None and not torch.isfinite(flat).all():
        off = 0
        badn = []
        for (n, p) in eng.model.named_parameters():
            g = flat[off:off + p.numel()]
            off += p.numel()
            if not torch.isfinite(g).all():
                badn.append((n, int((~torch.isfinite(g)).sum()), p.numel()))
        print("  non-finite gradients:", badn[:12], "...", len(badn), flush=True)
        if it == 2:
            off = 0
            for (n, p) in eng.model.named_parameters():
                g = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                if n in ("ResidualControl.Conv3.0.0.conv2d.weight", "ResidualControl.Conv3.0.0.conv2d.bias", "ResidualControl.Conv3.2.0.conv2d.weight"):
                    idx = (~torch.isfinite(g)).nonzero()
                    print("   ", n, "non-finite at", idx[:5].tolist(), "...", idx[-3:].tolist(), "values", g[~torch.isfinite(g)][:4].tolist(), flush=True)
            print("   all non-finite params:", [b[0] for b in badn], flush=True)
