#!/usr/bin/env python3
"""FAC on fp16 filter planes: padded input (the caller pads / folds the padding's adjoint) against the in-kernel replicate
padding (round 5), same box, library event pairs.  usage: python tools/facbench.py [B C H W]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
import torch  # noqa: E402

from ebfi_amd import _native as N  # noqa: E402
from ebfi_amd import f16scale  # noqa: E402

B, C, H, W = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 64, 128, 128)
K = 5
torch.manual_seed(0)
ev = torch.randn(B, C, H, W).cuda()
evp = torch.nn.functional.pad(ev, (2, 2, 2, 2), mode="replicate")
filt = (torch.randn(B, C * K * K, H, W).cuda() * 0.2)
go = torch.randn(B, C, H, W).cuda() * 1e-2
book = f16scale.ScaleBook("cuda")
sf, sg = book.slot("f"), book.slot("g")
book.calibrate(sf, filt)
book.slots[f16scale.SLOT_STRIDE * sg] = 256.0
f16 = (filt * book.scale(sf)).half()
del filt
lib, st = N.lib(), N.stream_ptr(ev.device)
out = torch.empty(B, C, H, W, device="cuda")
gk = torch.empty(B, C * K * K, H, W, dtype=torch.float16, device="cuda")
gp, gu = torch.empty_like(evp), torch.empty_like(ev)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    N.prof_reset()
    N.prof_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    N.prof_enable(False)
    return {k: 1e3 * v[1] / v[0] for k, v in N.prof_collect().items() if v[0]}


cases = {
    "fwd padded": lambda: lib.ebfi_fac_forward_p16(N.ptr(evp), 0, N.ptr(f16), book.ptr(sf), N.ptr(out), B, C, H, W, K, st),
    "fwd in-kernel pad": lambda: lib.ebfi_fac_forward_p16(N.ptr(ev), 1, N.ptr(f16), book.ptr(sf), N.ptr(out), B, C, H, W, K, st),
    "bwd padded": lambda: lib.ebfi_fac_backward_p16(N.ptr(evp), 0, N.ptr(f16), book.ptr(sf), N.ptr(go), N.ptr(gp), N.ptr(gk), book.ptr(sg), 0.01, B, C, H, W, K, st),
    "bwd in-kernel pad": lambda: lib.ebfi_fac_backward_p16(N.ptr(ev), 1, N.ptr(f16), book.ptr(sf), N.ptr(go), N.ptr(gu), N.ptr(gk), book.ptr(sg), 0.01, B, C, H, W, K, st),
}
for name, fn in cases.items():
    t = timed(fn)
    print("%-20s %s" % (name, "  ".join("%s %.1f us" % kv for kv in t.items())), flush=True)
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in (("F.pad replicate", lambda: torch.nn.functional.pad(ev, (2, 2, 2, 2), mode="replicate")),
                 ("replication_pad2d_backward", lambda: torch.ops.aten.replication_pad2d_backward(gp, go, [2, 2, 2, 2]))):
    for _ in range(3):
        fn()
    t0.record()
    for _ in range(20):
        fn()
    t1.record()
    torch.cuda.synchronize()
    print("%-28s %.1f us" % (name, 1e3 * t0.elapsed_time(t1) / 20))
