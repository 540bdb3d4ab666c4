"""fp16 subnormal operands on the matrix cores: the weight gradient of a 64 -> 64 layer with both operand scales pushed down by
2^-k (|max| * scale from 2^8 to 2^-16).  The error stays at 2.7e-4 down to 2^-10 and degrades gradually below: subnormals are
not flushed.  (The data-gradient column is meaningless here: the weight image was packed with the unshifted scale.)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))
os.environ["EBFI_DEV"] = "1"
from tests.test_gpu_conv import _banked_layer, _ref, _rel
from ebfi_amd import conv
from ebfi_amd.f16scale import SLOT_STRIDE
torch.manual_seed(1)
Cin, Cout, B, H, W = 64, 64, 2, 16, 64
w, b, bank, book = _banked_layer(Cin, Cout)
x = torch.randn(B, Cin, H, W); g = torch.randn(B, Cout, H, W)
xr, wr, br = x.clone().requires_grad_(), w.detach().cpu().clone().requires_grad_(), b.detach().cpu().clone().requires_grad_()
_ref(xr, wr, br, 1, 1, 0, 0.01).backward(g)
conv.set_compute_dtype("bf16x3")
for k in (0, 4, 8, 12, 16, 18, 20, 22, 24):
    for tr in ("1", "0"):
        os.environ["EBFI_WGRAD_TR"] = tr
        w.grad = None; b.grad = None
        xd = x.cuda().requires_grad_()
        with bank.active(), book.active():
            y = conv.conv_bias_act(xd, w, b, 1, 1, 0, 0.01)
            if k == 0 and tr == "1":
                y.backward(g.cuda())          # calibrates
                base = {i: book.scale(i) for i in book.index.values()}
                w.grad = None; b.grad = None
                xd = x.cuda().requires_grad_()
                y = conv.conv_bias_act(xd, w, b, 1, 1, 0, 0.01)
            for i, s in base.items():
                book.slots[SLOT_STRIDE * i] = s * 2.0 ** (-k)
            y.backward(g.cuda())
        torch.cuda.synchronize()
        print("scales x 2^-%-2d (|max| -> 2^%d) tr=%s: dgrad err %.3e  wgrad err %.3e  bias err %.3e"
              % (k, 8 - k, tr, _rel(xd.grad, xr.grad), _rel(w.grad, wr.grad), _rel(b.grad, br.grad)), flush=True)
