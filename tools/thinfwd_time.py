"""Device time of ebfi_conv2d_thin_forward at the model's two thin-out shapes (development aid).  usage (GPU box): python tools/thinfwd_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "ebfi-be_amd"))
from ebfi_amd import _native as N
lib = N.lib()
B, Cin, H, W = 8, 64, 256, 256
for Cout in (3, 1):
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda"); b = torch.randn(Cout, device="cuda")
    out = torch.empty(B, Cout, H, W, device="cuda")
    st = N.stream_ptr(x.device)
    def run():
        rc = lib.ebfi_conv2d_thin_forward(N.ptr(x), N.ptr(w), N.ptr(b), N.ptr(out), B, Cin, H, W, Cout, 3, 1, 1, 2, 0.0, st)
        assert rc == 0, rc
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print("Cout", Cout, "%.1f us" % (e0.elapsed_time(e1) * 1e3 / 50))
