"""Blur-level maps on the device (reference: myutils/utils.py:34-49 Frame2Lap, :15-31 Frame2DCP).

Same call signatures and output layout ([B,1,H,W] float32 on the input's device), but computed by
``ebfi_frame2lap`` / ``ebfi_frame2dcp`` instead of a per-sample host OpenCV loop.  Not
differentiable (neither is the reference's: it goes through numpy).  Arithmetic = OpenCV's 8-bit
BGR2GRAY + 3x3 Laplacian / channel-min + 35x35 erosion; PARITY UNPINNED (oracle/blur_ref.py).
"""
import torch

from . import _native as N


@torch.no_grad()
def Frame2Lap(ims):
    N.require_gpu(ims)
    x = ims.detach().contiguous().float()
    B, C, H, W = x.shape
    assert C == 3, "Frame2Lap expects Bx3xHxW"
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device_of(x):
        rc = N.lib().ebfi_frame2lap(N.ptr(x), N.ptr(out), B, H, W, N.stream_ptr(x.device))
    N.check(rc, "ebfi_frame2lap")
    return out


@torch.no_grad()
def Frame2DCP(ims, sz=35):
    N.require_gpu(ims)
    x = ims.detach().contiguous().float()
    B, C, H, W = x.shape
    assert C == 3, "Frame2DCP expects Bx3xHxW"
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    tmp = torch.empty((B, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device_of(x):
        rc = N.lib().ebfi_frame2dcp(N.ptr(x), N.ptr(out), N.ptr(tmp), B, H, W, int(sz), N.stream_ptr(x.device))
    N.check(rc, "ebfi_frame2dcp")
    return out
