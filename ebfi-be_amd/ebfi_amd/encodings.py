"""Event list -> voxel stack on the device (reference: dataloader/encodings.py:307-350).

``events_to_stack(xs, ys, ts, ps, B, sensor_size)`` keeps the reference's signature and returns the
same ``[2, B, H, W]`` float32 stack (index 0 positive, 1 negative; callers transpose to
``[B, 2, H, W]`` as h5dataset.py:349 does), bit-exact including the reference's shared-bin-edge
and out-of-range-event behaviour (csrc/events.hip).  Differences: inputs must be GPU tensors and
the result stays on the GPU; the caller's xs/ys are NOT zeroed in place (the reference mutates
them as a side effect of events_to_image).
"""
import torch

from . import _native as N


@torch.no_grad()
def events_to_stack(xs, ys, ts, ps, B, sensor_size=(180, 240)):
    N.require_gpu(xs, ys, ts, ps)
    assert len(xs) == len(ys) and len(ys) == len(ts) and len(ts) == len(ps)
    H, W = int(sensor_size[0]), int(sensor_size[1])
    dev = ts.device
    xs, ys, ts = (t.to(torch.float64).contiguous() for t in (xs, ys, ts))
    ps = ps.to(torch.float32).contiguous()
    out = torch.empty((2, int(B), H, W), dtype=torch.float32, device=dev)
    lib = N.lib()
    need = int(lib.ebfi_events_workspace(int(B)))
    ws = torch.empty(max(need, 8), dtype=torch.uint8, device=dev)
    with torch.cuda.device_of(ts):
        rc = lib.ebfi_events_to_stack(N.ptr(xs), N.ptr(ys), N.ptr(ts), N.ptr(ps), int(len(ts)), int(B), H, W,
                                      N.ptr(out), N.ptr(ws), need, N.stream_ptr(dev))
    N.check(rc, "ebfi_events_to_stack")
    return out
