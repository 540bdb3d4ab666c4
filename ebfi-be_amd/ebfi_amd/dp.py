"""Data parallelism for the training step: one process per GPU, RCCL over xGMI.

The reference wraps the model in DistributedDataParallel (train_ours.py:754) but runs forward AND
backward inside ``model.no_sync()`` on every iteration (train_ours.py:250-272), so its gradient
all-reduce never fires; ranks only share the initial broadcast.  This module implements what that
code evidently intends: identical replicas, per-rank batches, gradients averaged every step.

MI355X-first choices: the whole model is 5.69 M parameters = 22.8 MB of fp32
gradients.  xGMI is point-to-point (7 links per GPU), a ring all-reduce is bound by one link, so
many small buckets would only multiply latency.  We therefore keep ALL gradients in ONE flat
buffer -- ``param.grad`` tensors are views into it, autograd accumulates in place, nothing is
copied -- and issue exactly one all-reduce per optimiser step on that buffer (average = SUM then
one fused scale).  With gloo (CPU tests) the same code path runs unchanged.
"""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


@torch.no_grad()
def reduce_tensor(input_tensor, average=True):
    """myutils/utils.py:80-92: barrier + all_reduce(SUM) (+ / world); identity on one rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return input_tensor
    world_size = dist.get_world_size()
    if world_size < 2:
        return input_tensor
    dist.barrier()
    dist.all_reduce(input_tensor)
    if average:
        input_tensor /= world_size
    return input_tensor


@torch.no_grad()
def broadcast_parameters(module, src=0):
    """What the DDP constructor does once (train_ours.py:754): rank `src`'s parameters and buffers
    to everyone, as one flat message per dtype."""
    if not is_distributed():
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for group in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in group])
        dist.broadcast(flat, src)
        off = 0
        for t in group:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


class FlatGradBucket:
    """All trainable gradients of `module` in one contiguous buffer.

    ``param.grad`` is pre-set to a view of the buffer, so backward accumulates straight into it;
    ``zero()`` is one memset, ``all_reduce_mean()`` one collective.  Optimisers see ordinary
    ``.grad`` tensors.  Call ``zero()`` instead of ``optimizer.zero_grad()`` (which by default would
    drop the views by setting grads to None)."""

    def __init__(self, module, dtype=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev = self.params[0].device
        self.dtype = dtype or self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=self.dtype, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self):
        """Average over ranks in place; no-op on one rank.  Returns the flat buffer."""
        if is_distributed():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / dist.get_world_size())
        return self.flat

    def views_intact(self):
        """True while every param.grad still aliases the flat buffer (debug / test helper)."""
        base = self.flat.untyped_storage().data_ptr()
        return all(p.grad is not None and p.grad.untyped_storage().data_ptr() == base for p in self.params)
