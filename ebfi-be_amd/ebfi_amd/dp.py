"""Data parallelism for the training step: one process per GPU, RCCL over xGMI.

The reference wraps the model in DistributedDataParallel (train_ours.py:754) but runs forward AND
backward inside ``model.no_sync()`` on every iteration (train_ours.py:250-272), so its gradient
all-reduce never fires; ranks only share the initial broadcast.  This module implements what that
code evidently intends: identical replicas, per-rank batches, gradients averaged every step.

MI355X-first choices: the whole model is 5.69 M parameters = 22.8 MB of fp32
gradients.  xGMI is point-to-point (7 links per GPU), a ring all-reduce is bound by one link, so
many small buckets would only multiply latency.  We therefore pack ALL gradients into ONE flat
buffer (one gather launch after backward: csrc/optim.hip grad_gather) and issue exactly ONE collective
per optimiser step on it (average = SUM then one fused scale); afterwards ``param.grad`` are views of
that buffer.  The overflow flag of the fp16 backward rides in the same message (one extra element:
SUM > 0 <=> some rank raised it), so a step has one all-reduce, not two.
With gloo (CPU tests) the same code path runs unchanged.
"""
import os

import torch
import torch.distributed as dist


def is_distributed():
    """More than one rank -- or EBFI_FORCE_COLLECTIVES=1 with an initialised group of one: the collectives then run at
    world size 1 (bench.py EBFI_BENCH_FORCE_DIST: what a one-GPU box can exercise of the RCCL path)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("EBFI_FORCE_COLLECTIVES") == "1"


@torch.no_grad()
def sync_guard(guard):
    """Overflow guard of the fp16 backward (ebfi_amd.f16scale: int32[2] = [flag of this step, skipped steps]) MAX-reduced over
    ranks with a collective of its own.  The engine does NOT use this any more: FlatGradBucket.gather(guard=...) carries the
    flag inside the gradient message (one collective per step); kept for callers that hold a guard without a bucket."""
    if guard is not None and is_distributed():
        if guard.is_cuda and dist.get_backend() == "gloo":     # (rehearsal fabric: reduce through the host)
            host = guard[0:1].cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX)
            guard[0:1].copy_(host)
        else:
            dist.all_reduce(guard[0:1], op=dist.ReduceOp.MAX)
    return guard


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


WIRE_PAD = 4               # floats behind the gradients in the wire buffer: [overflow flag, 0, 0, 0] (keeps 16-byte multiples)


class FlatGradBucket:
    """All trainable gradients of `module` in one contiguous buffer, built AFTER backward.

    Autograd hands every parameter a freshly written gradient tensor (the conv kernels already write
    their own outputs), so the cheapest way to a flat buffer is one gather per step (libebfi_hip.so
    `ebfi_grad_gather`: the source pointers travel by value in the kernel arguments, 128 tensors per
    launch, one workgroup per 16384-element chunk; 22.8 MB read + written once) instead of 255 in-place
    accumulations into pre-assigned views plus a memset -- or a 255-piece torch.cat, whose batched copy
    gives every input the same number of workgroups and took 90 us.  ``gather()`` packs,
    ``all_reduce_mean()`` averages over ranks with one collective, and afterwards every ``param.grad``
    is a view of the flat buffer, so optimisers see ordinary ``.grad`` tensors.  ``zero()`` drops the
    gradients (set-to-None) for the next step.

    Wire format: ``wire`` = [numel gradients | flag | 3 zeros]; ``flat`` = wire[:numel].  ``flag`` is the
    overflow guard of the fp16 backward as a float (gather(guard=book.guard) writes guard[0] != 0 there),
    summed over ranks by the SAME all-reduce: a positive value on any rank skips the update on all."""

    def __init__(self, module, dtype=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        self.dtype = dtype or self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.wire = None
        self.flat = None
        self._offsets = []
        off = 0
        for p in self.params:
            self._offsets.append(off)
            off += p.numel()
        self._numels = None                         # ctypes int64 array of the parameters' element counts (native gather)

    @property
    def flag(self):
        """The overflow flag element of the wire buffer (a 1-element view), None before the first gather."""
        return None if self.wire is None else self.wire[self.numel:self.numel + 1]

    def zero(self):
        for p in self.params:
            p.grad = None

    def adopt(self, wire):
        """Make `wire` (a full wire buffer, e.g. a graph's static output or an accumulation buffer) the current one."""
        self.wire = wire
        self.flat = wire[:self.numel]

    def gather(self, guard=None):
        """Pack the per-parameter gradients (missing ones count as zero) into a fresh wire buffer and re-point .grad at it.
        `guard`: int32[2] device tensor of f16scale.ScaleBook (or a CPU tensor in the gloo tests): its flag rides along."""
        dev = self.params[0].device
        grads = []
        for p in self.params:
            g = p.grad
            if g is not None and (g.dtype != self.dtype or not g.is_contiguous()):
                g = g.to(self.dtype).contiguous()
            grads.append(g)
        if dev.type != "cuda" or self.dtype != torch.float32:
            # CPU parameters (host tests, gloo rehearsals) and bucket dtypes the native launch does not pack: one concatenation
            flag = torch.zeros(WIRE_PAD, dtype=self.dtype, device=dev)
            if guard is not None:
                flag[0] = 1.0 if int(guard[0]) != 0 else 0.0
            pieces = [(g if g is not None else torch.zeros_like(p, dtype=self.dtype)).reshape(-1) for g, p in zip(grads, self.params)]
            self.adopt(torch.cat(pieces + [flag]))
        else:
            if guard is not None and not (guard.is_cuda and guard.device == dev and guard.dtype == torch.int32):
                # (the kernel dereferences this pointer: a host tensor -- the gloo tests pass one with CPU parameters -- or
                # another dtype must be staged, never handed over as it is)
                guard = guard.to(device=dev, dtype=torch.int32)
            import ctypes
            from . import _native as N
            n = len(self.params)
            if self._numels is None:
                self._numels = (ctypes.c_int64 * n)(*[p.numel() for p in self.params])
            ptrs = (ctypes.c_void_p * n)(*[None if g is None else g.data_ptr() for g in grads])
            wire = torch.empty(self.numel + WIRE_PAD, dtype=self.dtype, device=dev)
            with torch.cuda.device(dev):
                rc = N.lib().ebfi_grad_gather(ptrs, self._numels, n, N.ptr(wire), self.numel, WIRE_PAD, N.ptr(guard), N.stream_ptr(dev))
            N.check(rc, "ebfi_grad_gather")
            # (the sources may be freed right away: the caching allocator reuses memory in stream order only; inside a graph
            # capture they are pool allocations whose addresses the captured launches keep)
            self.adopt(wire)
        for p, off in zip(self.params, self._offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)
        return self.flat

    def reduce_mean_packed(self):
        """Average the already packed wire buffer over ranks in place -- gradients AND the overflow flag in ONE collective
        (no-op on one rank).  The flag ends up as (ranks that raised it) / world: > 0 iff any did."""
        if is_distributed():
            if self.wire.is_cuda and dist.get_backend() == "gloo":
                # rehearsal / CPU-only fabrics: gloo reduces device tensors through tiny staged chunks (seconds for
                # 22.8 MB); one explicit round trip through host memory is two copies and a host all-reduce
                host = self.wire.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                self.wire.copy_(host)
            else:
                dist.all_reduce(self.wire, op=dist.ReduceOp.SUM)
            self.wire.mul_(1.0 / dist.get_world_size())
        return self.flat

    def all_reduce_mean(self, guard=None):
        """gather() + average over ranks in place.  Returns the flat buffer."""
        self.gather(guard)
        return self.reduce_mean_packed()

    def views_intact(self):
        """True while every param.grad aliases the flat buffer (debug / test helper)."""
        if self.flat is None:
            return False
        base = self.flat.untyped_storage().data_ptr()
        return all(p.grad is not None and p.grad.untyped_storage().data_ptr() == base for p in self.params)


class FlatAdam:
    """Adam over ONE flat parameter buffer.

    The trainable parameters are re-pointed at views of a single contiguous buffer (their values, names and shapes do
    not change), the packed gradient of `FlatGradBucket` becomes that buffer's `.grad`, and the update is one fused
    launch over 5.7 M elements instead of a multi-tensor pass over 255 tensors.  Same arithmetic as
    ``torch.optim.Adam(params, lr, betas, eps)`` (it IS torch's fused Adam, applied to one tensor).
    `state_dict()` / `load_state_dict()` speak torch's per-parameter layout -- what the reference's checkpoints hold
    (train_ours.py:621-671) -- so checkpoints stay interchangeable."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        with torch.no_grad():
            flat = torch.cat([p.data.reshape(-1) for p in self.params])
        self._offsets, off = [], 0
        for p in self.params:
            self._offsets.append(off)
            p.data = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.flat = torch.nn.Parameter(flat)
        self.inner = torch.optim.Adam([self.flat], lr=lr, betas=betas, eps=eps, amsgrad=False, fused=bool(flat.is_cuda))

    @property
    def param_groups(self):
        return self.inner.param_groups

    def step(self, flat_grad, guard=None, flag=None):
        """`flat_grad`: the packed gradient in the order of `params` (FlatGradBucket.flat).  `guard`: int32[2] tensor of
        ebfi_amd.f16scale.ScaleBook -- a non-zero guard[0] skips this update (guard[1] counts the skipped steps).  `flag`:
        FlatGradBucket.flag after the all-reduce -- the ranks' overflow flags summed by the gradient collective; a non-zero
        (or non-finite) value skips the update as well and is written back to guard[0], so every rank takes the same branch."""
        if flat_grad.numel() != self.flat.numel():
            raise ValueError("packed gradient has %d elements, parameters %d" % (flat_grad.numel(), self.flat.numel()))
        if flag is not None and guard is None:
            raise ValueError("an overflow flag needs the guard tensor that counts the skipped steps")
        if self._native_step(flat_grad, guard, flag):
            return
        if guard is not None:                                     # torch's own step (CPU / unusual options): host-side check
            if flag is not None and not float(flag[0]) == 0.0:
                guard[0] = 1
            if int(guard[0].item()) != 0:
                guard[1] += 1
                return
        self.flat.grad = flat_grad
        self.inner.step()
        self.flat.grad = None

    def _native_step(self, flat_grad, guard=None, flag=None):
        """The update as ONE bandwidth-bound launch of libebfi_hip.so (csrc/optim.hip) on the state tensors of the inner
        torch.optim.Adam (which keeps owning hyper-parameters, state and checkpoint layout).  CPU tensors, weight decay,
        amsgrad or maximize take torch's own step."""
        from . import _native as N
        grp = self.inner.param_groups[0]
        if not (self.flat.is_cuda and self.flat.dtype == torch.float32 and flat_grad.dtype == torch.float32 and
                flat_grad.is_contiguous() and not grp.get("amsgrad") and not grp.get("maximize") and
                float(grp.get("weight_decay", 0.0)) == 0.0 and not torch.is_tensor(grp["lr"]) and
                N.dev_env("EBFI_NO_NATIVE_ADAM") is None):
            return False
        st = self.inner.state[self.flat]
        if not st:
            st["step"] = torch.zeros((), dtype=torch.float32, device=self.flat.device)
            st["exp_avg"] = torch.zeros_like(self.flat.data)
            st["exp_avg_sq"] = torch.zeros_like(self.flat.data)
        if not (torch.is_tensor(st["step"]) and st["step"].is_cuda):
            st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=self.flat.device)
        # the bookkeeping torch's own step() wrapper does: step hooks, and the marks the LR schedulers look at
        # (`_opt_called`: without it every scheduler.step() warns "lr_scheduler.step() before optimizer.step()")
        opt = self.inner
        for hook in list(getattr(opt, "_optimizer_step_pre_hooks", {}).values()):
            hook(opt, (), {})
        st["step"] += 1
        b1, b2 = grp["betas"]
        with torch.cuda.device_of(self.flat):
            rc = N.lib().ebfi_adam_step_guarded(N.ptr(self.flat.data), N.ptr(flat_grad), N.ptr(st["exp_avg"]), N.ptr(st["exp_avg_sq"]),
                                                N.ptr(st["step"]), self.flat.numel(), float(grp["lr"]), float(b1), float(b2),
                                                float(grp["eps"]), N.ptr(guard), N.ptr(flag), N.stream_ptr(self.flat.device))
        N.check(rc, "ebfi_adam_step_guarded")
        opt._opt_called = True
        if hasattr(opt, "_step_count"):
            opt._step_count += 1
        for hook in list(getattr(opt, "_optimizer_step_post_hooks", {}).values()):
            hook(opt, (), {})
        return True

    def views_intact(self):
        base = self.flat.untyped_storage().data_ptr()
        return all(p.untyped_storage().data_ptr() == base for p in self.params)

    def state_dict(self):
        st = self.inner.state.get(self.flat, {})
        state = {}
        if st:
            for i, (p, off) in enumerate(zip(self.params, self._offsets)):
                sl = slice(off, off + p.numel())
                state[i] = {"step": st["step"].clone() if torch.is_tensor(st["step"]) else st["step"],
                            "exp_avg": st["exp_avg"][sl].view_as(p).clone(), "exp_avg_sq": st["exp_avg_sq"][sl].view_as(p).clone()}
        group = {k: v for k, v in self.inner.param_groups[0].items() if k != "params"}
        group["lr"] = float(group["lr"])
        group["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [group]}

    # param_group entries that describe HOW this process runs the update, not the optimisation state: a checkpoint written
    # by torch.optim.Adam (the reference's) carries fused=None / foreach=None, which would silently turn the single fused
    # launch into the per-tensor path with a host sync per step
    _LOCAL_KEYS = ("params", "fused", "foreach", "capturable", "differentiable")

    def load_state_dict(self, sd):
        """Accepts torch's per-parameter Adam layout (what the reference's checkpoints hold, train_ours.py:621-671).
        Parameters without an entry (never received a gradient before the save) start from zero moments.  All parameters
        share ONE step counter here (the update is one launch over the flat buffer, missing gradients count as zero), so
        the per-parameter steps of the checkpoint must agree -- they do whenever every parameter had a gradient on every
        step, which is the case for this model; a checkpoint whose steps differ (parameters frozen for part of the run) is
        refused instead of being silently re-timed."""
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("optimizer state has %d parameters in %d group(s), this model trains %d"
                             % (sum(len(g["params"]) for g in groups), len(groups), len(self.params)))
        for k, v in groups[0].items():
            if k not in self._LOCAL_KEYS and k in self.inner.param_groups[0]:
                self.inner.param_groups[0][k] = v
        if not sd["state"]:
            self.inner.state.pop(self.flat, None)
            return
        ids = groups[0]["params"]
        m, v = torch.zeros_like(self.flat.data), torch.zeros_like(self.flat.data)
        step = None
        for pid, p, off in zip(ids, self.params, self._offsets):
            e = sd["state"].get(pid)
            if e is None:
                continue
            m[off:off + p.numel()] = e["exp_avg"].reshape(-1).to(m)
            v[off:off + p.numel()] = e["exp_avg_sq"].reshape(-1).to(v)
            if step is not None and float(e["step"]) != step:
                raise ValueError("optimizer state holds different step counts per parameter (%g and %g): the flat Adam keeps "
                                 "one counter for all parameters" % (step, float(e["step"])))
            step = float(e["step"])
        step = 0.0 if step is None else step
        step = torch.tensor(step, dtype=torch.float32, device=self.flat.device if self.flat.is_cuda else "cpu")
        self.inner.state[self.flat] = {"step": step, "exp_avg": m, "exp_avg_sq": v}
