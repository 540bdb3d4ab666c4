"""Power-of-two operand scales of the fp16 backward kernels (csrc/conv2d_f16.inc.hpp) -- delayed scaling.

The data gradient and the weight gradient of the training step run one fp16 MFMA per product.  fp16 has the significand
the parity bar needs (11 bits: the packed gradient moves from 0.95e-3 to 1.09e-3 of the oracle's) but
not the range: with the reference's x0.1 initialisation (model_util.py:16-36) gradients fall to 1e-29 in the early layers.
Every operand tensor of those kernels therefore has a SLOT of two device floats {scale, running |max|}:

  * the kernel multiplies what it stages by `scale` (a power of two: exact in fp32) and raises `|max|` atomically,
  * `finish()` -- one tiny launch after the backward pass, captured with the step's graph -- sets every slot's next scale so
    that |max| * scale lies in [2, 4) (14 binades of headroom to fp16's 65504; below, fp16 keeps 11 bits down to 2^-14 and
    the matrix cores take subnormal operands: measured, the result's error is flat until |max| * scale < 2^-10),
    clears the maxima and raises the GUARD flag when a value was not finite or could have overflowed; the guarded Adam
    launch then skips the update of that step (guard[1] counts such steps),
  * the FIRST use of a slot is calibrated just in time from the tensor itself (`calibrate`: torch reductions on the launch
    stream, no host synchronisation) -- that is the first eager pass of a run, before any graph is captured.  The engine
    repeats that for every slot during the first `Engine.calibration_steps` (2) optimiser steps of a run: from the x0.1
    initialisation the first Adam update moves activations by fourteen decades (biases 0 -> 1e-4 against 1e-18 signals),
    which no one-step-old scale survives (measured with tools/scaletrace.py: steps 1-4 skipped without it, none with it),
  * the kernels convert with MODE.FP16_OVFL set: a value past the range saturates at 65504 instead of becoming inf, so one
    stale scale does not turn every maximum recorded further down the backward chain into inf (a repair per layer per step).

After those first steps gradient magnitudes move by a few percent from step to step, so a scale that is one step old is as
good as an exact one: the same recipe fp8 training uses, with far more headroom.
"""
import contextlib

import torch

from . import _native as N

_ACTIVE = None
TARGET_EXP = 2                      # |max| * scale in [2^(TARGET_EXP-1), 2^TARGET_EXP)
SLOT_STRIDE, SLOT_AMAX = 64, 32     # floats per slot; offset of the running |max| (csrc/c16.hpp: separate cache lines)
SLOT_FLOOR = 1                      # floor of the running |max| (7/8 of the previous step's: waves below it send no atomic)


def active_book():
    return _ACTIVE


# Engine(forward_f16=...): which FORWARD convolutions of a training step read fp16 operands, in increasing reach
FORWARD_LEVELS = (None, "filters", "all")


def forward_level(book):
    """0 = none, 1 = the KernelConv ("filters"), 2 = + every convolution of ResidualControl ("all": outside the parity bar, an
    experiment)."""
    return FORWARD_LEVELS.index(book.forward_f16) if book is not None else 0


class ScaleBook:
    def __init__(self, device, capacity=4096):
        self.device = torch.device(device)
        self.capacity = int(capacity)
        init = torch.zeros(self.capacity, SLOT_STRIDE)
        init[:, 0] = 1.0
        self.slots = init.reshape(-1).to(self.device)          # per slot: scale at [0], running |max| at [SLOT_AMAX]
        self.guard = torch.zeros(2, dtype=torch.int32, device=self.device)   # [flag of the current step, skipped steps]
        self.index = {}
        self.calibrated = set()
        self.forward_f16 = None         # Engine(forward_f16=...): one of FORWARD_LEVELS

    def slot(self, key):
        """Index of the slot named `key` (created on first use; keys are any hashable: (site key, role))."""
        i = self.index.get(key)
        if i is None:
            i = len(self.index)
            if i >= self.capacity:
                raise RuntimeError("ScaleBook: more than %d operand slots" % self.capacity)
            self.index[key] = i
        return i

    def ptr(self, i):
        return N._vp(self.slots.data_ptr() + 4 * SLOT_STRIDE * i)

    def scale(self, i):
        """Current scale of slot i (host value: synchronises; tests / diagnostics)."""
        return float(self.slots[SLOT_STRIDE * i].item())

    def amax(self, i):
        return float(self.slots[SLOT_STRIDE * i + SLOT_AMAX].item())

    def calibrate(self, i, *tensors):
        """First use of slot i: set its scale from the tensors about to be staged through it (device-side, no sync)."""
        if i in self.calibrated:
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ScaleBook: slot %d met for the first time inside a graph capture; run one eager pass first" % i)
        amax = torch.stack([t.detach().abs().amax().float() for t in tensors]).amax()
        e = torch.floor(torch.log2(amax.clamp_min(1e-37))) + 1.0          # amax = m * 2^e, m in [0.5, 1)
        scale = torch.where((amax > 0) & torch.isfinite(amax), torch.exp2(TARGET_EXP - e), torch.ones_like(amax))
        self.slots[SLOT_STRIDE * i:SLOT_STRIDE * i + 1].copy_(scale.reshape(1))
        self.calibrated.add(i)

    def operand(self, key, *tensors):
        """Slot pointer for the operand named `key`, calibrated on first use from `tensors`."""
        i = self.slot(key)
        self.calibrate(i, *tensors)
        return self.ptr(i)

    def begin_step(self):
        """Clear the guard flag at the start of an accumulation window -- an EAGER launch, never part of the captured graph
        (Engine._begin_micro_step: a clear baked into the graph would wipe a flag an earlier micro-step raised)."""
        self.guard[0:1].zero_()

    def finish(self):
        """After the backward pass: next scales from the recorded maxima, guard flag on overflow / non-finite data."""
        n = len(self.index)
        if n == 0:
            return
        with torch.cuda.device(self.device):
            rc = N.lib().ebfi_f16_scales_finish(N.ptr(self.slots), n, N.ptr(self.guard), N.stream_ptr(self.device))
        N.check(rc, "ebfi_f16_scales_finish")

    def skipped_steps(self):
        return int(self.guard[1].item())

    @contextlib.contextmanager
    def active(self):
        global _ACTIVE
        prev, _ACTIVE = _ACTIVE, self
        try:
            yield self
        finally:
            _ACTIVE = prev
