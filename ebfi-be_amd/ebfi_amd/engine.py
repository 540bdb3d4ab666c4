"""Training / inference engine for the hot path: what train_ours.py:250-277 and
infer_ours.py:113-118 do per step, minus the reference's I/O, logging and bookkeeping.

One process per GPU.  A step = forward -> Lap+census loss -> backward -> ONE flat RCCL gradient
all-reduce (ebfi_amd.dp) -> Adam.  Synthetic batches follow SURVEY.md section 8(d).
"""
import contextlib

import torch

from .dp import FlatAdam, FlatGradBucket, broadcast_parameters
from .loss import TrainLoss
from .model import EVFIAutoEx

# config/train_ours.yml:28-57 of the reference (the hot-path hyper-parameters)
DEFAULT_MODEL_ARGS = dict(
    FrameBasech=64, EventBasech=64, InterCH=64, TB=16, norm=None, activation="LeakyReLU",
    UseGTEx=False, FixEx=None, BlurryFashion="RGBLap", BLInch=4, UseEvents=True, LoadPretrainEX=False,
    PretrainedEXPath=None, FrozenEX=False, step=12, DualPath=True, residual=True, DetailEnabled=True,
    channels=[16, 24, 32, 64])


def synthetic_batch(B, H, W, TB=16, device="cuda", seed=123, rank=0, on_device=False):
    """Frame in [0,1], Event = Poisson(0.35) integer counts, T in [0,1),
    GTEx in [0.55,0.95), random GT frame.  Seed = reference default (train_ours.py:806) + rank.
    on_device: draw with the DEVICE generator (no host work, no PCIe copy: what a training loop that makes one batch per
    step needs -- the host Poisson draw of a B=8 256x256 batch takes ~0.5 s against a 20 ms step); the default draws on the
    host so that tests and benchmarks see the same numbers on every machine."""
    dev = torch.device(device)
    if on_device and dev.type == "cuda":
        g = torch.Generator(device=dev).manual_seed(seed + rank)
        rnd = lambda *shape: torch.rand(*shape, generator=g, device=dev)
        frame = rnd(B, 3, H, W)
        event = torch.poisson(torch.full((B, TB, 2, H, W), 0.35, device=dev), generator=g)
        t = rnd(B, 1)
        gtex = rnd(B, 1) * 0.4 + 0.55
        return frame, event, t, gtex, rnd(B, 3, H, W)
    g = torch.Generator(device="cpu").manual_seed(seed + rank)
    frame = torch.rand(B, 3, H, W, generator=g)
    event = torch.poisson(torch.full((B, TB, 2, H, W), 0.35), generator=g)
    t = torch.rand(B, 1, generator=g)
    gtex = torch.rand(B, 1, generator=g) * 0.4 + 0.55
    target = torch.rand(B, 3, H, W, generator=g)
    return tuple(v.to(device) for v in (frame, event, t, gtex, target))


def synthetic_batch_from_raw_events(B, H, W, TB=16, device="cuda", seed=123, rank=0, rate=0.35, on_device=False):
    """Same batch statistics as `synthetic_batch`, but the event tensor comes the way the reference produces it
    (h5dataset.py:327-352): a sorted raw event list per sample (N ~ rate*H*W*TB events uniform in x, y, sorted uniform t in
    [0,1), polarity +-1) binned by the device `events_to_stack` kernel and transposed to [TB, 2, H, W].
    on_device: the event lists are drawn and sorted on the device too (see `synthetic_batch`)."""
    from .encodings import events_to_stack
    dev = torch.device(device)
    on_device = on_device and dev.type == "cuda"
    # (the Poisson voxel draw of synthetic_batch is skipped here: TB=0 gives an empty event tensor in the same stream position)
    frame, _, t, gtex, target = synthetic_batch(B, H, W, 0 if on_device else TB, device=device, seed=seed, rank=rank,
                                                on_device=on_device)
    gdev = dev if on_device else torch.device("cpu")
    g = torch.Generator(device=gdev).manual_seed(seed + rank + 7919)
    n = int(rate * H * W * TB)
    stacks = []
    for _ in range(B):
        xs = torch.randint(0, W, (n,), generator=g, device=gdev).to(device)
        ys = torch.randint(0, H, (n,), generator=g, device=gdev).to(device)
        ts = torch.sort(torch.rand(n, generator=g, dtype=torch.float64, device=gdev))[0].to(device)
        ps = (torch.randint(0, 2, (n,), generator=g, device=gdev) * 2 - 1).float().to(device)
        stacks.append(events_to_stack(xs, ys, ts, ps, TB, sensor_size=(H, W)).transpose(0, 1))   # [TB, 2, H, W]
    return frame, torch.stack(stacks).contiguous(), t, gtex, target


class Engine:
    def __init__(self, model_args=None, device="cuda", precision="fp32", lr=1e-4, seed=None, train=True, graph=False,
                 accu_step=1, betas=(0.9, 0.999), backward_f16=None, forward_f16="filters", strict_graph=None):
        """strict_graph: a hipGraph capture that fails raises instead of continuing with eager launches.  Default: on when
        the process is one rank of several (a rank that silently runs ~20 % slower drags every rank of the job), off alone.
        forward_f16 (only with the fp16 backward; ignored otherwise): which FORWARD convolutions read fp16 operand images (one matrix-core product per
        tap, fp32 accumulation) instead of running in split precision:
          "filters"  the 128 -> 1600 KernelConv of Modification, whose output exists only as fp16 planes anyway (ebfi_amd.fac):
                     Sharp / Final move from 1.6-2.9e-4 to 2.8-4.1e-4 of the fp32 result (tools/f16fwd_check.py, three seeds),
                     inside the 1e-3 of BASELINE.json (tests/test_gpu_model.py::test_training_forward_within_tolerance);
                     1.14 -> 0.60 ms per step.  The default.
          "all"      + every convolution of ResidualControl: 1.4-1.9e-3 -- OUTSIDE the parity bar (the twelve rounds amplify
                     the operand rounding); an experiment switch, never what bench.py or the tests run
          None       split-precision (bf16x3) forward everywhere.
        (EBFI_DEV=1 EBFI_F16_FWD=none|filters|all overrides the argument: same-box A/B runs of bench.py.)
        backward_f16 (training, precision 'bf16x3' only; default on): the data / weight gradients of the 3x3 layers run ONE
        fp16 MFMA per product with delayed power-of-two operand scales (ebfi_amd.f16scale) instead of three bf16 ones; the
        forward pass keeps the split-precision kernels.  Gradient parity is unchanged (tests/test_gpu_model.py:
        test_benchmarked_step_vs_oracle); False restores the split-precision backward."""
        if precision not in ("fp32", "bf16x3", "bf16"):
            raise ValueError("precision must be 'fp32', 'bf16x3' or 'bf16'")
        self.device = torch.device(device)
        self.precision = precision
        if seed is not None:
            torch.manual_seed(seed)
        self.model_args = dict(DEFAULT_MODEL_ARGS, **(model_args or {}))
        self.model = EVFIAutoEx(**self.model_args).to(self.device)
        broadcast_parameters(self.model, 0)
        self.iteration = 0                                  # optimiser steps taken (the reference's train_iter_idx)
        # gradient accumulation (trainer.accu_step, train_ours.py:206,259-277): every call of train_step is one
        # forward+backward on loss / accu_step; the all-reduce and the optimiser step happen on every accu_step-th call
        self.accu_step = max(1, int(accu_step))
        self._micro = 0
        # fp16 backward: the first optimiser steps of a run measure every operand scale just in time (eager launches).  The
        # x0.1 initialisation leaves activations at 1e-18 in the deep stacks; the first Adam update (every weight and bias
        # moves by lr) lifts them to 1e-4 -- fourteen decades in one step, which no one-step-old scale survives.  From the
        # third step on magnitudes move by a few percent per step and the delayed scales (f16scale) take over.
        self.calibration_steps = 2
        self._steps_run = 0
        self._accum = None
        # graph=True: forward + loss + backward + gradient packing are captured once into a hipGraph (per input shape,
        # precision and loss phase) and replayed; the all-reduce and the optimiser step stay eager (train_step_graph).
        self.use_graph = bool(graph) and self.device.type == "cuda"
        self._graphs = {}
        self.graph_capture_failed, self.graph_capture_error = False, None    # set when a capture fell back to eager launches
        if strict_graph is None:
            from .dp import is_distributed
            strict_graph = is_distributed()
        self.strict_graph = bool(strict_graph)
        if train:
            self.model.train()
            self.loss = TrainLoss(self.model_args.get("DetailEnabled", True)).to(self.device)
            self.bucket = FlatGradBucket(self.model)
            # one flat parameter buffer (parameters become views of it) updated by ONE fused Adam launch on the packed
            # gradient (0.4 ms per step less than the multi-tensor pass over 255 tensors); state_dict() keeps torch's
            # per-parameter layout for checkpoints
            self.optimizer = FlatAdam(self.bucket.params, lr=lr, betas=tuple(betas))
        else:
            self.model.eval()
        # MFMA operand images of every conv weight, refreshed by one launch per step instead of one per conv call
        # (ebfi_amd.weightbank); training reads the optimiser's flat parameter buffer, inference keeps its own copy
        self.bank = None
        from . import _native as N
        if backward_f16 is None:
            backward_f16 = N.dev_env("EBFI_NO_F16_BWD", "0") != "1"
        use_book = train and precision == "bf16x3" and backward_f16
        forward_f16 = N.dev_env("EBFI_F16_FWD", forward_f16) if use_book else None
        if forward_f16 == "none":
            forward_f16 = None
        from .f16scale import FORWARD_LEVELS
        if forward_f16 not in FORWARD_LEVELS:
            raise ValueError("forward_f16 must be one of %r" % (FORWARD_LEVELS,))
        if self.device.type == "cuda" and N.dev_env("EBFI_NO_BANK", "0") != "1":     # (switch for A/B measurements)
            from . import weightbank
            self.bank = weightbank.build_for(self.model, flat=self.optimizer.flat.data, params=self.optimizer.params,
                                             fwd16=forward_f16) if train else weightbank.build_for(self.model, inference=True)
        self.book = None
        if use_book and self.bank is not None:
            from . import f16scale
            self.book = f16scale.ScaleBook(self.device)
            self.bank.attach_scale_book(self.book)
            self.book.forward_f16 = forward_f16

    @property
    def settled(self):
        """True once the steps that are not the steady state are over: the just-in-time calibration steps of the fp16 backward
        and, with graph=True, the capture.  (Throughput loggers start their clock here.)"""
        if self.book is not None and self.precision == "bf16x3" and self._steps_run < self.calibration_steps:
            return False
        return not self.use_graph or bool(self._graphs)

    @contextlib.contextmanager
    def _autocast(self):
        """precision 'bf16' = bf16 matrix-core operands for the 3x3 / 1x1 convs (fp32 storage and
        accumulation, ebfi_amd.conv); everything else, and precision 'fp32', computes in fp32."""
        from . import conv
        prev = conv.get_compute_dtype()
        conv.set_compute_dtype(self.precision)
        try:
            yield
        finally:
            conv.set_compute_dtype(prev)

    def _bank(self):
        """Context of one pass: the weight bank refreshed (ONE pack launch, captured with the step's graph) and active
        while the split-precision mode runs; a no-op otherwise."""
        if self.bank is None or self.precision != "bf16x3":
            return contextlib.nullcontext()
        self.bank.refresh()
        return self.bank.active()

    def _book(self):
        """Context of one pass: the fp16 backward kernels active on the scale book (a no-op without one)."""
        if self.book is None or self.precision != "bf16x3":
            return contextlib.nullcontext()
        return self.book.active()

    def _fwd_bwd(self, frame, event, t, gtex, target):
        with self._autocast(), self._bank(), self._book() as book:
            if book is not None and self._steps_run < self.calibration_steps:
                book.calibrated.clear()             # (measure every operand of this pass again, see __init__)
            sharp_pre, sharp = self.model(frame, event, t, gtex)
            loss = self.loss(sharp_pre.float(), sharp.float(), target, self.iteration, self.accu_step)
            loss.backward()
            if book is not None:
                book.finish()                       # next step's operand scales from this pass's maxima; guard on overflow
        return loss.detach()

    def _begin_micro_step(self):
        """Clears the fp16 overflow guard at the start of an accumulation window -- EAGERLY, never inside the captured
        region: a graph replays whatever `_micro` was at capture time, so a clear baked into it would wipe the flag a
        previous micro-step of the same window raised (round-3 advisory); the guard therefore stays an OR over the window
        and is only cleared here, before the window's first forward."""
        if self.book is not None and self.precision == "bf16x3" and self._micro == 0:
            self.book.begin_step()

    def _guard(self):
        return self.book.guard if (self.book is not None and self.precision == "bf16x3") else None

    def _finish_micro_step(self, wire):
        """`wire`: this call's packed gradient (of loss / accu_step) with the overflow flag behind it (FlatGradBucket.wire).
        Sums the calls of one accumulation window in place and, on the window's last call, averages over ranks -- ONE
        collective: the flag travels with the gradients -- and takes the optimiser step.  True when a step was taken."""
        if self.accu_step > 1:
            if self._micro == 0:
                self._accum = wire.clone() if self._accum is None else self._accum.copy_(wire)
            else:
                self._accum.add_(wire)           # (the flags add up too: > 0 if any micro-step raised the guard)
            self._micro += 1
            if self._micro < self.accu_step:
                return False
            self._micro = 0
            wire = self._accum
        self.bucket.adopt(wire)
        self.bucket.reduce_mean_packed()
        guard = self._guard()
        self.optimizer.step(self.bucket.flat, guard=guard, flag=self.bucket.flag if guard is not None else None)
        self.iteration += 1
        self._steps_run += 1
        return True

    def train_step(self, frame, event, t, gtex, target):
        """One forward+backward on this rank's batch (plus, every `accu_step`-th call, the gradient all-reduce and the
        optimiser step); returns the (unreduced) loss tensor, already divided by accu_step like the reference's."""
        calibrating = self.book is not None and self.precision == "bf16x3" and self._steps_run < self.calibration_steps
        if self.use_graph and not calibrating:
            return self.train_step_graph(frame, event, t, gtex, target)
        self._begin_micro_step()
        self.bucket.zero()
        loss = self._fwd_bwd(frame, event, t, gtex, target)
        self.bucket.gather(self._guard())
        self._finish_micro_step(self.bucket.wire)
        return loss

    def train_step_graph(self, frame, event, t, gtex, target):
        """Same step with the ~560 launches of forward + loss + backward + gradient packing replayed from one
        hipGraph.  Inputs are copied into static buffers; gradients live in the graph's memory pool, `param.grad`
        are views of the packed buffer, so the eager all-reduce / Adam that follow see ordinary tensors."""
        inputs = (frame, event, t, gtex, target)
        phase = self.iteration < 10e3                      # TrainLoss switches its weighting at 10k iterations
        key = (self.precision, phase, self.accu_step) + tuple((tuple(v.shape), v.dtype) for v in inputs)
        entry = self._graphs.get(key)
        if entry is None:
            static_in = [torch.empty_like(v) for v in inputs]
            for s, v in zip(static_in, inputs):
                s.copy_(v)
            # The warm-up passes and the capture run _fwd_bwd -> book.finish() outside any accumulation window: a flag they
            # raise (or their three advances of the delayed scales on one batch) must not leak into the window this call
            # belongs to -- with accu_step > 1 the capture can happen mid-window, where nothing clears the guard afterwards
            # (round-4 advisory).  The guard words are restored after the capture; the scales keep what the warm-up passes
            # measured on this batch (restoring them could undo the just-in-time calibration of a slot first met here).
            guard = self._guard()
            saved = None if guard is None else guard.clone()
            side = torch.cuda.Stream(self.device)          # warm-up off the default stream (allocator, lazy inits)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.bucket.zero()
                    self._fwd_bwd(*static_in)
            torch.cuda.current_stream(self.device).wait_stream(side)
            self.bucket.zero()
            graph = torch.cuda.CUDAGraph()
            try:
                # thread_local: another thread of the process (the RCCL watchdog polling its events) must not
                # invalidate the capture
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    loss = self._fwd_bwd(*static_in)
                    self.bucket.gather(guard)
                    wire = self.bucket.wire
            except RuntimeError as err:
                if saved is not None:
                    torch.cuda.synchronize(self.device)
                    guard.copy_(saved)
                if self.strict_graph:
                    raise RuntimeError("ebfi_amd.engine: hipGraph capture failed and strict_graph is set (the default for one "
                                       "rank of several): %s" % str(err).splitlines()[0]) from err
                # a capture that cannot be taken must not cost the run: say so and continue with eager launches
                import sys
                print("ebfi_amd.engine: hipGraph capture failed (%s); continuing with eager launches" % str(err).splitlines()[0],
                      file=sys.stderr, flush=True)
                self.use_graph = False
                self.graph_capture_failed, self.graph_capture_error = True, str(err).splitlines()[0]
                torch.cuda.synchronize(self.device)
                return self.train_step(frame, event, t, gtex, target)
            if saved is not None:
                guard.copy_(saved)
            entry = (graph, static_in, loss, wire, [p.grad for p in self.bucket.params])
            self._graphs[key] = entry
        graph, static_in, loss, wire, grads = entry
        self._begin_micro_step()
        for s, v in zip(static_in, inputs):
            if s.data_ptr() != v.data_ptr():
                s.copy_(v)
        graph.replay()
        self.bucket.adopt(wire)
        for p, g in zip(self.bucket.params, grads):        # (another shape's graph may have re-pointed them)
            p.grad = g
        self._finish_micro_step(wire)
        return loss.clone()

    @torch.no_grad()
    def infer(self, frame, event, t, gtex=None):
        with self._autocast(), self._bank():
            return self.model(frame, event, t, gtex)[-1]


class ClipInterpolator:
    """Inference over the latent timestamps of a clip (reference loop infer_ours.py:113-118: the same Frame / Event / GTEx for
    every timestamp, only T changes): the timestamp-independent prefix of the network (`EVFIAutoEx.encode`: both feature
    extractors, Frame2Lap + ExposureDecision) runs ONCE per clip, the per-timestamp rest (`decode`) is captured into a hipGraph
    per input shape and replayed with a new T.  Outputs are bit-identical to calling the model once per timestamp
    (tests/test_gpu_model.py::test_hoisted_inference_is_bit_identical).  precision 'bf16x3' packs the conv weight images once
    (inference weight bank, incl. the fused KernelConv -> FAC layout)."""

    GROUP_PIXELS = 16 * 256 * 256      # auto grouping: timestamps per pass so that B * k * H * W stays at or below this

    def __init__(self, model, precision="bf16x3", graph=True, hoist=True, filters_f16=True, group=None):
        """group (round 6; with hoist): how many latent timestamps ONE pass of the per-timestamp network computes.  The prefix's
        outputs are the same for every timestamp, so k timestamps are a batch of k * B samples that differ only in T (samples are
        independent: no BatchNorm, per-sample GroupNorm / pooling).  None = as many as keep B * k * H * W <= GROUP_PIXELS (4 at
        BASELINE config 2, B=4 256x256: the launches of a B=4 pass are mostly at their latency floor; 1 at config 5); 1 = one
        timestamp per pass, bit-identical to the reference loop's per-timestamp module call.  Grouped passes agree with that to
        fp32 rounding (the kernels pick tile geometries by problem size; tests/test_gpu_entrypoints.py).
        filters_f16 (precision 'bf16x3' only; default on): the fused KernelConv(128 -> 1600) -> FAC kernel -- a third of the
        model's multiply-adds -- reads fp16 operands, one matrix-core product per tap instead of the split precision's three,
        with exact power-of-two operand scales (the input's measured on the device before every launch).  It is the one
        convolution the TRAINING step runs on fp16 operands too (Engine(forward_f16='filters')): outputs stay within 4e-4 of the
        exact fp32 mode against the path's 1e-3 (tests/test_gpu_model.py, tests/test_infer_cli.py); False keeps split precision."""
        # (inference needs eval mode; the caller's model is NOT switched for good: its mode is restored after every call)
        self.model = model
        self.precision, self.graph, self.hoist = precision, bool(graph), bool(hoist)
        self.group = None if group is None else max(1, int(group))
        self.graph_capture_failed, self.graph_capture_error = False, None    # as Engine: a failed capture continues eagerly
        self.bank = self.book = None
        if precision == "bf16x3" and next(model.parameters()).is_cuda:
            from . import f16scale, weightbank
            if filters_f16:
                self.book = f16scale.ScaleBook(next(model.parameters()).device, capacity=64)
            self.bank = weightbank.build_for(model, inference=True, book=self.book)
            self.bank.refresh()
        self._captured = {}

    @contextlib.contextmanager
    def _ctx(self):
        from . import conv
        prev = conv.get_compute_dtype()
        conv.set_compute_dtype(self.precision)
        # (the caller's model may be a training model with some sub-modules held in eval -- frozen norm layers: remember and
        #  restore every module's own flag; `model.train(flag)` is recursive and would flatten them)
        modes = [(m, m.training) for m in self.model.modules()]
        self.model.eval()
        try:
            if self.bank is not None:
                # the images are packed once; a load_state_dict / optimiser step since then bumps the parameters' version
                # counters (WeightBank._current_stamp) and the bank is re-packed here instead of serving stale weights
                self.bank.ensure_fresh()
            with (self.bank.active() if self.bank is not None else contextlib.nullcontext()):
                yield
        finally:
            conv.set_compute_dtype(prev)
            for m, flag in modes:
                m.training = flag

    def refresh_weights(self):
        if self.bank is not None:
            self.bank.refresh()

    def _step(self, state, frame, event, gtex, t):
        return self.model.decode(state, t)[-1] if self.hoist else self.model(frame, event, t, gtex)[-1]

    @torch.no_grad()
    def __call__(self, frame, event, gtex, timestamps, out=None, timing=False):
        """timestamps: iterable of floats (or [B,1] tensors) -> [B, len(timestamps), 3, H, W] (the `Final` output).
        out: a preallocated [B, len(timestamps), 3, H, W] tensor to fill (a serving loop hands the same buffer in for every
        clip); otherwise ONE result tensor is allocated per call and every timestamp's output is copied into its slice -- no
        per-timestamp clone, no torch.stack of the clones (at B=8 720x1280 those were 5 fresh allocations of 88-354 MB per clip).
        timing: record `self.last_timing = {"encode_ms", "decode_ms"}` (device time of the timestamp-independent prefix and of
        all per-timestamp passes of this call) with two event pairs on the current stream; the call synchronises."""
        timestamps = list(timestamps)
        if not timestamps:
            raise ValueError("ClipInterpolator: no timestamps")
        B, dev = frame.shape[0], frame.device
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if (timing and dev.type == "cuda") else None
        with self._ctx():
            if ev:
                ev[0].record()
            state = self.model.encode(frame, event, gtex) if self.hoist else None
            if ev:
                ev[1].record()
            result = out
            n = len(timestamps)
            k = 1
            if self.hoist:
                k = self.group if self.group is not None else max(1, self.GROUP_PIXELS // max(1, B * frame.shape[-2] * frame.shape[-1]))
                k = max(1, min(k, n))
            self.last_group = k

            def put(i, value, count=1):
                """value: [count * B, 3, H, W] (pass-major: sample b of timestamp j at row j * B + b) -> result[:, i : i + count]"""
                nonlocal result
                shape = (B, n) + tuple(value.shape[1:])
                if result is None:
                    result = torch.empty(shape, dtype=value.dtype, device=value.device)
                elif i == 0 and (tuple(result.shape) != shape or result.dtype != value.dtype or result.device != value.device):
                    raise ValueError("ClipInterpolator: out must be a %s tensor of shape %r on %s" % (value.dtype, shape, value.device))
                if count == 1:
                    result[:, i].copy_(value[:B])
                else:
                    result[:, i:i + count].copy_(value[:count * B].view((count, B) + tuple(value.shape[1:])).transpose(0, 1))

            def t_tensor(chunk, rows):
                """[rows * B, 1]: T of every sample of a pass (a short last chunk repeats its last timestamp; the extra rows are dropped)"""
                cols = []
                for j in range(rows):
                    ts = chunk[min(j, len(chunk) - 1)]
                    cols.append(ts.reshape(B, 1).to(dev) if torch.is_tensor(ts) else torch.full((B, 1), float(ts), device=dev))
                return torch.cat(cols, 0)

            def replicated(st, rows):
                return tuple(torch.cat([v] * rows, 0) if torch.is_tensor(v) and rows > 1 else v for v in st) if self.hoist else st

            def eager():
                rep = replicated(state, k) if self.hoist else None
                for i0 in range(0, n, k):
                    chunk = timestamps[i0:i0 + k]
                    if self.hoist:
                        put(i0, self.model.decode(rep, t_tensor(chunk, k))[-1], len(chunk))
                    else:
                        put(i0, self._step(state, frame, event, gtex, t_tensor(chunk, 1)))

            def done():
                if ev:
                    ev[2].record()
                    ev[2].synchronize()
                    self.last_timing = {"encode_ms": ev[0].elapsed_time(ev[1]), "decode_ms": ev[1].elapsed_time(ev[2])}
                return result

            if not self.graph or dev.type != "cuda":
                eager()
                return done()
            key = (tuple(frame.shape), tuple(event.shape), gtex is not None, k)
            ent = self._captured.get(key)
            if ent is None:
                # static inputs of the captured graph: the prefix's outputs, replicated k times (hoisted), or the raw inputs; plus T
                st_in = list(replicated(tuple(v.clone() if torch.is_tensor(v) else v for v in state), k)) if self.hoist else \
                    [frame.clone(), event.clone(), None if gtex is None else gtex.clone()]
                t_static = torch.zeros(k * B, 1, device=dev)
                call = (lambda: self.model.decode(tuple(st_in), t_static)[-1]) if self.hoist else \
                    (lambda: self.model(st_in[0], st_in[1], t_static, st_in[2])[-1])
                side = torch.cuda.Stream(dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    call()                                  # warm-up off the capturing stream (allocator, lazy inits)
                torch.cuda.current_stream(dev).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        out_static = call()
                except RuntimeError as err:
                    # as Engine.train_step_graph: say so and continue with eager launches (the flag is there for callers)
                    import sys
                    print("ebfi_amd.engine: hipGraph capture of the inference step failed (%s); continuing with eager launches"
                          % str(err).splitlines()[0], file=sys.stderr, flush=True)
                    self.graph, self.graph_capture_failed, self.graph_capture_error = False, True, str(err).splitlines()[0]
                    torch.cuda.synchronize(dev)
                    eager()
                    return done()
                ent = self._captured[key] = (g, st_in, t_static, out_static)
            g, st_in, t_static, out_static = ent
            src = state if self.hoist else (frame, event, gtex)
            for dst, v in zip(st_in, src):
                if torch.is_tensor(dst):
                    if dst.shape[0] == v.shape[0]:
                        dst.copy_(v)
                    else:                                   # k replicas of the prefix's output, pass-major
                        dst.view((k,) + tuple(v.shape)).copy_(v.unsqueeze(0).expand((k,) + tuple(v.shape)))
            for i0 in range(0, n, k):
                chunk = timestamps[i0:i0 + k]
                t_static.copy_(t_tensor(chunk, k))
                g.replay()
                put(i0, out_static, len(chunk))
            return done()
