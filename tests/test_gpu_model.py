"""GPU parity of the whole frame-synthesis path: the product EVFIAutoEx (HIP FAC kernel inside)
vs fixtures produced by the reference's own modules, vs the functional CPU oracle, and one
training step vs the oracle's autograd.  Tolerance 1e-3 relative (BASELINE.json), checked in both parity-grade
conv modes: exact fp32 matrix cores and split-precision bf16x3."""
import ast
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import loss_ref, model_ref  # noqa: E402
from ebfi_amd.engine import DEFAULT_MODEL_ARGS as DEFAULT_ARGS_FULL  # noqa: E402

TOL = 1e-3


def _rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def fix(golden_dir):
    z = np.load(os.path.join(golden_dir, "model_small.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    cfg = ast.literal_eval(str(z["cfg"]))
    return z, sd, cfg


@pytest.fixture(params=["fp32", "bf16x3"])
def mode(request):
    from ebfi_amd import conv
    conv.set_compute_dtype(request.param)
    yield request.param
    conv.set_compute_dtype("fp32")


def _net(cfg, sd, **over):
    from ebfi_amd.model import EVFIAutoEx
    net = EVFIAutoEx(**dict(cfg, **over))
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert all(k.startswith("ExposureDecision.") for k in unexpected)     # absent under UseGTEx
    return net.cuda().eval(), missing


def test_forward_vs_reference_fixture(fix, mode):
    z, sd, cfg = fix
    net, missing = _net(cfg, sd)
    assert not missing
    c = lambda k: torch.from_numpy(z[k]).cuda()
    ev = c("in.Event").view(2, -1, 32, 40)
    with torch.no_grad():
        ex = net.ExposureDecision(ev, c("in.Blurry"))
        assert _rel(ex, z["mid.Ex"]) < TOL
        pe = net.ResidualControl(c("mid.EventFeat"), c("mid.Ex"), c("in.T"))
        assert _rel(pe, z["mid.ResidualControl"]) < TOL
        pf = net.Modification(c("mid.FrameFeat"), c("mid.ResidualControl"))      # HIP FAC inside
        assert _rel(pf, z["mid.Modification"]) < TOL
        det = net.Detail(img0=c("in.Frame"), img1=c("out.Sharp"))
        assert _rel(det, z["mid.Detail"]) < TOL
    net2, _ = _net(cfg, sd, UseGTEx=True)
    with torch.no_grad():
        sharp, final = net2(c("in.Frame"), c("in.Event"), c("in.T"), c("in.GTEx"))
        assert _rel(sharp, z["gtex.Sharp"]) < TOL and _rel(final, z["gtex.Final"]) < TOL
        so, fo = net2(c("odd.Frame"), c("odd.Event"), c("in.T")[:1], c("in.GTEx")[:1])
        assert so.shape[-2:] == (27, 37)
        assert _rel(so, z["odd.Sharp"]) < TOL and _rel(fo, z["odd.Final"]) < TOL


def test_gradients_vs_reference_fixture(fix, mode):
    z, sd, cfg = fix
    net, _ = _net(cfg, sd, UseGTEx=True)
    net.train()
    c = lambda k: torch.from_numpy(z[k]).cuda()
    sharp, final = net(c("in.Frame"), c("in.Event"), c("in.T"), c("in.GTEx"))
    ((sharp * c("gtex.wS")).sum() + (final * c("gtex.wF")).sum()).backward()
    n = 0
    for name, p in net.named_parameters():
        key = "grad." + name
        if key in z.files:
            assert _rel(p.grad, z[key]) < TOL, name
            n += 1
    assert n > 80


def test_full_width_forward_vs_oracle(mode):
    """config 1 of BASELINE.json (single 128x128 sample, default widths) against the CPU oracle,
    including the device Frame2Lap inside forward (RGBLap branch)."""
    from ebfi_amd.engine import DEFAULT_MODEL_ARGS, synthetic_batch
    from ebfi_amd.model import EVFIAutoEx
    torch.manual_seed(0)
    net = EVFIAutoEx(**DEFAULT_MODEL_ARGS)
    with torch.no_grad():           # O(1) activations instead of the x0.1 default init (Sharp == 0.5)
        for p in net.parameters():
            if p.dim() > 1:
                p.copy_(torch.randn_like(p) * (1.2 / p[0].numel() ** 0.5))
            else:
                p.add_(0.05 * torch.randn_like(p))
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    frame, event, t, gtex, _ = synthetic_batch(1, 128, 128, device="cpu")
    ref_s, ref_f = model_ref.evfi_forward(sd, DEFAULT_MODEL_ARGS, frame, event, t)
    net = net.cuda().eval()
    with torch.no_grad():
        s, f = net(frame.cuda(), event.cuda(), t.cuda())
    assert ref_s.std() > 0.01
    assert _rel(s, ref_s) < TOL and _rel(f, ref_f) < TOL


def test_train_step_vs_oracle(fix, mode):
    """One Engine.train_step (fwd, Lap+census loss, bwd, Adam) vs the oracle's loss and autograd
    gradients on the small config."""
    from ebfi_amd.engine import Engine
    z, sd, cfg = fix
    cfg2 = dict(cfg, UseGTEx=True)
    eng = Engine(cfg2, device="cuda", precision=mode, lr=1e-4)
    eng.model.load_state_dict(sd, strict=False)
    torch.manual_seed(9)
    frame = torch.rand(2, 3, 64, 64)
    event = torch.poisson(torch.full((2, cfg["TB"], 2, 64, 64), 0.35))
    t, gtex, target = torch.rand(2, 1), torch.rand(2, 1) * 0.4 + 0.55, torch.rand(2, 3, 64, 64)
    sdo = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()
           if not k.startswith("ExposureDecision")}
    s, f = model_ref.evfi_forward(sdo, cfg2, frame, event, t, gtex)
    loss_ref_v = loss_ref.train_loss(s, f, target, iteration=0)
    loss_ref_v.backward()
    before = {k: v.detach().clone() for k, v in eng.model.state_dict().items()}
    loss = eng.train_step(frame.cuda(), event.cuda(), t.cuda(), gtex.cuda(), target.cuda())
    assert abs(loss.item() - loss_ref_v.item()) <= TOL * abs(loss_ref_v.item())
    for name, p in eng.model.named_parameters():
        assert _rel(p.grad, sdo[name].grad) < 5 * TOL, name       # grads still in the flat bucket
    moved = sum((eng.model.state_dict()[k] - before[k]).abs().sum().item() for k in before)
    assert moved > 0 and eng.iteration == 1


def test_gauss5_kernel_pair_vs_reference_conv():
    """The loss's GaussianConv on the native kernel: forward vs the depthwise reflect-padded conv of the
    reference (loss/restore.py:149-163), backward vs autograd of that conv (exact adjoint incl. borders)."""
    import torch.nn.functional as F
    from ebfi_amd.loss import GaussianConv
    torch.manual_seed(3)
    k1 = torch.tensor([1., 4., 6., 4., 1.])
    kern = (k1[:, None] * k1[None, :] / 256).repeat(3, 1, 1, 1)
    for (B, H, W, factor) in [(2, 16, 16, 1.0), (1, 5, 7, 4.0), (2, 64, 48, 4.0), (1, 3, 3, 1.0)]:
        x = torch.randn(B, 3, H, W, requires_grad=True)
        ref = F.conv2d(F.pad(x, (2, 2, 2, 2), mode="reflect"), factor * kern, groups=3)
        g = torch.randn_like(ref)
        ref.backward(g)
        xd = x.detach().cuda().requires_grad_()
        out = GaussianConv().cuda()(xd, factor)
        out.backward(g.cuda())
        assert _rel(out.detach(), ref.detach()) < 1e-5, (H, W)
        assert _rel(xd.grad, x.grad) < 1e-5, (H, W)


def test_laplacian_difference_pyramid_vs_operator_formulation():
    """csrc/laploss.hip: both Laplacian terms of a step as ONE pyramid of [sharp - target ; sharp_pre - target]
    (lap_i is linear) against the operator-by-operator TrainLoss on CPU tensors (the formulation pinned by
    tests/golden/loss_small.npz) and the oracle: value, and gradients of both predictions.  The L1 gradient is a sign:
    an element whose Laplacian difference is ~0 may flip, so gradients are compared by the share of deviating elements."""
    from ebfi_amd import _native as N
    from ebfi_amd.loss import TrainLoss
    torch.manual_seed(21)
    for (B, H, W, it, detail, accu) in [(2, 64, 64, 0, True, 1), (1, 32, 48, 20000, True, 2), (2, 48, 32, 0, False, 1),
                                        (1, 256, 256, 0, True, 1)]:     # (valid sizes: multiples of 16, >= 32)
        a, b, t = torch.rand(B, 3, H, W), torch.rand(B, 3, H, W), torch.rand(B, 3, H, W)
        a[:, :, :4, :4] = t[:, :, :4, :4]                       # an exactly matching patch: zero differences, zero sign
        ar, br = a.clone().requires_grad_(), b.clone().requires_grad_()
        ref = TrainLoss(detail)(br, ar, t, iteration=it, accu_step=accu)
        ref.backward()
        ad, bd = a.cuda().requires_grad_(), b.cuda().requires_grad_()
        N.prof_reset()
        N.prof_enable(True)
        out = TrainLoss(detail).cuda()(bd, ad, t.cuda(), iteration=it, accu_step=accu)
        out.backward()
        torch.cuda.synchronize()
        N.prof_enable(False)
        names = {k for k, v in N.prof_collect().items() if v[0] > 0}
        assert "lap_level" in names and "gauss5_fwd" not in names, names
        assert abs(out.item() - ref.item()) <= 2e-5 * abs(ref.item()), (H, W)
        if detail and H <= 64:
            orc = loss_ref.train_loss(b, a, t, iteration=it) / accu
            assert abs(out.item() - orc.item()) <= 2e-5 * abs(orc.item())
        for d, r in ((ad, ar), (bd, br)) if detail else ((ad, ar),):
            dev = (d.grad.cpu() - r.grad).abs() > 1e-4 * r.grad.abs().max()
            assert dev.float().mean().item() < 2e-3, (H, W, dev.sum().item())
        if not detail:
            assert bd.grad is None


def test_groupnorm_kernels_vs_torch_cpu():
    import torch.nn as nn
    from ebfi_amd.norm import group_norm
    torch.manual_seed(5)
    for (B, C, G, H, W) in [(2, 8, 4, 8, 12), (3, 64, 4, 32, 32), (1, 6, 3, 5, 4)]:
        gn = nn.GroupNorm(G, C)
        with torch.no_grad():
            gn.weight.copy_(torch.randn(C))
            gn.bias.copy_(torch.randn(C))
        x = (torch.randn(B, C, H, W) * 2 + 0.7).requires_grad_()
        ref = gn(x)
        g = torch.randn_like(ref)
        ref.backward(g)
        gnd = nn.GroupNorm(G, C).cuda()
        gnd.load_state_dict(gn.state_dict())
        xd = x.detach().cuda().requires_grad_()
        out = group_norm(xd, gnd)
        out.backward(g.cuda())
        assert _rel(out.detach(), ref.detach()) < 1e-5
        assert _rel(xd.grad, x.grad) < 2e-5
        assert _rel(gnd.weight.grad, gn.weight.grad) < 2e-5 and _rel(gnd.bias.grad, gn.bias.grad) < 2e-5


def test_exposure_decision_head_vs_torch_cpu():
    """csrc/edhead.hip: cat([ev * sigmoid(AVGPool(GN(ev) * GN(bl))), bl], 1) (model_singleframe.py:66-72) from plane moments,
    gradients of ev, bl and the shared GroupNorm's affine parameters in closed form, against float64 autograd on the CPU."""
    import torch.nn as nn
    import torch.nn.functional as F
    from ebfi_amd import fused
    torch.manual_seed(15)
    for (B, C, G, H, W) in [(2, 8, 4, 8, 12), (3, 64, 4, 32, 32), (1, 6, 3, 6, 6), (2, 64, 4, 128, 160)]:
        gn = nn.GroupNorm(G, C)
        with torch.no_grad():
            gn.weight.copy_(torch.randn(C) * 0.5 + 1.0)
            gn.bias.copy_(torch.randn(C) * 0.3)
        ev = (torch.randn(B, C, H, W) * 1.5 + 0.4)
        bl = (torch.randn(B, C, H, W) * 0.7 - 0.2 + 0.5 * ev)        # correlated maps: the pooled product is not ~0
        gnd = nn.GroupNorm(G, C).double()
        gnd.load_state_dict(gn.state_dict())
        evr, blr = ev.double().requires_grad_(), bl.double().requires_grad_()
        atten = torch.sigmoid(F.adaptive_avg_pool2d(gnd(evr) * gnd(blr), 1))
        ref = torch.cat([evr * atten, blr], dim=1)
        g = torch.randn(B, 2 * C, H, W)
        ref.backward(g.double())
        gdev = nn.GroupNorm(G, C).cuda()
        gdev.load_state_dict(gn.state_dict())
        evd, bld = ev.cuda().requires_grad_(), bl.cuda().requires_grad_()
        out = fused.ed_head(evd, bld, gdev)
        assert out is not None
        out.backward(g.cuda())
        assert _rel(out.detach(), ref.detach().float()) < 1e-5, (C, H, W)
        assert _rel(evd.grad, evr.grad.float()) < 2e-5 and _rel(bld.grad, blr.grad.float()) < 2e-5, (C, H, W)
        assert _rel(gdev.weight.grad, gnd.weight.grad.float()) < 5e-5 and _rel(gdev.bias.grad, gnd.bias.grad.float()) < 5e-5


def test_native_flat_adam_matches_torch_adam():
    """csrc/optim.hip: the Adam update of the flat parameter buffer as one launch, against torch.optim.Adam on the CPU over
    several steps (odd element count: scalar tail), through a state_dict round trip, and against torch's own fused step."""
    import copy
    from ebfi_amd import _native as N
    from ebfi_amd.dp import FlatAdam, FlatGradBucket
    torch.manual_seed(4)
    net_a = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.Conv2d(5, 3, 1))     # 140 + 5 + 15 + 3 = 163 elements
    net_b = copy.deepcopy(net_a).cuda()
    opt_a = torch.optim.Adam(net_a.parameters(), lr=3e-3, betas=(0.9, 0.99), eps=1e-7)
    opt_b = FlatAdam(list(net_b.parameters()), lr=3e-3, betas=(0.9, 0.99), eps=1e-7)
    bucket = FlatGradBucket(net_b)
    x = torch.randn(2, 3, 8, 8)
    N.prof_reset()
    N.prof_enable(True)
    for it in range(5):
        opt_a.zero_grad()
        (net_a(x) * (1.0 + it)).square().sum().backward()
        opt_a.step()
        bucket.zero()
        (net_b(x.cuda()) * (1.0 + it)).square().sum().backward()
        opt_b.step(bucket.gather())
        if it == 2:                      # checkpoint interchange mid-run: torch's per-parameter layout both ways
            sd = opt_b.state_dict()
            assert float(sd["state"][0]["step"]) == 3.0
            opt_b = FlatAdam(list(net_b.parameters()), lr=3e-3, betas=(0.9, 0.99), eps=1e-7)
            opt_b.load_state_dict(sd)
            bucket = FlatGradBucket(net_b)
    torch.cuda.synchronize()
    N.prof_enable(False)
    assert N.prof_collect()["adam_flat"][0] == 5
    for pa, pb in zip(net_a.parameters(), net_b.parameters()):
        assert torch.allclose(pa, pb.cpu(), rtol=2e-6, atol=1e-7)
    sa, sb = opt_a.state_dict()["state"], opt_b.state_dict()["state"]
    for i in sa:
        # (the two sides' gradients come from different conv implementations: compare against the tensor's scale)
        assert _rel(sb[i]["exp_avg"], sa[i]["exp_avg"]) < 1e-5 and _rel(sb[i]["exp_avg_sq"], sa[i]["exp_avg_sq"]) < 1e-5


def test_native_gradient_gather_packs_like_cat_and_carries_the_guard_flag():
    """csrc/optim.hip grad_gather (FlatGradBucket.gather on the GPU): bit-identical to the concatenation of the per-parameter
    gradients for parameters of every alignment class (odd element counts shift the destination offsets, a sliced gradient
    shifts the source), zeros for a parameter without gradient, segments > 16384 elements, and the trailer
    [guard flag, 0, 0, 0]; the guarded Adam launch skips on the all-reduced flag alone and writes it back to guard[0]."""
    from ebfi_amd.dp import WIRE_PAD, FlatAdam, FlatGradBucket
    torch.manual_seed(11)
    shapes = [(3,), (7, 5), (1,), (40000,), (33, 3, 3), (2,), (16385,), (64, 64, 3, 3), (5,)]
    params = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(*sh, device="cuda")) for sh in shapes])
    bucket = FlatGradBucket(params)
    big = torch.randn(2 * 40000 + 3, device="cuda")
    grads = [torch.randn(*sh, device="cuda") for sh in shapes]
    grads[3] = big[3:40003]                                        # a source that is 12 bytes off a 16-byte boundary
    grads[5] = None                                                # no gradient: counts as zero
    for p, g in zip(params, grads):
        p.grad = g
    guard = torch.zeros(2, dtype=torch.int32, device="cuda")
    want = torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for p, g in zip(params, grads)])
    for flagged in (0, 1):
        for p, g in zip(params, grads):
            p.grad = g
        guard[0] = flagged
        flat = bucket.gather(guard)
        assert torch.equal(flat, want) and bucket.views_intact()
        assert bucket.wire.numel() == bucket.numel + WIRE_PAD
        assert bucket.wire[bucket.numel:].tolist() == [float(flagged), 0.0, 0.0, 0.0]
    # the optimiser launch takes the decision from the FLAG (the local guard word is clear: another rank raised it)
    opt = FlatAdam(list(params), lr=1e-2)
    bucket2 = FlatGradBucket(params)
    for p, g in zip(params, grads):
        p.grad = g
    guard.zero_()
    bucket2.gather(guard)
    before = opt.flat.detach().clone()
    bucket2.flag.fill_(0.5)                                        # = one of two ranks raised the guard, after the averaging
    opt.step(bucket2.flat, guard=guard, flag=bucket2.flag)
    assert torch.equal(before, opt.flat.detach()) and guard.tolist() == [1, 1]
    guard[0] = 0
    bucket2.flag.zero_()
    opt.step(bucket2.flat, guard=guard, flag=bucket2.flag)
    assert not torch.equal(before, opt.flat.detach()) and guard.tolist() == [0, 1]


def test_census_kernel_pair_vs_slice_formulation():
    from ebfi_amd.loss import Ternary
    torch.manual_seed(9)
    tern = Ternary()
    for (B, C, H, W) in [(2, 3, 20, 37), (1, 3, 16, 16), (1, 1, 7, 9), (2, 3, 64, 48)]:
        x = torch.rand(B, C, H, W, requires_grad=True)
        y = torch.rand(B, C, H, W)
        ref = tern(x, y)                       # CPU tensors: slice formulation (pinned by tests/golden/loss_small.npz)
        (ref * 3.0).backward()
        xd = x.detach().cuda().requires_grad_()
        out = tern.cuda()(xd, y.cuda())
        (out * 3.0).backward()
        tern.cpu()
        assert abs(out.item() - ref.item()) <= 2e-6 + 1e-5 * abs(ref.item())
        assert _rel(xd.grad, x.grad) < 5e-5


def test_scale_residual_cat_kernel_pair():
    from ebfi_amd.fused import scale_residual_cat
    torch.manual_seed(11)
    for (B, C, H, W) in [(2, 4, 8, 8), (3, 16, 12, 20), (1, 64, 32, 32)]:
        cpu = [torch.randn(B, C, H, W), torch.randn(B, C, 1, 1), torch.randn(B, C, H, W), torch.randn(B, C, 1, 1),
               torch.randn(B, C, H, W)]
        ref_in = [t.clone().requires_grad_() for t in cpu]
        a0, s0, a1, s1, x = ref_in
        ref = torch.cat([s0 * a0 + x, s1 * a1 + x], dim=1)
        g = torch.randn_like(ref)
        ref.backward(g)
        dev_in = [t.cuda().requires_grad_() for t in cpu]
        out = scale_residual_cat(*dev_in)
        out.backward(g.cuda())
        assert _rel(out.detach(), ref.detach()) < 1e-6
        for d, r in zip(dev_in, ref_in):
            assert d.grad.shape == r.grad.shape and _rel(d.grad, r.grad) < 1e-5


def test_full_size_modes_agree_and_batch_is_independent():
    """BASELINE size (B=8, 256x256, default widths): the split-precision mode against the exact fp32 mode on the same
    weights and inputs -- outputs within 1e-3, the packed gradient within 1e-2 in norm -- and sample independence:
    sample 3 of the batch equals the same sample run alone."""
    from ebfi_amd import conv
    from ebfi_amd.engine import DEFAULT_MODEL_ARGS, Engine, synthetic_batch
    torch.manual_seed(4)
    eng = Engine(DEFAULT_MODEL_ARGS, device="cuda", seed=4)
    with torch.no_grad():           # O(1) activations instead of the x0.1 default init (Sharp == 0.5 everywhere)
        for p in eng.model.parameters():
            if p.dim() > 1:
                p.copy_(torch.randn_like(p) * (1.2 / p[0].numel() ** 0.5))
    batch = synthetic_batch(8, 256, 256, device="cuda", seed=77)
    res = {}
    for mode in ("fp32", "bf16x3"):
        eng.precision = mode
        eng.bucket.zero()
        with eng._autocast():
            s, f = eng.model(*batch[:4])
            loss = eng.loss(s, f, batch[4], 0)
            loss.backward()
        res[mode] = (s.detach().clone(), f.detach().clone(), eng.bucket.gather().clone(), loss.detach().clone())
    a, b = res["fp32"], res["bf16x3"]
    assert a[0].std() > 0.01
    assert _rel(b[0], a[0]) < TOL and _rel(b[1], a[1]) < TOL
    assert abs(b[3].item() - a[3].item()) <= TOL * abs(a[3].item())
    assert ((b[2] - a[2]).norm() / a[2].norm()).item() < 1e-2
    conv.set_compute_dtype("bf16x3")
    try:
        with torch.no_grad():
            s1, f1 = eng.model(*[v[3:4] for v in batch[:4]])
    finally:
        conv.set_compute_dtype("fp32")
    assert _rel(s1, b[0][3:4]) < 1e-5 and _rel(f1, b[1][3:4]) < 1e-5
    # ... and that sample against the CPU oracle at the full 256x256 size (both modes, 1e-3)
    sd = {k: v.detach().cpu().clone() for k, v in eng.model.state_dict().items()}
    cpu = [v[3:4].cpu() for v in batch[:3]]
    ref_s, ref_f = model_ref.evfi_forward(sd, DEFAULT_MODEL_ARGS, *cpu)
    for mode in ("fp32", "bf16x3"):
        assert _rel(res[mode][0][3:4], ref_s) < TOL and _rel(res[mode][1][3:4], ref_f) < TOL, mode


def test_fp16_backward_takes_every_step_from_the_reference_initialisation():
    """The reference's x0.1 initialisation (model_util.py:16-36) leaves the deep stacks' activations at 1e-18; the first Adam
    update lifts them to 1e-4.  A one-step-old fp16 operand scale does not survive that jump (four optimiser steps were
    skipped by the overflow guard before the engine measured the first two steps' scales just in time): from the default
    initialisation, with a fresh batch every step, no step may be skipped -- eagerly and with the captured graph -- and the
    two launch modes must agree."""
    from ebfi_amd.engine import DEFAULT_MODEL_ARGS, Engine, synthetic_batch
    losses = {}
    for graph in (False, True):
        eng = Engine(DEFAULT_MODEL_ARGS, device="cuda", precision="bf16x3", graph=graph, seed=9, lr=1e-4)
        assert eng.book is not None and eng.calibration_steps == 2
        out = []
        for it in range(7):
            batch = synthetic_batch(2, 128, 128, device="cuda", seed=500 + it, on_device=True)
            out.append(eng.train_step(*batch).item())
        assert eng.book.skipped_steps() == 0, (graph, eng.book.skipped_steps())
        assert float(eng.optimizer.inner.state[eng.optimizer.flat]["step"]) == 7.0
        assert len(eng._graphs) == (1 if graph else 0)
        losses[graph] = out
    assert np.allclose(losses[False], losses[True], rtol=1e-5), losses


def _oracle_packed_gradient(sd, names, batch, iteration=0):
    sdo = {k: v.detach().cpu().clone().requires_grad_(k in names) for k, v in sd.items()}
    s, f = model_ref.evfi_forward(sdo, DEFAULT_ARGS_FULL, *batch[:3])
    # (sanity only: the output carries signal.  After the five lr = 1e-3 steps on random targets of test_benchmarked_step_vs_oracle
    #  the std of Sharp sits at 0.0097-0.0103 depending on the last digit of the trajectory -- profiles/r06/std_probe_benchmarked_step.log;
    #  the bounds that matter are relative to the loss and to the gradient norms)
    assert s.std() > 5e-3
    loss = loss_ref.train_loss(s, f, batch[4], iteration=iteration)
    loss.backward()
    return loss.item(), torch.cat([sdo[n].grad.reshape(-1) for n in names]), {n: sdo[n].numel() for n in names}


def _check_packed_gradient(flat, ref_flat, sizes, names, per_param=5e-2):
    assert flat.numel() == ref_flat.numel() == 5693543
    err = ((flat - ref_flat).norm() / ref_flat.norm()).item()
    assert err < 5e-3, err
    off, worst = 0, (0.0, None)
    for n in names:
        k = sizes[n]
        g, r = flat[off:off + k], ref_flat[off:off + k]
        off += k
        if r.norm() > 1e-6 * ref_flat.norm():          # (gradients that vanish against the rest: pure rounding)
            worst = max(worst, (((g - r).norm() / r.norm()).item(), n))
        if os.environ.get("EBFI_TEST_VERBOSE") == "1":
            print("   grad %-52s share %.3e  rel %.3e" % (n, (r.norm() / ref_flat.norm()).item(), ((g - r).norm() / r.norm().clamp_min(1e-30)).item()), flush=True)
    assert worst[0] < per_param, worst
    return err


@pytest.mark.parametrize("seed", [31, 77, 5])
def test_training_forward_within_tolerance(seed):
    """The forward pass AS THE TRAINING STEP RUNS IT -- default Engine in split precision: fp16 operand images written inside
    ResidualControl, the 128 -> 1600 KernelConv on fp16 operands writing fp16 filter planes (Engine(forward_f16='filters')) --
    at the benchmark size (B=8, 256x256, default widths): Sharp and Final within 1e-3 of the exact fp32 mode on the same weights
    and inputs, and sample 0 within 1e-3 of the CPU oracle.  (The losses and gradients of the same step are pinned by
    test_benchmarked_step_vs_oracle; this test pins the OUTPUTS the reduced-precision operands could move.)"""
    from ebfi_amd.engine import Engine, synthetic_batch
    eng = Engine(DEFAULT_ARGS_FULL, device="cuda", precision="bf16x3", seed=4)
    assert eng.book is not None and eng.book.forward_f16 == "filters"
    gen = torch.Generator(device="cpu").manual_seed(11 + seed)
    with torch.no_grad():                       # (the reference's x0.1 initialisation gives Sharp == 0.5 everywhere)
        for p in eng.model.parameters():
            if p.dim() > 1:
                p.copy_((torch.randn(p.shape, generator=gen) * (1.2 / p[0].numel() ** 0.5)).cuda())
            else:
                p.add_((0.05 * torch.randn(p.shape, generator=gen)).cuda())
    batch = synthetic_batch(8, 256, 256, device="cuda", seed=seed)
    out = {}
    for mode, passes in (("fp32", 1), ("bf16x3", 3)):     # (pass 1 calibrates the operand scales, 2-3 run on delayed ones)
        eng.precision = mode
        for _ in range(passes):
            eng.bucket.zero()
            eng.book.begin_step()
            with eng._autocast(), eng._bank(), eng._book() as book:
                s, f = eng.model(*batch[:4])
                eng.loss(s.float(), f.float(), batch[4], 0, 1).backward()
                if book is not None:
                    book.finish()
        out[mode] = (s.detach().clone(), f.detach().clone())
    assert eng.book.guard.tolist() == [0, 0]
    assert out["fp32"][0].std() > 0.01
    es, ef = _rel(out["bf16x3"][0], out["fp32"][0]), _rel(out["bf16x3"][1], out["fp32"][1])
    assert es < TOL and ef < TOL, (es, ef)
    sd = {k: v.detach().cpu() for k, v in eng.model.state_dict().items()}
    with torch.no_grad():
        ref_s, ref_f = model_ref.evfi_forward(sd, DEFAULT_ARGS_FULL, *[v[:1].cpu() for v in batch[:3]])
    es, ef = _rel(out["bf16x3"][0][:1], ref_s), _rel(out["bf16x3"][1][:1], ref_f)
    assert es < TOL and ef < TOL, (es, ef)
    del eng, out, batch, s, f
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def test_forward_f16_levels():
    """Engine(forward_f16=...): None and "filters" run the same kernels except for the KernelConv (conv_fwd_f16_ws/img_p16 instead of
    conv_fwd_bf16x3_ws/fwd_img with planar output); "all" -- the experiment switch that also runs ResidualControl's forward on fp16
    operand images -- stays a working path (its outputs within 5e-3 of the fp32 mode: OUTSIDE the 1e-3 parity bar, which is why it
    is never a default); anything else is refused."""
    from ebfi_amd import _native as N
    from ebfi_amd.engine import Engine, synthetic_batch
    with pytest.raises(ValueError):
        Engine(DEFAULT_ARGS_FULL, device="cuda", precision="bf16x3", forward_f16="everything")
    batch = synthetic_batch(2, 128, 128, device="cuda", seed=3)
    outs, profs = {}, {}
    for level in (None, "filters", "all"):
        eng = Engine(DEFAULT_ARGS_FULL, device="cuda", precision="bf16x3", seed=4, forward_f16=level)
        assert eng.book.forward_f16 == level
        gen = torch.Generator(device="cpu").manual_seed(21)
        with torch.no_grad():
            for p in eng.model.parameters():
                if p.dim() > 1:
                    p.copy_((torch.randn(p.shape, generator=gen) * (1.2 / p[0].numel() ** 0.5)).cuda())
        for k in range(3):                    # (pass 1 calibrates the operand scales, 2-3 run the image paths)
            eng.bucket.zero()
            eng.book.begin_step()
            if k == 2:
                N.prof_reset()
                N.prof_enable(True)
            with eng._autocast(), eng._bank(), eng._book() as book:
                s, f = eng.model(*batch[:4])
                eng.loss(s.float(), f.float(), batch[4], 0, 1).backward()
                book.finish()
        torch.cuda.synchronize()
        N.prof_enable(False)
        profs[level] = {k: v[0] for k, v in N.prof_collect().items() if v[0] > 0}
        outs[level] = (s.detach().clone(), f.detach().clone())
        assert eng.book.guard.tolist() == [0, 0]
        del eng
    assert profs[None].get("conv_fwd_f16_ws/img_p16", 0) == 0 and profs["filters"]["conv_fwd_f16_ws/img_p16"] == 1
    assert profs["all"]["conv_fwd_f16_ws/img_p16"] == 1
    # ResidualControl's 36 forward convolutions move from the split-precision kernel to the fp16 one only under "all"
    assert profs["filters"].get("conv_fwd_f16_ws/img_both", 0) == 0 and profs["all"].get("conv_fwd_f16_ws/img_both", 0) >= 12
    base = outs[None]
    assert base[0].std() > 0.01
    assert _rel(outs["filters"][0], base[0]) < TOL and _rel(outs["filters"][1], base[1]) < TOL
    assert _rel(outs["all"][0], base[0]) < 5e-3 and _rel(outs["all"][1], base[1]) < 5e-3


@pytest.mark.parametrize("B,seed", [(8, 31), (2, 77)])
def test_benchmarked_step_vs_oracle(B, seed):
    """The exact step bench.py times -- default widths, B=8 (and a second seed at B=2), split-precision forward, fp16 backward,
    weight bank, fused ResidualControl node, pre-activation FAC gradient, the whole forward + loss + backward replayed from a
    hipGraph -- against the CPU oracle (model_ref + loss_ref autograd) at 256x256: the loss within 1e-3, the PACKED gradient (all
    5.69 M parameters, the buffer the all-reduce and Adam consume) within 5e-3 in norm, every parameter's gradient within 5e-2
    of its own norm (the L1 / census terms have sign kinks: single pixels may flip, which is what separates the per-parameter
    from the packed bound).  Checked TWICE: on the first replay, and again after five more optimiser steps on fresh batches
    (lr 1e-3: the weights have moved by ~10 %, every fp16 operand scale in use is one step old) against the oracle run from the
    weights of that moment.  Weights are re-randomised: the reference's x0.1 initialisation gives Sharp == 0.5 everywhere."""
    from ebfi_amd import rc_fused
    from ebfi_amd.engine import Engine, synthetic_batch
    eng = Engine(DEFAULT_ARGS_FULL, device="cuda", precision="bf16x3", graph=True, seed=4, lr=1e-3)
    eng.calibration_steps = 0                   # straight to the captured graph (its two eager warm-up passes calibrate the
    #                                             fp16 operand scales): the step compared below is a REPLAY, what bench.py times
    gen = torch.Generator(device="cpu").manual_seed(11 + seed)
    with torch.no_grad():                       # (parameters are views of the optimiser's flat buffer: copy in place)
        for p in eng.model.parameters():
            if p.dim() > 1:
                p.copy_((torch.randn(p.shape, generator=gen) * (1.2 / p[0].numel() ** 0.5)).cuda())
            else:
                p.add_((0.05 * torch.randn(p.shape, generator=gen)).cuda())
    assert eng.bank is not None and eng.use_graph and rc_fused.fusable(eng.model.ResidualControl)
    calls = {"rc": 0, "refresh": 0}
    rc_orig, refresh_orig = rc_fused.residual_control, eng.bank.refresh

    def rc_counted(*a, **k):
        out = rc_orig(*a, **k)
        calls["rc"] += out is not None
        return out

    def refresh_counted():
        calls["refresh"] += 1
        return refresh_orig()
    rc_fused.residual_control, eng.bank.refresh = rc_counted, refresh_counted
    batch = synthetic_batch(B, 256, 256, device="cpu", seed=seed)
    names = [n for n, p in eng.model.named_parameters() if p.requires_grad]
    sd_first = {k: v.detach().cpu().clone() for k, v in eng.model.state_dict().items()}
    ref_loss, ref_flat, sizes = _oracle_packed_gradient(sd_first, names, batch)
    try:
        dev = [v.cuda() for v in batch]
        loss = eng.train_step(*dev)             # 2 eager warm-up passes, the capture, one replay, all-reduce (1 rank), Adam
        flat = eng.bucket.flat.detach().cpu().clone()      # the packed gradient Adam just consumed
        assert eng.book is not None and int(eng.book.guard[0].item()) == 0 and eng.book.skipped_steps() == 0   # (fp16 range guard)
        loss2 = eng.train_step(*dev)            # a second replay of the same graph (parameters have moved)
    finally:
        rc_fused.residual_control, eng.bank.refresh = rc_orig, refresh_orig
    torch.cuda.synchronize()
    # the fused node and the bank ran in the warm-up passes and were captured (3 Python-level passes), nothing re-captured
    assert calls["rc"] == 3 and calls["refresh"] == 3 and len(eng._graphs) == 1, calls
    assert abs(loss.item() - ref_loss) <= TOL * abs(ref_loss), (loss.item(), ref_loss)
    assert loss2.item() != loss.item()          # the replay picked up the optimiser update through the bank refresh
    _check_packed_gradient(flat, ref_flat, sizes, names)
    # ---- five more optimiser steps on fresh batches, then the same comparison from the weights of that moment
    for k in range(4):
        eng.train_step(*synthetic_batch(B, 256, 256, device="cuda", seed=seed + 100 + k, on_device=True))
    sd_now = {k: v.detach().cpu().clone() for k, v in eng.model.state_dict().items()}
    moved = max(((sd_now[n] - sd_first[n]).norm() / sd_first[n].norm()).item() for n in names if sd_first[n].dim() > 1)
    assert moved > 1e-2, moved                  # the optimiser really moved the weights
    batch7 = synthetic_batch(B, 256, 256, device="cpu", seed=seed + 999)
    ref_loss7, ref_flat7, _ = _oracle_packed_gradient(sd_now, names, batch7)
    loss7 = eng.train_step(*[v.cuda() for v in batch7])
    flat7 = eng.bucket.flat.detach().cpu().clone()
    assert eng.book.skipped_steps() == 0 and len(eng._graphs) == 1
    assert abs(loss7.item() - ref_loss7) <= TOL * abs(ref_loss7), (loss7.item(), ref_loss7)
    # (round 4 ran this check at 1e-1: torch's reflection_pad2d / replication_pad2d backward accumulated with atomics and three
    # runs of one build gave 1.5e-2 ... 5.2e-2 for the worst parameter.  Both adjoints are deterministic gathers now
    # (test_training_step_is_bit_reproducible), so the bound is back at the 5e-2 of the first-replay check.)
    _check_packed_gradient(flat7, ref_flat7, sizes, names, per_param=5e-2)


def test_training_step_is_bit_reproducible():
    """Two fresh engines from the same seed take the same four optimiser steps (eager calibration steps, capture, replay) and end
    with BIT-IDENTICAL packed gradients, losses and parameters: nothing in the step accumulates with atomics any more -- the
    adjoints of the two paddings that did (torch's replication_pad2d / reflection_pad2d backward) are own gather kernels with a
    fixed summation order (csrc/fac.hip CLAMP, csrc/imgops.hip reflect_pad_bwd).  Default widths at a reduced size."""
    from ebfi_amd.engine import Engine, synthetic_batch
    runs = []
    names = None
    for _ in range(2):
        eng = Engine(dict(step=3), device="cuda", seed=21, graph=True, precision="bf16x3")
        names = [n for n, p in eng.model.named_parameters() if p.requires_grad]
        sizes = [p.numel() for p in eng.bucket.params]
        losses, grads = [], []
        for k in range(4):
            losses.append(eng.train_step(*synthetic_batch(2, 128, 128, device="cuda", seed=500 + k)).item())
            grads.append(eng.bucket.flat.detach().clone())
        torch.cuda.synchronize()
        runs.append((losses, grads, eng.optimizer.flat.detach().clone()))
        assert eng.book is not None and eng.book.skipped_steps() == 0 and len(eng._graphs) == 1
        del eng
    (la, ga, pa), (lb, gb, pb) = runs
    for k in range(4):
        if not torch.equal(ga[k], gb[k]):         # say WHERE the two runs part: the first step and the parameters that differ
            bad, off = [], 0
            for n, sz in zip(names, sizes):
                a, b = ga[k][off:off + sz], gb[k][off:off + sz]
                if not torch.equal(a, b):
                    bad.append((n, ((a - b).abs().max() / a.abs().max().clamp_min(1e-30)).item()))
                off += sz
            same = [n for n in names if n not in {b[0] for b in bad}]
            raise AssertionError("step %d: %d of %d parameter gradients differ between two runs (worst %.1e); DIFFERENT: %s; IDENTICAL: %s"
                                 % (k, len(bad), len(names), max(b[1] for b in bad), " ".join(b[0] for b in bad), " ".join(same)))
    assert la == lb, (la, lb)
    assert torch.equal(pa, pb)


def test_scale_cat_stage_of_exposure_decision():
    from ebfi_amd.fused import scale_cat
    torch.manual_seed(12)
    for (B, C, H, W) in [(2, 4, 8, 8), (1, 64, 16, 32)]:
        cpu = [torch.randn(B, C, H, W), torch.randn(B, C, 1, 1), torch.randn(B, C, H, W)]
        ref_in = [t.clone().requires_grad_() for t in cpu]
        ref = torch.cat([ref_in[1] * ref_in[0], ref_in[2]], dim=1)
        g = torch.randn_like(ref)
        ref.backward(g)
        dev_in = [t.cuda().requires_grad_() for t in cpu]
        out = scale_cat(*dev_in)
        out.backward(g.cuda())
        assert _rel(out.detach(), ref.detach()) < 1e-6
        for d, r in zip(dev_in, ref_in):
            assert d.grad.shape == r.grad.shape and _rel(d.grad, r.grad) < 1e-5


def test_product_mean_stage():
    from ebfi_amd.fused import product_mean
    torch.manual_seed(13)
    for (B, C, H, W) in [(2, 4, 8, 8), (1, 64, 16, 32), (3, 5, 6, 10)]:
        a, b = torch.randn(B, C, H, W, requires_grad=True), torch.randn(B, C, H, W, requires_grad=True)
        ref = (a * b).mean(dim=(2, 3), keepdim=True)
        g = torch.randn_like(ref)
        ref.backward(g)
        ad, bd = a.detach().cuda().requires_grad_(), b.detach().cuda().requires_grad_()
        out = product_mean(ad, bd)
        out.backward(g.cuda())
        assert out.shape == ref.shape and _rel(out.detach(), ref.detach()) < 1e-5
        assert _rel(ad.grad, a.grad) < 1e-6 and _rel(bd.grad, b.grad) < 1e-6


def test_hd_config5_forward_vs_oracle_and_batch_independence():
    """BASELINE.json config 5 (B=8, 720x1280 inference, default widths, default bf16x3 convs): one HD sample against the CPU
    oracle with re-randomised weights (the x0.1 default init gives Sharp == 0.5: no signal), then the full B=8 batch
    through the HIP path with that sample at index 5 -- it must come out as when run alone (the per-sample descriptors and
    64-bit plane arithmetic hold across > 2^31-element tensors: the filter tensor is [8,1600,360,640] = 2.95e9 elements,
    where the reference's own launcher overflows, KernelConv2D_kernel.cu:166-167)."""
    from ebfi_amd import conv
    from ebfi_amd.engine import DEFAULT_MODEL_ARGS, synthetic_batch
    from ebfi_amd.model import EVFIAutoEx
    import gc
    gc.collect()                            # (earlier tests of the process: engines in reference cycles still hold device tensors)
    torch.cuda.empty_cache()
    held_before = torch.cuda.memory_allocated()      # whatever is still alive is not this test's: subtracted from the peaks below
    torch.manual_seed(6)
    net = EVFIAutoEx(**DEFAULT_MODEL_ARGS)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() > 1:
                p.copy_(torch.randn_like(p) * (1.2 / p[0].numel() ** 0.5))
            else:
                p.add_(0.05 * torch.randn_like(p))
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    frame, event, t, _, _ = synthetic_batch(8, 720, 1280, device="cpu", seed=55)
    one = (frame[5:6], event[5:6], t[5:6])
    ref_s, ref_f = model_ref.evfi_forward(sd, DEFAULT_MODEL_ARGS, *one)
    assert ref_s.std() > 0.01
    net = net.cuda().eval()
    conv.set_compute_dtype("bf16x3")
    try:
        with torch.no_grad():
            s1, f1 = net(*[v.cuda() for v in one])
            assert _rel(s1, ref_s) < TOL and _rel(f1, ref_f) < TOL
            dev = (frame.cuda(), event.cuda(), t.cuda())
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            s8, f8 = net(*dev)
            torch.cuda.synchronize()
            peak_unfused = torch.cuda.max_memory_allocated() - held_before
            # with an inference weight bank the KernelConv -> FAC pair runs as one kernel and the
            # [8,1600,360,640] filter tensor (11.8 GB) is never allocated
            from ebfi_amd import weightbank
            bank = weightbank.build_for(net, inference=True)
            bank.refresh()
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            with bank.active():
                s8f, f8f = net(*dev)
            torch.cuda.synchronize()
            peak_fused = torch.cuda.max_memory_allocated() - held_before
    finally:
        conv.set_compute_dtype("fp32")
    # (measured: 17.9 GB unfused -- the filter tensor plus the 128-channel input next to it -- against 12.9 GB fused, where the
    # peak moves to the full-resolution maps of ExposureDecision)
    assert peak_unfused - peak_fused > 4.0e9 and peak_fused < 14.0e9, (peak_unfused, peak_fused)
    # (the bank run also takes the merged / grouped ResidualControl convolutions: another summation order, 2e-5 measured)
    assert _rel(s8f, s8) < 1e-4 and _rel(f8f, f8) < 1e-4
    assert s8.shape == (8, 3, 720, 1280) and torch.isfinite(f8).all()
    # (global means over 921 600 pixels are reduced in a batch-dependent partition: 2e-5 measured)
    assert _rel(s8[5:6], s1) < 1e-4 and _rel(f8[5:6], f1) < 1e-4
    assert _rel(s8[5:6], ref_s) < TOL and _rel(f8[5:6], ref_f) < TOL


def test_weight_bank_path_equals_per_call_packing(fix):
    """ebfi_amd.weightbank: one table-driven pack launch per step for all conv weights (plain, folded depth-2 3-D, biases)
    instead of a pack launch inside every conv call.  Same kernels on the same images: forward outputs must be
    bit-identical with and without the bank, parameter gradients equal, and a parameter update must be picked up."""
    from ebfi_amd import conv, weightbank
    from ebfi_amd import _native as N
    z, sd, cfg = fix
    net, _ = _net(cfg, sd)
    net.train()
    c = lambda k: torch.from_numpy(z[k]).cuda()
    inputs = (c("in.Frame"), c("in.Event"), c("in.T"))
    conv.set_compute_dtype("bf16x3")
    try:
        def run(bank):
            net.zero_grad(set_to_none=True)
            N.prof_reset()
            N.prof_enable(True)
            if bank is not None:
                bank.ensure_fresh()
                with bank.active():
                    s, f = net(*inputs)
                    (s.square().sum() + f.sum()).backward()
            else:
                s, f = net(*inputs)
                (s.square().sum() + f.sum()).backward()
            torch.cuda.synchronize()
            N.prof_enable(False)
            prof = N.prof_collect()
            return s.detach().clone(), f.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters()}, prof
        s0, f0, g0, prof0 = run(None)
        bank = weightbank.build_for(net)
        assert {k for _, k in bank.sites} == {"id", "conv3d", "convT3d", "rcA", "rcB", "fuse_d2"}
        s1, f1, g1, prof1 = run(bank)
        # (not bit-equal since the bank's feature_fuse image sums its 64 input channels in the decoder's memory order,
        # c*2 + d, where the per-call form sums them in cat(unbind) order: the same products, another fp32 summation order)
        assert _rel(s1, s0) < 2e-6 and _rel(f1, f0) < 2e-6
        for n in g0:     # (PyTorch's replication / reflection pad backward accumulate with atomics: last-bit run-to-run noise)
            assert _rel(g1[n], g0[n]) < 5e-5, n
        assert prof0["conv_pack_w_bf16"][0] > 50 and "conv_pack_w_bf16" not in prof1 and prof1["pack_table_bf16"][0] == 1
        with torch.no_grad():                      # an in-place update (what the optimiser does) must invalidate the images
            for p in net.parameters():
                p.mul_(1.01)
        s2, _, _, prof2 = run(bank)
        assert prof2["pack_table_bf16"][0] == 1 and not torch.equal(s2, s1)
        s3, _, _, _ = run(None)
        assert _rel(s2, s3) < 2e-6 and _rel(s2, s1) > 1e-4
    finally:
        conv.set_compute_dtype("fp32")


def test_fused_residual_control_equals_layerwise():
    """ebfi_amd.rc_fused: ResidualControl as one hand-scheduled node (merged first layers, grouped second layers, gradients
    handed down as pre-activation gradients through the data-gradient epilogues) against the layer-by-layer autograd form
    of the same module on the same split-precision kernels (default width 64, ragged 20x36 maps): output, input gradient,
    scalar-input gradients and every parameter gradient; the layer-by-layer form against the CPU oracle."""
    from ebfi_amd import conv, rc_fused, weightbank
    from ebfi_amd import _native as N
    from ebfi_amd.model import ResidualControl
    torch.manual_seed(31)
    step = 3
    rc = ResidualControl(BLinch=1, Tinch=1, Basech=64, step=step)
    with torch.no_grad():
        for p in rc.parameters():
            p.copy_(torch.randn_like(p) * (1.0 / p[0].numel() ** 0.5) if p.dim() > 1 else 0.1 * torch.randn_like(p))
    x0, ex0, t0 = torch.randn(2, 64, 20, 36), torch.rand(2, 1), torch.rand(2, 1)
    sd = {"ResidualControl." + k: v.detach().clone() for k, v in rc.state_dict().items()}
    ref = model_ref.residual_control(sd, "ResidualControl", x0, ex0, t0, step)
    rc = rc.cuda().train()
    gout = torch.randn(2, 64, 20, 36).cuda()
    conv.set_compute_dtype("bf16x3")
    try:
        def run(bank):
            rc.zero_grad(set_to_none=True)
            x, ex, t = x0.cuda().requires_grad_(), ex0.cuda().requires_grad_(), t0.cuda().requires_grad_()
            N.prof_reset()
            N.prof_enable(True)
            if bank is not None:
                bank.ensure_fresh()
                with bank.active():
                    assert rc_fused.usable(rc, x) and rc_fused.sites_of(rc) is not None
                    y = rc(x, ex, t)
                    y.backward(gout)
            else:
                y = rc(x, ex, t)
                y.backward(gout)
            torch.cuda.synchronize()
            N.prof_enable(False)
            return y.detach(), x.grad, ex.grad, t.grad, {n: p.grad.clone() for n, p in rc.named_parameters()}, N.prof_collect()
        y0, gx0, ge0, gt0, gp0, prof0 = run(None)
        assert _rel(y0, ref) < TOL
        y1, gx1, ge1, gt1, gp1, prof1 = run(weightbank.build_for(rc))
        assert y0.abs().max() > 1e-3 and _rel(y1, y0) < 1e-5
        assert _rel(gx1, gx0) < 3e-5 and _rel(ge1, ge0) < 3e-5 and _rel(gt1, gt0) < 3e-5
        for n in gp0:
            assert _rel(gp1[n], gp0[n]) < 5e-5, n
        # 3 convolutions forward and 3 data gradients per round on one kernel, 3 weight gradients per round
        assert prof1["conv_fwd_bf16x3_db/fwd"][0] == 6 * step and prof1["conv_wgrad_x3_ws"][0] == 3 * step
        assert prof0["conv_fwd_bf16x3_db/fwd"][0] == 5 * step and prof0["conv_wgrad_x3_ws"][0] == 5 * step
    finally:
        conv.set_compute_dtype("fp32")


def test_se_gate_through_the_pixel_shuffle_vs_torch_cpu():
    """csrc/segate.hip (round 6): the gate of an up-convolution stage reads the folded transposed convolution's output THROUGH the
    pixel shuffle.  (a) the gate alone on a random unshuffled tensor against pixel_shuffle + the reference formulation on the CPU;
    (b) a whole upConv3D stage (ConvTranspose3d (3,4,4)/(1,2,2) -> SEGating -> LeakyReLU(0.2): model_singleframe.py:200-221)
    against torch's own modules on the CPU, output and all gradients; and the stage takes the new kernels."""
    import torch.nn as nn
    import torch.nn.functional as F
    from ebfi_amd import _native as N
    from ebfi_amd import fold3d
    from ebfi_amd.model import SEGating, upConv3D
    torch.manual_seed(29)
    for (B, C, h, w, act) in [(2, 16, 6, 10, 0.2), (1, 8, 5, 8, None), (3, 4, 64, 96, 0.2), (1, 32, 3, 2, 0.2), (70, 4, 2, 4, None)]:
        gate = SEGating(C)
        with torch.no_grad():
            gate.attn_layer[0].weight.copy_(torch.randn_like(gate.attn_layer[0].weight) * 0.5)
            gate.attn_layer[0].bias.copy_(torch.randn(C) * 0.3)
        y2 = torch.randn(B, 8 * C, h, w, requires_grad=True)
        x = F.pixel_shuffle(y2, 2).view(B, C, 2, 2 * h, 2 * w)
        y = x * torch.sigmoid(gate.attn_layer[0](gate.pool(x)))
        if act is not None:
            y = F.leaky_relu(y, act)
        g = torch.randn_like(y)
        y.backward(g)
        gd = SEGating(C).cuda()
        gd.load_state_dict(gate.state_dict())
        y2d = y2.detach().cuda().requires_grad_()
        conv1 = gd.attn_layer[0]
        yd = fold3d._SEGateShuffled.apply(y2d, conv1.weight, conv1.bias, 0 if act is None else 1, act or 0.0)
        yd.backward(g.cuda())
        assert yd.shape == y.shape and _rel(yd.detach(), y.detach()) < 1e-5
        assert _rel(y2d.grad, y2.grad) < 2e-5
        assert _rel(conv1.weight.grad, gate.attn_layer[0].weight.grad) < 5e-5
        assert _rel(conv1.bias.grad, gate.attn_layer[0].bias.grad) < 5e-5
    for (B, Ci, Co, H, W) in [(2, 16, 8, 6, 10), (1, 24, 16, 16, 32)]:
        up = upConv3D(Ci, Co, kernel_size=(3, 4, 4), stride=(1, 2, 2), padding=(1, 1, 1), upmode="transpose", bn=False)
        with torch.no_grad():
            up.upconv[1].attn_layer[0].weight.mul_(3.0)
            up.upconv[1].attn_layer[0].bias.copy_(torch.randn(Co) * 0.3)
        x = torch.randn(B, Ci, 2, H, W, requires_grad=True)
        y = F.leaky_relu(up.upconv(x), 0.2)          # torch's ConvTranspose3d + pool + 1x1x1 conv + sigmoid on the CPU
        g = torch.randn_like(y)
        y.backward(g)
        ud = upConv3D(Ci, Co, kernel_size=(3, 4, 4), stride=(1, 2, 2), padding=(1, 1, 1), upmode="transpose", bn=False).cuda()
        ud.load_state_dict(up.state_dict())
        xd = x.detach().cuda().requires_grad_()
        N.prof_reset()
        N.prof_enable(True)
        yd = ud(xd, 0.2)
        yd.backward(g.cuda())
        torch.cuda.synchronize()
        N.prof_enable(False)
        prof = N.prof_collect()
        assert prof["se_gate_fwd/shuffle"][0] == 1 and prof["se_gate_bwd/shuffle"][0] == 1
        assert _rel(yd.detach(), y.detach()) < 2e-5
        assert _rel(xd.grad, x.grad) < 5e-5
        for (n, p), (_, q) in zip(ud.named_parameters(), up.named_parameters()):
            assert _rel(p.grad, q.grad) < 1e-4, n


def test_se_gate_kernels_vs_torch_cpu():
    """csrc/segate.hip: SEGating (+ residual + ReLU / LeakyReLU) against the reference formulation
    x * sigmoid(Conv3d_1x1x1(AdaptiveAvgPool3d(1)(x))) on the CPU (resnet_3D.py:89-105,:137-141): output and all gradients."""
    import torch.nn as nn
    import torch.nn.functional as F
    from ebfi_amd.model import SEGating
    torch.manual_seed(23)
    # (the 64 x 96 and 128 x 128 planes are cut into 2 / 4 slices by the sliced plane reductions of round 3)
    for (B, C, H, W, res, act) in [(2, 16, 8, 12, True, 0.0), (3, 24, 6, 10, False, 0.2), (1, 8, 5, 8, False, None), (2, 64, 4, 4, True, 0.0),
                                   (2, 16, 64, 96, True, 0.0), (1, 16, 128, 128, False, 0.2),
                                   (70, 8, 4, 4, False, None),       # more than 64 samples: grad_W / grad_b accumulated in rounds
                                   (2, 288, 2, 2, True, 0.0)]:       # more channels than threads in the gate prologue
        gate = SEGating(C)
        with torch.no_grad():
            gate.attn_layer[0].weight.copy_(torch.randn_like(gate.attn_layer[0].weight) * 0.5)
            gate.attn_layer[0].bias.copy_(torch.randn(C) * 0.3)
        x = torch.randn(B, C, 2, H, W, requires_grad=True)
        r = torch.randn(B, C, 2, H, W, requires_grad=True) if res else None
        y = x * torch.sigmoid(gate.attn_layer[0](gate.pool(x)))
        if r is not None:
            y = y + r
        if act is not None:
            y = F.leaky_relu(y, act)
        g = torch.randn_like(y)
        y.backward(g)
        gd = SEGating(C).cuda()
        gd.load_state_dict(gate.state_dict())
        xd = x.detach().cuda().requires_grad_()
        rd = r.detach().cuda().requires_grad_() if r is not None else None
        yd = gd(xd, rd, act)
        yd.backward(g.cuda())
        assert _rel(yd.detach(), y.detach()) < 1e-5
        assert _rel(xd.grad, x.grad) < 2e-5
        if r is not None:
            assert _rel(rd.grad, r.grad) < 1e-5
        assert _rel(gd.attn_layer[0].weight.grad, gate.attn_layer[0].weight.grad) < 5e-5
        assert _rel(gd.attn_layer[0].bias.grad, gate.attn_layer[0].bias.grad) < 5e-5
