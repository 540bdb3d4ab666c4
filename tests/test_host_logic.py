"""Host-side logic on CPU: module tree / state_dict contract, import-path shims, loss module vs
the reference fixture, pad/crop helper, and the data-parallel gradient bucket over gloo (world 2)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ebfi_amd.engine import DEFAULT_MODEL_ARGS
from ebfi_amd.loss import LaplacianLoss, Ternary, TrainLoss
from ebfi_amd.model import CropSize, EVFIAutoEx


def test_state_dict_matches_reference_names(golden_dir):
    ref = [l.split() for l in open(os.path.join(golden_dir, "state_dict_default.txt"))]
    net = EVFIAutoEx(**DEFAULT_MODEL_ARGS)
    mine = [[k, "x".join(map(str, v.shape))] for k, v in net.state_dict().items()]
    assert mine == ref
    assert sum(p.numel() for p in net.parameters()) == 5693543


def test_reference_import_paths():
    from models.DCNv2.dcn_v2 import DCN, DCN_sep, DCNv2, dcn_v2_conv          # noqa: F401
    from models.FAC.kernelconv2d.KernelConv2D import KernelConv2D, KernelConv2DFunction  # noqa: F401
    from models.Ours.model_singleframe import EVFIAutoEx as E2
    from myutils.utils import Frame2DCP, Frame2Lap, reduce_tensor               # noqa: F401
    from dataloader.encodings import events_to_stack                           # noqa: F401
    assert E2 is EVFIAutoEx
    m = DCN_sep(4, 6, 3, 1, 1, deformable_groups=2)
    assert sorted(n for n, _ in m.named_parameters()) == ["bias", "conv_offset_mask.bias",
                                                          "conv_offset_mask.weight", "weight"]
    assert m.conv_offset_mask.weight.abs().sum() == 0 and m.conv_offset_mask.out_channels == 2 * 3 * 9
    assert m.bias.abs().sum() == 0 and m.weight.abs().max() <= 1.0 / (4 * 9) ** 0.5


def test_model_variants_construct():
    small = dict(DEFAULT_MODEL_ARGS, FrameBasech=8, EventBasech=8, InterCH=8, TB=4, step=2, channels=[4, 4, 8, 8])
    a = EVFIAutoEx(**dict(small, UseGTEx=True))
    assert not hasattr(a, "ExposureDecision")
    b = EVFIAutoEx(**dict(small, DetailEnabled=False))
    assert not hasattr(b, "Detail")
    assert "Trainable parameters" in str(b)
    # layer4 has no downsample when channels[2] == channels[3] (resnet_3D.py:259)
    assert a.Detail.encoder.layer4[0].downsample is None and a.Detail.encoder.layer3[0].downsample is not None


def test_cropsize_matches_oracle():
    from oracle import model_ref
    x = torch.arange(2 * 27 * 37, dtype=torch.float32).view(1, 2, 27, 37)
    c = CropSize(37, 27, {"h": 8, "w": 8})
    padded = c.pad(x)
    assert padded.shape[-2:] == (32, 40)
    assert torch.equal(padded, model_ref.pad_to_multiple(x, 27, 37))
    assert torch.equal(c.crop(padded), x)


def test_loss_matches_reference_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "loss_small.npz"))
    x = torch.from_numpy(z["x"]).requires_grad_()
    y = torch.from_numpy(z["y"])
    lap, cen = LaplacianLoss()(x, y), Ternary()(x, y)
    assert abs(lap.item() - float(z["lap"])) <= 1e-5 * float(z["lap"])
    assert abs(cen.item() - float(z["census"])) <= 1e-5
    (lap + cen).backward()
    assert (x.grad - torch.from_numpy(z["grad_x"])).abs().max() <= 1e-4 * np.abs(z["grad_x"]).max()
    tl = TrainLoss()
    a = tl(x.detach(), y, y, iteration=0)
    b = tl(x.detach(), y, y, iteration=20000)
    assert abs(a.item() - (lap.item() + cen.item())) < 1e-2 and abs(b.item() - 0.1 * (lap.item() + cen.item())) < 1e-2


# ----------------------------------------------------------------------------- data parallel (gloo, world 2)
def _dp_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ebfi_amd.dp import FlatGradBucket, broadcast_parameters, reduce_tensor
    torch.manual_seed(100 + rank)                       # different init per rank on purpose
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))
    broadcast_parameters(net, 0)
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    bucket = FlatGradBucket(net)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    torch.manual_seed(7)
    data = torch.randn(4, 3, 8, 8)                      # global batch 4 -> 2 per rank
    mine = data[rank * 2:(rank + 1) * 2]
    bucket.zero()
    net(mine).pow(2).sum().backward()                   # sum-reduced loss, like the reference's
    local = bucket.gather().clone()
    assert bucket.views_intact()
    bucket.all_reduce_mean()
    opt.step()
    w1 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    red = reduce_tensor(torch.tensor([float(rank + 1)]))
    torch.save((rank, w0, local, bucket.flat.clone(), w1, red), os.path.join(outdir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_equals_large_batch_math(tmp_path):
    ctx = mp.get_context("spawn")
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    res = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(2)]
    (_, w0a, la, ga, w1a, ra), (_, w0b, lb, gb, w1b, rb) = res
    assert torch.equal(w0a, w0b)                        # broadcast made the replicas identical
    assert torch.allclose(ga, (la + lb) / 2, atol=1e-6) and torch.equal(ga, gb)
    assert torch.equal(w1a, w1b)                        # replicas stay in lock-step after the step
    assert ra.item() == 1.5 and rb.item() == 1.5
    # single-process reference: the same 4 samples at once; loss is a SUM so DP mean = grad / world
    torch.manual_seed(100)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))
    torch.manual_seed(7)
    data = torch.randn(4, 3, 8, 8)
    net(data).pow(2).sum().backward()
    g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert torch.allclose(ga, g / 2, atol=1e-5)


def _guard_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ebfi_amd.dp import FlatAdam, FlatGradBucket, broadcast_parameters
    torch.manual_seed(100 + rank)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))
    broadcast_parameters(net, 0)
    bucket, opt = FlatGradBucket(net), FlatAdam(list(net.parameters()), lr=1e-2)
    guard = torch.zeros(2, dtype=torch.int32)           # f16scale.ScaleBook.guard: [flag of this step, skipped steps]
    snaps = []
    calls = []
    real_all_reduce = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(a[0].numel()), real_all_reduce(*a, **k))[1]
    for step, raise_on in enumerate((None, 1, None)):   # step 1: ONLY rank 1 sees an overflow
        torch.manual_seed(7 + step)
        data = torch.randn(4, 3, 8, 8)
        bucket.zero()
        net(data[rank * 2:(rank + 1) * 2]).pow(2).sum().backward()
        guard[0] = 1 if raise_on == rank else 0         # (what f16_scales_finish does on the rank whose operand overflowed)
        bucket.gather(guard)                            # Engine.train_step's order: the flag rides in the wire buffer
        bucket.reduce_mean_packed()                     # the step's ONE collective
        opt.step(bucket.flat, guard=guard, flag=bucket.flag)
        st = opt.inner.state.get(opt.flat, {})
        snaps.append((opt.flat.detach().clone(), int(guard[0]), int(guard[1]),
                      st["exp_avg"].clone() if st else None, float(st["step"]) if st else 0.0))
    dist.all_reduce = real_all_reduce
    assert calls == [bucket.numel + 4] * 3, calls       # exactly one all-reduce per step, gradients + flag in one message
    torch.save(snaps, os.path.join(outdir, "g%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_overflow_guard_raised_on_one_rank_skips_the_update_on_every_rank(tmp_path):
    """Data parallelism with the fp16 backward: the guard flag travels INSIDE the gradient all-reduce (one element of the
    wire buffer, SUM > 0 <=> raised somewhere; one collective per step) and the guarded optimiser step reads it, so a flag
    raised on ONE rank skips the update on BOTH (parameters, moments and step count untouched, the skip
    counted) and the replicas stay bit-identical; the next clean step updates both."""
    ctx = mp.get_context("spawn")
    port = 29500 + ((os.getpid() + 137) % 500)
    procs = [ctx.Process(target=_guard_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    a, b = (torch.load(os.path.join(str(tmp_path), "g%d.pt" % r)) for r in range(2))
    for sa, sb in zip(a, b):
        assert torch.equal(sa[0], sb[0]) and sa[1:3] == sb[1:3] and sa[4] == sb[4]      # replicas in lock-step at every step
    (w0, f0, n0, m0, t0), (w1, f1, n1, m1, t1), (w2, f2, n2, m2, t2) = a
    assert (f0, n0, t0) == (0, 0, 1.0) and (f1, n1, t1) == (1, 1, 1.0) and (f2, n2, t2) == (0, 1, 2.0)
    assert torch.equal(w0, w1) and torch.equal(m0, m1)      # the flagged step changed nothing on either rank
    assert not torch.equal(w1, w2) and not torch.equal(m1, m2)


def test_flat_bucket_single_process():
    from ebfi_amd.dp import FlatGradBucket
    net = torch.nn.Linear(3, 2)
    b = FlatGradBucket(net)
    net(torch.ones(1, 3)).sum().backward()
    before = torch.cat([net.weight.grad.reshape(-1), net.bias.grad.reshape(-1)])
    b.all_reduce_mean()                                 # no process group: just packs
    assert b.views_intact() and torch.equal(before, b.flat) and b.flat.abs().sum() > 0
    b.flat.mul_(2.0)
    assert torch.equal(net.weight.grad.reshape(-1), before[:6] * 2)      # .grad are views of the flat buffer
    b.zero()
    assert net.weight.grad is None and net.bias.grad is None


def test_fold_index_tables_reproduce_the_tensor_folds():
    """The gather tables the GPU path uses for the Conv3d/ConvTranspose3d weight folds are derived from the
    tensor-op statement of the folds; applied with plain indexing they must reproduce it and its adjoint."""
    from ebfi_amd import fold3d
    torch.manual_seed(2)
    cases = [("conv3d", fold3d.fold_conv3d_weight, (4, 3, 3, 3, 3)), ("conv3d", fold3d.fold_conv3d_weight, (4, 3, 1, 1, 1)),
             ("conv3d", fold3d.fold_conv3d_weight, (2, 3, 3, 7, 7)), ("convT3d", fold3d.fold_conv_transpose3d_weight, (3, 2, 3, 4, 4)),
             ("rep2", fold3d._rep2, (5,)), ("rep8", fold3d._rep8, (3,))]
    for kind, fn, shape in cases:
        fwd, inv, R, oshape = fold3d._index_tables(kind, fn, shape, "cpu")
        w = torch.randn(*shape, dtype=torch.float64, requires_grad=True)
        ref = fn(w)
        zero = torch.zeros(1, dtype=torch.float64)
        got = torch.cat([w.detach().flatten(), zero])[fwd.long()].view(oshape)
        assert torch.equal(got, ref.detach()), kind
        g = torch.randn_like(ref)
        ref.backward(g)
        gw = torch.cat([g.flatten(), zero])[inv.long()].sum(-1).view(shape)
        assert inv.shape == (w.numel(), R) and torch.allclose(gw, w.grad, atol=1e-12), kind


def test_flat_adam_matches_torch_adam_and_speaks_its_checkpoint_layout():
    """FlatAdam (one fused update over a flat parameter buffer) against torch.optim.Adam on the same gradients, and
    checkpoint interchange in both directions in torch's per-parameter layout (the reference's checkpoint format)."""
    import copy
    from ebfi_amd.dp import FlatAdam, FlatGradBucket
    torch.manual_seed(3)
    net_a = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 1))
    net_b = copy.deepcopy(net_a)
    names = [n for n, _ in net_a.named_parameters()]
    opt_a = torch.optim.Adam(net_a.parameters(), lr=1e-2)
    opt_b = FlatAdam(list(net_b.parameters()), lr=1e-2)
    assert opt_b.views_intact() and [n for n, _ in net_b.named_parameters()] == names
    bucket = FlatGradBucket(net_b)
    x = torch.randn(2, 3, 8, 8)
    for _ in range(3):
        opt_a.zero_grad()
        net_a(x).square().sum().backward()
        opt_a.step()
        bucket.zero()
        net_b(x).square().sum().backward()
        opt_b.step(bucket.gather())
    for pa, pb in zip(net_a.parameters(), net_b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-7)
    # FlatAdam -> torch.optim.Adam
    sd = opt_b.state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == 4 and sd["state"][0]["exp_avg"].shape == (4, 3, 3, 3)
    opt_c = torch.optim.Adam(net_a.parameters(), lr=1e-2)
    opt_c.load_state_dict(sd)
    # torch.optim.Adam -> FlatAdam, then one more identical step on both sides
    net_d = copy.deepcopy(net_a)
    opt_d = FlatAdam(list(net_d.parameters()), lr=1e-2)
    opt_d.load_state_dict(opt_a.state_dict())
    bucket_d = FlatGradBucket(net_d)
    opt_a.zero_grad()
    net_a(x).square().sum().backward()
    opt_a.step()
    net_d(x).square().sum().backward()
    opt_d.step(bucket_d.gather())
    for pa, pd in zip(net_a.parameters(), net_d.parameters()):
        assert torch.allclose(pa, pd, rtol=1e-6, atol=1e-7)


def test_weight_bank_tables_reproduce_the_per_call_packing():
    """ebfi_amd.weightbank on the host: emulate `ebfi_pack_table_bf16` (packed[e] = bf16 hi / lo of flat[table[e]], 0 for
    structural zeros) and compare every site's four images with the layouts the conv kernels expect, computed directly from
    the (folded / concatenated) fp32 weight: forward [tap][co][ci16], data gradient [tap][ci][co16] with flipped taps;
    folded biases likewise; and the adjoint tables route a gradient of the folded weight back exactly like autograd."""
    from ebfi_amd import fold3d, weightbank
    torch.manual_seed(21)
    net = EVFIAutoEx(**dict(DEFAULT_MODEL_ARGS, FrameBasech=8, EventBasech=8, InterCH=8, TB=4, step=2, channels=[4, 4, 8, 8]))
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn_like(p))
    params = list(net.parameters())
    bank = weightbank.build_for(net, params=params)
    rc = net.ResidualControl
    cat_site = bank.register([rc.Conv3[0][0].conv2d.weight, rc.Conv4[0][0].conv2d.weight],
                             [rc.Conv3[0][0].conv2d.bias, rc.Conv4[0][0].conv2d.bias], kind="cat34")
    bank._finalize()
    flat = bank.flat.double()
    table = bank.table.long()
    idx = torch.where(table >= 0, table & (weightbank.LO_FLAG - 1), torch.zeros_like(table))
    src = torch.where(table >= 0, flat[idx], torch.zeros((), dtype=torch.float64)).float()
    hi = src.bfloat16()
    packed = torch.where((table >= 0) & ((table & weightbank.LO_FLAG) != 0), (src - hi.float()).bfloat16(), hi)
    bias_buf = torch.where(bank.bias_table >= 0, bank.flat[bank.bias_table.long().clamp_min(0)], torch.zeros(()))

    def expect(W2, groups=1):
        M, K, ks, _ = W2.shape
        Mg = M // groups
        K16, M16 = (K + 15) // 16 * 16, (Mg + 15) // 16 * 16
        f = torch.zeros(ks * ks, M, K16)
        f[:, :, :K] = W2.permute(2, 3, 0, 1).reshape(ks * ks, M, K)
        t = torch.zeros(ks * ks, groups * K, M16)      # rows (group, ci), columns = output channels inside the group
        for gi in range(groups):
            t[:, gi * K:(gi + 1) * K, :Mg] = W2[gi * Mg:(gi + 1) * Mg].flip((2, 3)).permute(2, 3, 1, 0).reshape(ks * ks, K, Mg)
        split = lambda v: (v.bfloat16(), (v - v.bfloat16().float()).bfloat16())
        return split(f.reshape(-1)), split(t.reshape(-1))

    kinds = set()
    by_ptr = {p.data_ptr(): p for p in params}
    pair = {}                                    # first weight of a ResidualControl pair -> (Conv3 layer, Conv4 layer)
    for i in range(rc.step):
        for j in (0, 1):
            pair[rc.Conv3[i][j].conv2d.weight.data_ptr()] = (rc.Conv3[i][j].conv2d, rc.Conv4[i][j].conv2d)
    for (ptr, kind), s in bank.sites.items():
        w = by_ptr[ptr].detach()
        groups = 1
        if kind in ("rcA", "rcB"):               # declared by ResidualControl._ebfi_bank_register
            W2, b2 = torch.cat([pair[ptr][0].weight, pair[ptr][1].weight]).detach(), None
            groups = 2 if kind == "rcB" else 1
            assert s.groups == groups
            assert torch.equal(bias_buf[s.bias_off:s.bias_off + s.M], torch.cat([pair[ptr][0].bias, pair[ptr][1].bias]).detach())
        elif kind == "conv3d":
            W2, b2 = fold3d.fold_conv3d_weight(w), fold3d._rep2
        elif kind == "convT3d":
            W2, b2 = fold3d.fold_conv_transpose3d_weight(w), fold3d._rep8
        elif kind == "fuse_d2":                  # UNet3d_18._ebfi_bank_register: feature_fuse on y's own channel order
            W2, b2 = net.Detail._fuse_weight_on_depth_minor_channels(w), None
        elif kind == "cat34":
            W2, b2 = torch.cat([rc.Conv3[0][0].conv2d.weight, rc.Conv4[0][0].conv2d.weight]).detach(), None
        else:
            W2, b2 = w, (lambda t: t)
        assert tuple(W2.shape) == (s.M, s.K, s.ks, s.ks)
        (fh, fl), (th, tl) = expect(W2, groups)
        n = fh.numel()
        o = s.fwd_off // 2
        assert torch.equal(packed[o:o + n], fh) and torch.equal(packed[o + n:o + 2 * n], fl), kind
        assert s.fwd_bytes == 4 * n
        n, o = th.numel(), s.tr_off // 2
        assert torch.equal(packed[o:o + n], th) and torch.equal(packed[o + n:o + 2 * n], tl), kind
        kinds.add(kind)
        # adjoint of the fold: gradient of W2 routed back to the parameter
        if kind in ("conv3d", "convT3d", "fuse_d2"):
            wr = w.clone().requires_grad_()
            fold = {"conv3d": fold3d.fold_conv3d_weight, "convT3d": fold3d.fold_conv_transpose3d_weight,
                    "fuse_d2": net.Detail._fuse_weight_on_depth_minor_channels}[kind]
            g2 = torch.randn(s.M, s.K, s.ks, s.ks)
            fold(wr).backward(g2)
            inv = s.w_inv[0].long()
            routed = torch.cat([g2.flatten(), torch.zeros(1)])[inv].sum(-1).view(w.shape)
            assert torch.allclose(routed, wr.grad, atol=1e-6), kind
    assert kinds == {"id", "conv3d", "convT3d", "cat34", "rcA", "rcB", "fuse_d2"}
    # folded biases: one Conv_3d of the decoder (bias=True) and the concatenated pair
    dec = net.Detail.decoder[0].conv[0]
    sb = bank.sites[(dec.weight.data_ptr(), "conv3d")]
    assert sb.has_bias and torch.equal(bias_buf[sb.bias_off:sb.bias_off + sb.M], dec.bias.detach().repeat_interleave(2))
    assert torch.equal(bias_buf[cat_site.bias_off:cat_site.bias_off + cat_site.M],
                       torch.cat([rc.Conv3[0][0].conv2d.bias, rc.Conv4[0][0].conv2d.bias]).detach())


def test_fused_residual_control_declines_norm_variants():
    """ebfi_amd.rc_fused: the fused node computes conv + bias + LeakyReLU only.  The reference's `norm` option ('BN' drops
    the conv bias, 'IN' inserts a norm layer, submodules.py:159-200) must leave the module on its layer-by-layer path and
    must not break the weight-bank construction (round-2 advisory: norm='BN' raised AttributeError in Engine.__init__,
    norm='IN' silently skipped the InstanceNorm layers)."""
    from ebfi_amd import rc_fused, weightbank
    small = dict(DEFAULT_MODEL_ARGS, FrameBasech=8, EventBasech=8, InterCH=8, TB=4, step=2, channels=[4, 4, 8, 8])
    for norm, fus in ((None, True), ("BN", False), ("IN", False)):
        net = EVFIAutoEx(**dict(small, norm=norm))
        assert rc_fused.fusable(net.ResidualControl) is fus, norm
        bank = weightbank.build_for(net)                      # host-side registration only: must not raise
        kinds = {k for _, k in bank.sites}
        assert ("rcA" in kinds) is fus and ("rcB" in kinds) is fus, (norm, kinds)
        with bank.active():
            assert (rc_fused.sites_of(net.ResidualControl) is not None) is fus
            x = torch.zeros(1, 8, 8, 8)
            assert not rc_fused.usable(net.ResidualControl, x)          # CPU tensor / fp32 mode: never the fused node
    # another activation on the scalar-conditioned 1x1 layers also declines
    net = EVFIAutoEx(**small)
    net.ResidualControl.Conv1[0][0].activation = torch.nn.ReLU()
    assert not rc_fused.fusable(net.ResidualControl)
