#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by RUNNING THE REFERENCE'S OWN PYTHON
(/root/reference, present only in the build container; never on the GPU box).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

What is imported from the reference and how:
  * dataloader/encodings.py  -- imported as a plain file (needs only numpy/torch).
  * models/Ours/model_singleframe.py, loss/restore.py -- imported after registering EMPTY
    placeholder modules for third-party imports that are absent from this image and that the
    executed code never touches (torchvision, cv2, open3d, h5py, skimage, the two CUDA extension
    modules `_ext` / `kernelconv2d_cuda`, LPIPS).  No behaviour is supplied by a placeholder.
  * Two call sites cannot execute on CPU in the reference and are substituted, as recorded in
    DESIGN.md: `Modification.KPN` (FAC is CUDA-only: KernelConv2D.py:38-39) is replaced by the
    oracle's FAC module, and full-model fixtures use `UseGTEx=True` / an explicit BlurryLevel so
    that OpenCV (`Frame2Lap`, myutils/utils.py:34-49) is not needed.  `Ternary.w` is turned into
    a tensor by hand because the reference only does so when CUDA is available (restore.py:118).

Only data (inputs, weights, expected outputs) is written; no reference source text is stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    for n in ["torchvision", "torchvision.models", "torchvision.transforms", "cv2", "open3d", "h5py",
              "_ext", "kernelconv2d_cuda", "skimage"]:
        _placeholder(n)
    _placeholder("torchvision.models.resnet", resnet34=None)
    _placeholder("torchvision.models.utils", load_state_dict_from_url=None)
    _placeholder("skimage.metrics", structural_similarity=None, peak_signal_noise_ratio=None)
    import matplotlib.style as mstyle
    orig = mstyle.use

    def tolerant_use(style):          # 'seaborn-whitegrid' was removed from matplotlib 3.8+
        try:
            orig(style)
        except Exception:
            pass
    mstyle.use = tolerant_use
    import matplotlib.pyplot as plt
    plt.style.use = tolerant_use
    sys.path.insert(0, REF)
    import models.Ours.model_singleframe as ms
    # loss/__init__ star-imports LPIPS etc.; load restore.py alone with its relative import stubbed
    _placeholder("refloss")
    sys.modules["refloss"].__path__ = []
    _placeholder("refloss.PerceptualSimilarity", models=None)
    spec = importlib.util.spec_from_file_location("refloss.restore", os.path.join(REF, "loss/restore.py"))
    restore = importlib.util.module_from_spec(spec)
    sys.modules["refloss.restore"] = restore
    spec.loader.exec_module(restore)
    spec = importlib.util.spec_from_file_location("ref_encodings", os.path.join(REF, "dataloader/encodings.py"))
    enc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(enc)
    return ms, restore, enc


def np32(t):
    return t.detach().cpu().numpy()


# --------------------------------------------------------------------------------- events
def event_cases():
    rng = np.random.default_rng(7)
    cases = {}

    def mk(n, H, W, oob=0.0, frac=False, dup=False):
        ts = np.sort(rng.random(n))
        if dup:
            ts = np.round(ts * 40) / 40          # many repeated timestamps
        ts = (ts - ts[0]) / (ts[-1] - ts[0] + 1e-6)        # h5dataset.py:334 normalisation
        xs = rng.integers(0, W, n).astype(np.float64)
        ys = rng.integers(0, H, n).astype(np.float64)
        if frac:
            xs += rng.random(n) * 0.9
            ys += rng.random(n) * 0.9
        if oob > 0:
            bad = rng.random(n) < oob
            xs[bad] += rng.choice([-W - 3, W, 2 * W], bad.sum())
            bad = rng.random(n) < oob
            ys[bad] += rng.choice([-H, H + 1], bad.sum())
        ps = rng.choice([-1.0, 1.0], n).astype(np.float32)
        return xs, ys, ts, ps

    cases["random"] = (*mk(3000, 24, 32), 16, (24, 32))
    cases["dup_ts"] = (*mk(2000, 16, 20, dup=True), 16, (16, 20))
    cases["oob"] = (*mk(2500, 20, 28, oob=0.2, dup=True), 16, (20, 28))
    cases["frac_xy"] = (*mk(1500, 12, 18, frac=True), 8, (12, 18))
    # events exactly on the bin edges the reference computes (ts[0] + delta_t * k)
    xs, ys, ts, ps = mk(800, 10, 14, oob=0.15)
    dt = ts[-1] - ts[0] + 1e-6
    edges = np.array([ts[0] + (dt / 16) * k for k in range(1, 16)] +
                     [ts[0] + (dt / 16) * k + (dt / 16) for k in range(0, 15)])
    idx = rng.choice(np.arange(1, len(ts) - 1), len(edges), replace=False)
    ts[idx] = edges
    order = np.argsort(ts, kind="stable")
    cases["on_edges"] = (xs[order], ys[order], ts[order], ps[order], 16, (10, 14))
    cases["three_events"] = (np.array([1., 2., 3.]), np.array([1., 1., 2.]), np.array([0., .5, 1.]),
                             np.array([1., -1., 1.], dtype=np.float32), 4, (6, 6))
    cases["four_events"] = (np.array([1., 2., 3., 7.]), np.array([1., 1., 2., 0.]),
                            np.array([0., .25, .5, 1.]), np.array([1., -1., 1., -1.], dtype=np.float32), 4, (6, 6))
    cases["zero_ts"] = (np.arange(6.), np.arange(6.), np.zeros(6), np.ones(6, dtype=np.float32), 4, (8, 8))
    cases["pol01"] = (*mk(500, 8, 8)[:3], rng.choice([0.0, 1.0], 500).astype(np.float32), 4, (8, 8))
    return cases


def make_events(enc):
    out = {}
    for name, (xs, ys, ts, ps, B, size) in event_cases().items():
        ref = enc.events_to_stack(torch.from_numpy(xs.copy()), torch.from_numpy(ys.copy()),
                                  torch.from_numpy(ts.copy()), torch.from_numpy(ps.copy()).float(),
                                  B=B, sensor_size=size)
        for k, v in dict(xs=xs, ys=ys, ts=ts, ps=ps, B=np.int64(B), size=np.array(size), out=np32(ref)).items():
            out[f"{name}.{k}"] = v
    np.savez_compressed(os.path.join(HERE, "events_to_stack.npz"), **out)
    print("events_to_stack.npz:", len(event_cases()), "cases")


# --------------------------------------------------------------------------------- model
SMALL_CFG = dict(FrameBasech=8, EventBasech=8, InterCH=8, TB=4, norm=None, activation="LeakyReLU",
                 BlurryFashion="RGBLap", BLInch=4, UseEvents=True, UseGTEx=False, FixEx=None,
                 LoadPretrainEX=False, PretrainedEXPath=None, FrozenEX=False, step=2, DualPath=True,
                 residual=True, DetailEnabled=True, channels=[4, 4, 8, 8])


class OracleKPN(torch.nn.Module):
    """Stand-in for Modification.KPN (reference FAC is CUDA-only)."""

    def forward(self, x, kernel):
        from oracle import ref_ops
        return ref_ops.fac_module(x, kernel, 5)


def rerandomise(net, gen):
    """Default init x0.1 makes Sharp ~ 0.5 everywhere; use O(1) activations instead."""
    with torch.no_grad():
        for name, p in net.named_parameters():
            if p.dim() > 1:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=gen) * (1.2 / fan_in ** 0.5))
            elif "GroupNorm.weight" in name:
                p.copy_(1 + 0.2 * torch.randn(p.shape, generator=gen))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=gen))


def make_model(ms):
    gen = torch.Generator().manual_seed(1234)
    net = ms.EVFIAutoEx(**SMALL_CFG)
    net.Modification.KPN = OracleKPN()
    rerandomise(net, gen)
    net.eval()
    B, TB, H, W = 2, SMALL_CFG["TB"], 32, 40
    frame = torch.rand(B, 3, H, W, generator=gen)
    event = torch.poisson(torch.full((B, TB, 2, H, W), 0.35), generator=gen)
    t = torch.rand(B, 1, generator=gen)
    gtex = torch.rand(B, 1, generator=gen) * 0.4 + 0.55
    blurry = torch.cat([frame, torch.round(torch.randn(B, 1, H, W, generator=gen) * 40)], 1)

    out = {"cfg": np.array(repr(SMALL_CFG))}
    for k, v in net.state_dict().items():
        out["sd." + k] = np32(v)
    out.update({"in.Frame": np32(frame), "in.Event": np32(event), "in.T": np32(t), "in.GTEx": np32(gtex),
                "in.Blurry": np32(blurry)})

    ev = event.view(B, -1, H, W)
    ff = net.FrameFeatExtract(frame)
    ef = net.EventFeatExtract(ev)
    ex = net.ExposureDecision(ev, blurry)
    pe = net.ResidualControl(ef, ex, t)
    pf = net.Modification(ff, pe)
    sharp = net.Reconstruction(pf)
    detail = net.Detail(img0=frame, img1=sharp)
    out.update({"mid.FrameFeat": np32(ff), "mid.EventFeat": np32(ef), "mid.Ex": np32(ex),
                "mid.ResidualControl": np32(pe), "mid.Modification": np32(pf),
                "out.Sharp": np32(sharp), "mid.Detail": np32(detail), "out.Final": np32(sharp + detail)})

    # full forward through the reference's own forward() with UseGTEx (no OpenCV on that branch)
    net.UseGTEx = True
    s2, f2 = net(frame, event, t, gtex)
    out.update({"gtex.Sharp": np32(s2), "gtex.Final": np32(f2)})
    # gradients of a fixed scalar of both outputs w.r.t. every parameter (UseGTEx branch)
    net.zero_grad()
    wS = torch.randn(s2.shape, generator=gen)
    wF = torch.randn(f2.shape, generator=gen)
    ((s2 * wS).sum() + (f2 * wF).sum()).backward()
    out.update({"gtex.wS": np32(wS), "gtex.wF": np32(wF)})
    for k, p in net.named_parameters():
        if p.grad is not None:
            out["grad." + k] = np32(p.grad)

    # odd size: exercises the pad-to-multiple-of-8 / crop path (model_singleframe.py:289-295,338-343)
    Ho, Wo = 27, 37
    frame_o = torch.rand(1, 3, Ho, Wo, generator=gen)
    event_o = torch.poisson(torch.full((1, TB, 2, Ho, Wo), 0.35), generator=gen)
    so, fo = net(frame_o, event_o, t[:1], gtex[:1])
    out.update({"odd.Frame": np32(frame_o), "odd.Event": np32(event_o), "odd.Sharp": np32(so), "odd.Final": np32(fo)})
    np.savez_compressed(os.path.join(HERE, "model_small.npz"), **out)
    print("model_small.npz:", len(out), "arrays,", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")

    # parameter names + shapes of the DEFAULT config (config/train_ours.yml:28-57): the drop-in contract
    full_cfg = dict(SMALL_CFG, FrameBasech=64, EventBasech=64, InterCH=64, TB=16, step=12, channels=[16, 24, 32, 64])
    full = ms.EVFIAutoEx(**full_cfg)
    with open(os.path.join(HERE, "state_dict_default.txt"), "w") as f:
        for k, v in full.state_dict().items():
            f.write("%s %s\n" % (k, "x".join(map(str, v.shape))))
    print("state_dict_default.txt:", len(full.state_dict()), "entries,",
          sum(p.numel() for p in full.parameters()), "params")


# --------------------------------------------------------------------------------- loss
def make_loss(restore):
    gen = torch.Generator().manual_seed(99)
    lap = restore.LaplacianLoss()
    census = restore.Ternary()
    if not torch.is_tensor(census.w):
        census.w = torch.tensor(census.w).float()
    x = torch.rand(2, 3, 64, 48, generator=gen).requires_grad_()
    y = torch.rand(2, 3, 64, 48, generator=gen)
    l1, l2 = lap(x, y), census(x, y)
    (l1 + l2).backward()
    np.savez_compressed(os.path.join(HERE, "loss_small.npz"), x=np32(x), y=np32(y), lap=np32(l1),
                        census=np32(l2), grad_x=np32(x.grad))
    print("loss_small.npz: lap=%.6f census=%.6f" % (l1.item(), l2.item()))


if __name__ == "__main__":
    torch.manual_seed(0)
    ms, restore, enc = import_reference()
    make_events(enc)
    make_model(ms)
    make_loss(restore)
