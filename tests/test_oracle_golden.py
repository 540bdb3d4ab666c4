"""Oracle (oracle/*.py) against fixtures produced by running the reference's own Python
(tests/golden/make_golden.py).  CPU only."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import events_ref, loss_ref, model_ref


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _event_case_names(golden_dir):
    z = _load(golden_dir, "events_to_stack.npz")
    return sorted({k.split(".")[0] for k in z.files})


def test_events_to_stack_bit_exact(golden_dir):
    z = _load(golden_dir, "events_to_stack.npz")
    names = sorted({k.split(".")[0] for k in z.files})
    assert len(names) >= 9
    for n in names:
        out = events_ref.events_to_stack(z[n + ".xs"], z[n + ".ys"], z[n + ".ts"], z[n + ".ps"],
                                         int(z[n + ".B"]), tuple(z[n + ".size"]))
        assert out.dtype == np.float32
        assert np.array_equal(out, z[n + ".out"]), n
    # the fixtures really contain the quirks they are meant to pin
    assert z["oob.out"][:, :, 0, 0].sum() > 0
    assert z["three_events.out"].sum() == 0 and z["zero_ts.out"].sum() == 0
    assert z["on_edges.out"].sum() > (np.abs(z["on_edges.ps"]) > 0).sum() * 0.8


@pytest.fixture(scope="module")
def model_fix(golden_dir):
    z = _load(golden_dir, "model_small.npz")
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    cfg = ast.literal_eval(str(z["cfg"]))
    t = lambda k: torch.from_numpy(z[k])
    return z, sd, cfg, t


def _close(a, b, tol=2e-5):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    return (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


def test_model_submodules(model_fix):
    z, sd, cfg, t = model_fix
    frame, event, T, blurry = t("in.Frame"), t("in.Event"), t("in.T"), t("in.Blurry")
    ev = event.view(event.size(0), -1, event.size(3), event.size(4))
    ff = model_ref.conv_layer(sd, "FrameFeatExtract", frame, 2, 1)
    ef = model_ref.conv_layer(sd, "EventFeatExtract", ev, 2, 1)
    assert _close(ff, z["mid.FrameFeat"]) and _close(ef, z["mid.EventFeat"])
    ex = model_ref.exposure_decision(sd, "ExposureDecision", ev, blurry)
    assert _close(ex, z["mid.Ex"])
    pe = model_ref.residual_control(sd, "ResidualControl", t("mid.EventFeat"), t("mid.Ex"), T, cfg["step"])
    assert _close(pe, z["mid.ResidualControl"])
    pf = model_ref.modification(sd, "Modification", t("mid.FrameFeat"), t("mid.ResidualControl"))
    assert _close(pf, z["mid.Modification"])
    sharp = model_ref.reconstruction(sd, "Reconstruction", t("mid.Modification"))
    assert _close(sharp, z["out.Sharp"])
    det = model_ref.unet3d_18(sd, "Detail", frame, t("out.Sharp"))
    assert _close(det, z["mid.Detail"])


def test_model_end_to_end(model_fix):
    z, sd, cfg, t = model_fix
    sharp, final = model_ref.evfi_forward(sd, cfg, t("in.Frame"), t("in.Event"), t("in.T"),
                                          blurry=t("in.Blurry"))
    assert _close(sharp, z["out.Sharp"]) and _close(final, z["out.Final"])
    cfg2 = dict(cfg, UseGTEx=True)
    sharp, final = model_ref.evfi_forward(sd, cfg2, t("in.Frame"), t("in.Event"), t("in.T"), t("in.GTEx"))
    assert _close(sharp, z["gtex.Sharp"]) and _close(final, z["gtex.Final"])
    # outputs are not degenerate
    assert z["gtex.Sharp"].std() > 0.05


def test_model_pad_crop(model_fix):
    z, sd, cfg, t = model_fix
    cfg2 = dict(cfg, UseGTEx=True)
    sharp, final = model_ref.evfi_forward(sd, cfg2, t("odd.Frame"), t("odd.Event"), t("in.T")[:1], t("in.GTEx")[:1])
    assert sharp.shape[-2:] == (27, 37)
    assert _close(sharp, z["odd.Sharp"]) and _close(final, z["odd.Final"])


def test_model_gradients(model_fix):
    z, sd, cfg, t = model_fix
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    cfg2 = dict(cfg, UseGTEx=True)
    sharp, final = model_ref.evfi_forward(sd, cfg2, t("in.Frame"), t("in.Event"), t("in.T"), t("in.GTEx"))
    ((sharp * t("gtex.wS")).sum() + (final * t("gtex.wF")).sum()).backward()
    checked = 0
    for k in z.files:
        if not k.startswith("grad."):
            continue
        g = sd[k[5:]].grad
        assert g is not None, k
        assert _close(g, z[k], 1e-4), k
        checked += 1
    assert checked > 80


def test_loss(golden_dir):
    z = _load(golden_dir, "loss_small.npz")
    x = torch.from_numpy(z["x"]).requires_grad_()
    y = torch.from_numpy(z["y"])
    lap, cen = loss_ref.laplacian_loss(x, y), loss_ref.census_loss(x, y)
    assert abs(lap.item() - float(z["lap"])) <= 1e-5 * float(z["lap"])
    assert abs(cen.item() - float(z["census"])) <= 1e-5
    (lap + cen).backward()
    assert _close(x.grad, z["grad_x"], 1e-4)
