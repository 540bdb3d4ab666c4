"""ORACLE (test infrastructure only): functional CPU restatement of ``EVFIAutoEx.forward``.

Everything is a pure function of a ``state_dict`` (reference parameter names) and the inputs;
no nn.Module, no product code.  Citations are to /root/reference/models:

  ConvLayer            model_misc/submodules.py:159-200  (conv2d -> activation; norm unused)
  ExposureDecision     Ours/model_singleframe.py:56-76
  ResidualControl      Ours/model_singleframe.py:115-136
  Modification         Ours/model_singleframe.py:152-165  (KPN = FAC K=5, replicate pad)
  Reconstruction       Ours/model_singleframe.py:257-266
  UNet3d_18            Ours/model_singleframe.py:200-223 + model_misc/resnet_3D.py:89-141,190-292,382-417
  EVFIAutoEx.forward   Ours/model_singleframe.py:277-348
  CropSize             model_misc/model_util.py:158-189

Pinned against outputs of the reference modules themselves (tests/golden/model_*.npz, produced by
tests/golden/make_golden.py; the FAC op inside Modification is the one piece the reference cannot
run on CPU and is supplied by oracle/ref_ops.py there too).
"""
from math import ceil, floor

import torch
import torch.nn.functional as F

from . import blur_ref, ref_ops


def _act(x, name):
    if name is None:
        return x
    if name == "LeakyReLU":
        return F.leaky_relu(x, 0.01)
    if name == "Sigmoid":
        return torch.sigmoid(x)
    if name == "ReLU":
        return F.relu(x)
    raise ValueError(name)


def conv_layer(sd, prefix, x, stride=1, padding=0, act="LeakyReLU"):
    y = F.conv2d(x, sd[prefix + ".conv2d.weight"], sd.get(prefix + ".conv2d.bias"), stride, padding)
    return _act(y, act)


def exposure_decision(sd, p, event, blurry_level, groups=4, act="LeakyReLU"):
    ef = conv_layer(sd, p + ".EventFeatExtract", event, 1, 1, act)
    bf = conv_layer(sd, p + ".BLFeatExtract", blurry_level, 1, 1, act)
    gw, gb = sd[p + ".GroupNorm.weight"], sd[p + ".GroupNorm.bias"]
    corr = F.group_norm(ef, groups, gw, gb) * F.group_norm(bf, groups, gw, gb)
    att = torch.sigmoid(corr.mean(dim=(2, 3), keepdim=True))
    y = conv_layer(sd, p + ".Conv1.0", torch.cat([ef * att, bf], 1), 1, 1, act)
    y = conv_layer(sd, p + ".Conv1.1", y, 1, 1, None)
    return torch.sigmoid(y.mean(dim=(2, 3)).view(-1, 1))


def residual_control(sd, p, data, ex, t, step, act="LeakyReLU"):
    ex = ex[:, :, None, None]
    t = t[:, :, None, None]
    x = data
    for i in range(step):
        ex_scale = conv_layer(sd, f"{p}.Conv1.{i}.0", ex, 1, 0, act)
        t_scale = conv_layer(sd, f"{p}.Conv2.{i}.0", t, 1, 0, act)
        a = conv_layer(sd, f"{p}.Conv3.{i}.1", conv_layer(sd, f"{p}.Conv3.{i}.0", x, 1, 1, act), 1, 1, act)
        b = conv_layer(sd, f"{p}.Conv4.{i}.1", conv_layer(sd, f"{p}.Conv4.{i}.0", x, 1, 1, act), 1, 1, act)
        x = conv_layer(sd, f"{p}.Conv5.{i}.0", torch.cat([ex_scale * a + x, t_scale * b + x], 1), 1, 1, act)
    return x


def modification(sd, p, frame_feat, event_feat, act="LeakyReLU", ksize=5, fac=None):
    fac = fac or ref_ops.fac_module
    e = conv_layer(sd, p + ".Conv1", event_feat, 1, 0, act)
    kernel = conv_layer(sd, p + ".KernelConv", torch.cat([e, frame_feat], 1), 1, 1, act)
    e1 = conv_layer(sd, p + ".Conv3", fac(e, kernel, ksize), 1, 1, act)
    return frame_feat * e1 + conv_layer(sd, p + ".Conv2", e1, 1, 1, act)


def reconstruction(sd, p, x, act="LeakyReLU"):
    y = conv_layer(sd, p + ".0.0", x, 1, 1, None)
    y = F.leaky_relu(F.pixel_shuffle(y, 2), 0.01)
    y = conv_layer(sd, p + ".1", y, 1, 1, act)
    return conv_layer(sd, p + ".2", y, 1, 1, "Sigmoid")


def _se_gate(sd, p, x):
    y = x.mean(dim=(2, 3, 4), keepdim=True)
    y = torch.sigmoid(F.conv3d(y, sd[p + ".attn_layer.0.weight"], sd[p + ".attn_layer.0.bias"]))
    return x * y


def _basic_block(sd, p, x, stride):
    out = F.relu(F.conv3d(x, sd[p + ".conv1.0.weight"], None, stride, 1))
    out = F.conv3d(out, sd[p + ".conv2.0.weight"], None, 1, 1)
    out = _se_gate(sd, p + ".fg", out)
    res = x
    if (p + ".downsample.0.weight") in sd:
        res = F.conv3d(x, sd[p + ".downsample.0.weight"], None, stride, 0)
    return F.relu(out + res)


def r3d18_encoder(sd, p, x):
    x0 = F.relu(F.conv3d(x, sd[p + ".stem.0.weight"], None, (1, 2, 2), (1, 3, 3)))
    feats = [x0]
    cur = x0
    for li, stride in ((1, 1), (2, (1, 2, 2)), (3, (1, 2, 2)), (4, 1)):
        cur = _basic_block(sd, f"{p}.layer{li}.0", cur, stride)
        cur = _basic_block(sd, f"{p}.layer{li}.1", cur, 1)
        feats.append(cur)
    return feats


def _conv3d_gated(sd, p, x):
    y = F.conv3d(x, sd[p + ".conv.0.weight"], sd[p + ".conv.0.bias"], 1, 1)
    return _se_gate(sd, p + ".conv.1", y)


def _upconv3d_gated(sd, p, x):
    y = F.conv_transpose3d(x, sd[p + ".upconv.0.weight"], sd[p + ".upconv.0.bias"], (1, 2, 2), (1, 1, 1))
    return _se_gate(sd, p + ".upconv.1", y)


def unet3d_18(sd, p, img0, img1):
    lrelu = lambda v: F.leaky_relu(v, 0.2)
    x0, x1, x2, x3, x4 = r3d18_encoder(sd, p + ".encoder", torch.stack((img0, img1), dim=2))
    d3 = torch.cat([lrelu(_conv3d_gated(sd, p + ".decoder.0", x4)), x3], 1)
    d2 = torch.cat([lrelu(_upconv3d_gated(sd, p + ".decoder.1", d3)), x2], 1)
    d1 = torch.cat([lrelu(_upconv3d_gated(sd, p + ".decoder.2", d2)), x1], 1)
    d0 = torch.cat([lrelu(_conv3d_gated(sd, p + ".decoder.3", d1)), x0], 1)
    do = lrelu(_upconv3d_gated(sd, p + ".decoder.4", d0))
    do = torch.cat(torch.unbind(do, 2), 1)
    out = lrelu(F.conv2d(do, sd[p + ".feature_fuse.0.weight"]))
    out = F.pad(out, (3, 3, 3, 3), mode="reflect")
    return F.conv2d(out, sd[p + ".outconv.1.weight"], sd[p + ".outconv.1.bias"])


def pad_to_multiple(x, H, W, m=8):
    """CropSize.pad (model_util.py:167-176): zero pad, larger half on top/left."""
    Hc, Wc = ceil(H / m) * m, ceil(W / m) * m
    top, bottom = ceil(0.5 * (Hc - H)), floor(0.5 * (Hc - H))
    left, right = ceil(0.5 * (Wc - W)), floor(0.5 * (Wc - W))
    return F.pad(x, (left, right, top, bottom))


def crop_from_multiple(x, H, W, m=8):
    """CropSize.crop (model_util.py:178-187)."""
    Hc, Wc = ceil(H / m) * m, ceil(W / m) * m
    cx, cy = floor(Wc / 2), floor(Hc / 2)
    return x[..., cy - floor(H / 2): cy + ceil(H / 2), cx - floor(W / 2): cx + ceil(W / 2)]


def blurry_level(frame, fashion):
    """model_singleframe.py:311-323; Lap/DCP via the (unpinned) OpenCV restatement."""
    lap = lambda: torch.from_numpy(blur_ref.frame2lap(frame.detach().numpy()))
    dcp = lambda: torch.from_numpy(blur_ref.frame2dcp(frame.detach().numpy()))
    if fashion == "DarkCh":
        return dcp()
    if fashion == "Lap":
        return lap()
    if fashion == "RGB":
        return frame
    if fashion == "RGBDark":
        return torch.cat([frame, dcp()], 1)
    if fashion == "RGBLap":
        return torch.cat([frame, lap()], 1)
    raise Exception("Wrong blurry convertion fashion!!")


def evfi_forward(sd, cfg, frame, event, t, gt_ex=None, fac=None, blurry=None):
    """cfg: the reference's model args (config/train_ours.yml:28-57).  Returns (Sharp, Final).
    ``blurry`` optionally overrides the BlurryLevel map (lets pinned tests bypass OpenCV)."""
    act = cfg.get("activation", "LeakyReLU")
    H, W = frame.shape[-2:]
    need_crop = (H % 8 != 0) or (W % 8 != 0)
    if need_crop:
        frame = pad_to_multiple(frame, H, W)
        event = pad_to_multiple(event, H, W)
    event = event.reshape(event.size(0), -1, event.size(3), event.size(4))
    ff = conv_layer(sd, "FrameFeatExtract", frame, 2, 1, act)
    ef = conv_layer(sd, "EventFeatExtract", event, 2, 1, act)
    if cfg.get("UseGTEx", False):
        assert cfg.get("FixEx") is None and gt_ex is not None
        ex = gt_ex
    elif cfg.get("FixEx"):
        ex = torch.full((frame.size(0), 1), float(cfg["FixEx"]), dtype=frame.dtype)
    else:
        bl = blurry if blurry is not None else blurry_level(frame, cfg.get("BlurryFashion", "DarkCh"))
        ex = exposure_decision(sd, "ExposureDecision", event, bl, 4, act)
    pe = residual_control(sd, "ResidualControl", ef, ex, t, cfg["step"], act)
    pf = modification(sd, "Modification", ff, pe, act, 5, fac)
    sharp = reconstruction(sd, "Reconstruction", pf, act)
    detail_on = cfg.get("DetailEnabled", True)
    final = sharp + unet3d_18(sd, "Detail", frame, sharp) if detail_on else sharp
    if need_crop:
        sharp = crop_from_multiple(sharp, H, W).contiguous()
        final = crop_from_multiple(final, H, W).contiguous() if detail_on else sharp
    return sharp, final
