"""EVFIAutoEx -- the frame-synthesis network of EBFI-BE behind the reference's nn.Module API.

Drop-in contract (reference models/Ours/model_singleframe.py:226-348): same constructor keywords,
``forward(Frame[B,3,H,W], Event[B,TB,2,H,W], T[B,1], GTEx=None) -> (Sharp, Final)``,
``LoadExposureDecision()``, and -- so that reference checkpoints load -- the same ``state_dict``
key names and shapes (checked against tests/golden/state_dict_default.txt).

What differs from the reference, on purpose:
  * the FAC op inside ``Modification`` is the gfx950 kernel of libebfi_hip.so (ebfi_amd.fac);
  * ``Frame2Lap`` / ``Frame2DCP`` run as device kernels (ebfi_amd.blur) instead of a
    GPU -> host -> OpenCV -> GPU round trip inside forward (myutils/utils.py:15-49);
  * the ``FixEx`` branch builds its tensor on the input's device instead of a hard ``.cuda()``
    (model_singleframe.py:309);
  * sub-modules are assembled from small builders rather than one class per block.
2-D convolutions of ConvLayer run on the hand-written MFMA kernels (ebfi_amd.conv); the depth-2 3-D convs of the
detail branch are folded onto the same 2-D kernels (ebfi_amd.fold3d) -- no MIOpen kernel is on the path, see DESIGN.md.
"""
from math import ceil, floor

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native as N
from . import conv, fold3d, fused, norm
from .blur import Frame2DCP, Frame2Lap
from .fac import KernelConv2D


class BaseModel(nn.Module):
    """models/model_misc/base.py:11-33: __str__ appends the parameter counts."""

    def __str__(self):
        trainable = sum(p.numel() for p in self.parameters() if p.requires_grad)
        total = sum(p.numel() for p in self.parameters())
        return super().__str__() + "\nTrainable parameters: {} \nAll parameters: {}".format(trainable, total)


class ConvLayer(nn.Module):
    """Conv2d (+BN/IN) (+activation); parameter container named ``conv2d`` like
    models/model_misc/submodules.py:159-200."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation="ReLU", norm=None,
                 BN_momentum=0.1):
        super().__init__()
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=(norm != "BN"))
        self.activation = getattr(nn, activation)() if activation is not None else None
        self.norm = norm
        if norm == "BN":
            self.norm_layer = nn.BatchNorm2d(out_channels, momentum=BN_momentum)
        elif norm == "IN":
            self.norm_layer = nn.InstanceNorm2d(out_channels, track_running_stats=True)

    def native(self, x):
        """(act code, slope) when this layer runs as ONE fused native kernel on `x`, else None."""
        c = self.conv2d
        fuse = conv.activation_code(self.activation) if self.norm not in ("BN", "IN") else None
        if fuse is not None and not torch.is_autocast_enabled() and \
                conv.supported(x, c.weight, c.stride, c.padding, c.dilation, c.groups):
            return fuse
        return None

    def forward(self, x, grad_is_preact=False):
        """grad_is_preact (native path only): the consumer of the output hands back the gradient of the PRE-activation
        (see fac.KernelConv2DFunction kernel_leaky_slope), so backward needs no saved output."""
        c = self.conv2d
        fuse = self.native(x)
        if fuse is not None:
            return conv.conv_bias_act(x, c.weight, c.bias, c.stride[0], c.padding[0], fuse[0], fuse[1], grad_is_preact)
        if grad_is_preact:
            raise RuntimeError("grad_is_preact needs the fused native conv kernel")
        if x.shape[-2:] == (1, 1) and c.kernel_size == (1, 1) and c.groups == 1:
            # scalar-conditioned scales (ResidualControl's Conv1 / Conv2 on Ex / T): an outer product + bias
            v = x.flatten(1)
            leaky = isinstance(self.activation, nn.LeakyReLU) and self.norm not in ("BN", "IN")
            if leaky and fused.scalar_conv_usable(v, [c.weight]):
                return fused.scalar_conv_bank(v, [c.weight], [c.bias], float(self.activation.negative_slope))[0][:, :, None, None]
            y = F.linear(v, c.weight.flatten(1), c.bias)[:, :, None, None]
        else:
            conv.left_native(x, c.weight, c.stride, c.padding, c.dilation, c.groups)   # (one line on stderr, or an error if strict)
            y = c(x)   # shapes the gfx950 conv kernels do not cover: PyTorch-ROCm conv (still GPU)
        if self.norm in ("BN", "IN"):
            y = self.norm_layer(y)
        return y if self.activation is None else self.activation(y)


def initialize_weights(nets, scale=1):
    """models/model_misc/model_util.py:16-36."""
    for net in nets if isinstance(nets, (list, tuple)) else [nets]:
        for m in net.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight, a=0, mode="fan_in")
                m.weight.data *= scale
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias.data, 0.0)


class CropSize:
    """Zero-pad to a multiple of patch_size and crop back (model_util.py:158-189)."""

    def __init__(self, width, height, patch_size):
        self.width, self.height = width, height
        self.wc = int(patch_size["w"] * ceil(width / patch_size["w"]))
        self.hc = int(patch_size["h"] * ceil(height / patch_size["h"]))
        dh, dw = self.hc - height, self.wc - width
        self.pad = nn.ZeroPad2d((ceil(0.5 * dw), floor(0.5 * dw), ceil(0.5 * dh), floor(0.5 * dh)))

    def crop(self, img):
        cx, cy = floor(self.wc / 2), floor(self.hc / 2)
        return img[..., cy - floor(self.height / 2): cy + ceil(self.height / 2),
                   cx - floor(self.width / 2): cx + ceil(self.width / 2)]


def _conv(cin, cout, k, s, p, norm, act):
    return ConvLayer(in_channels=cin, out_channels=cout, kernel_size=k, stride=s, padding=p, norm=norm, activation=act)


class ExposureDecision(BaseModel):
    """Event / blur-level correlation -> exposure duty in [0,1]  (model_singleframe.py:23-76)."""

    def __init__(self, EventInch=32, BLInch=1, InterCH=64, Group=4, norm=None, activation="LeakyReLU",
                 LoadPretrain=False, PretrainedEXPath=None, Frozen=False):
        super().__init__()
        self.LoadPretrain, self.PretrainedEXPath, self.Frozen = LoadPretrain, PretrainedEXPath, Frozen
        self.EventFeatExtract = _conv(EventInch, InterCH, 3, 1, 1, norm, activation)
        self.BLFeatExtract = _conv(BLInch, InterCH, 3, 1, 1, norm, activation)
        self.GroupNorm = nn.GroupNorm(Group, InterCH)
        self.AVGPool = nn.AdaptiveAvgPool2d(1)
        self.Conv1 = nn.Sequential(_conv(2 * InterCH, InterCH, 3, 1, 1, norm, activation),
                                   _conv(InterCH, 1, 3, 1, 1, norm, None))
        initialize_weights([self.EventFeatExtract, self.BLFeatExtract, self.GroupNorm, self.Conv1], 0.1)
        self.load_pretrain()

    def load_pretrain(self):
        if self.LoadPretrain:
            cpt = torch.load(self.PretrainedEXPath, map_location="cpu")
            self.load_state_dict(cpt["model"]["states"])
        if self.Frozen:
            for p in self.parameters():
                p.requires_grad = False
            self.eval()

    def forward(self, Event, BlurryLevel):
        ev = self.EventFeatExtract(Event)
        bl = self.BLFeatExtract(BlurryLevel)
        cat = fused.ed_head(ev, bl, self.GroupNorm)    # GN x2 -> pooled product -> sigmoid -> cat([ev * atten, bl]) as one node
        if cat is None:
            # sigmoid(AVGPool(GN(ev) * GN(bl))): the pooled product as one reduction (the product map is never written)
            atten = torch.sigmoid(fused.product_mean(norm.group_norm(ev, self.GroupNorm), norm.group_norm(bl, self.GroupNorm)))
            cat = fused.scale_cat(ev, atten, bl)                # cat([ev * atten, bl], 1) as one fused stage
        ex = self.Conv1(cat)
        return torch.sigmoid(self.AVGPool(ex).view(-1, 1))


class ResidualControl(BaseModel):
    """`step` rounds of exposure- and time-modulated residual refinement (model_singleframe.py:79-136)."""

    def __init__(self, BLinch=2, Tinch=1, Basech=16, step=4, norm=None, activation="LeakyReLU"):
        super().__init__()
        self.step = step

        def bank(make):
            return nn.ModuleList([nn.Sequential(*make()) for _ in range(step)])

        c3 = lambda cin: _conv(cin, Basech, 3, 1, 1, norm, activation)
        self.Conv1 = bank(lambda: [_conv(BLinch, Basech, 1, 1, 0, norm, activation)])
        self.Conv2 = bank(lambda: [_conv(Tinch, Basech, 1, 1, 0, norm, activation)])
        self.Conv3 = bank(lambda: [c3(Basech), c3(Basech)])
        self.Conv4 = bank(lambda: [c3(Basech), c3(Basech)])
        self.Conv5 = bank(lambda: [c3(2 * Basech)])
        initialize_weights([self.Conv1, self.Conv2, self.Conv3, self.Conv4, self.Conv5], 0.1)

    def _ebfi_bank_register(self, bank):
        from . import rc_fused
        rc_fused.register(bank, self)

    def forward(self, data, Ex, T):
        from . import rc_fused
        if rc_fused.usable(self, data):       # split-precision mode with an active weight bank: one hand-scheduled node
            out = rc_fused.residual_control(self, data, Ex, T)
            if out is not None:
                return out
        ex, t = Ex[:, :, None, None], T[:, :, None, None]
        x = data
        for i in range(self.step):
            # cat([Conv1(ex)*Conv3(x) + x, Conv2(t)*Conv4(x) + x], 1) as one fused stage
            x = self.Conv5[i](fused.scale_residual_cat(self.Conv3[i](x), self.Conv1[i](ex), self.Conv4[i](x), self.Conv2[i](t), x))
        return x


class Modification(BaseModel):
    """Event features -> per-pixel 5x5 filters applied to themselves (FAC), then gate the frame
    features (model_singleframe.py:139-165)."""

    def __init__(self, FrameBasech=64, EventBasech=32, TB=16, KernelSize=5, norm=None, activation="LeakyReLU"):
        super().__init__()
        self.Conv1 = _conv(EventBasech, FrameBasech, 1, 1, 0, norm, activation)
        self.Conv2 = _conv(FrameBasech, FrameBasech, 3, 1, 1, norm, activation)
        self.KernelConv = _conv(2 * FrameBasech, FrameBasech * KernelSize ** 2, 3, 1, 1, norm, activation)
        self.KPN = KernelConv2D(kernel_size=KernelSize)
        self.Conv3 = _conv(FrameBasech, FrameBasech, 3, 1, 1, norm, activation)
        initialize_weights([self.Conv1, self.Conv2, self.Conv3, self.KernelConv], 0.1)
        self.KernelConv.conv2d._ebfi_fwd16 = "filters"     # (weight bank: the layer Engine(forward_f16="filters") runs on fp16 operands)

    def _ebfi_bank_register(self, bank):
        """Inference banks also hold the KernelConv weight re-tiled to one FAC channel per 32-row matrix tile: the layout of
        the fused KernelConv -> FAC kernel (ebfi_amd.fac.kernelconv_fac_fused)."""
        from . import fac
        kc, k = self.KernelConv, self.KPN.kernel_size
        c = kc.conv2d
        if bank.inference and k == 5 and kc.norm is None and isinstance(kc.activation, nn.LeakyReLU) and c.bias is not None and \
                c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.out_channels % (k * k) == 0:
            # (with a scale book on the bank the site also gets an fp16 image: the fused kernel then runs one product per tap)
            bank.register(c.weight, c.bias, "facrows", fac.fac_rows_fold_weight, fac.fac_rows_fold_bias, need_tr=False,
                          fwd16=bank.book is not None)

    def _fused_filters_apply(self, ev, frame):
        """FAC(ev, LeakyReLU(KernelConv(cat([ev, frame], 1)))) as one kernel, or None when the fused form does not apply (training,
        other precision modes, no inference bank, rows that do not split into 16-byte quads).  The concatenation is left to the
        kernel's host side: on fp16 operands it exists only as the fp16 image."""
        from . import fac, weightbank
        if torch.is_grad_enabled() and (ev.requires_grad or frame.requires_grad or any(p.requires_grad for p in self.KernelConv.parameters())):
            return None
        if conv.get_compute_dtype() != "bf16x3" or not frame.is_cuda or frame.dtype != torch.float32 or ev.dtype != torch.float32 or \
                frame.shape[-1] % 4 != 0 or frame.shape[0] != ev.shape[0] or frame.shape[2:] != ev.shape[2:] or \
                N.dev_env("EBFI_NO_FAC_FUSION", "0") == "1":
            return None
        site = weightbank.lookup(self.KernelConv.conv2d.weight, "facrows")
        if site is None:
            return None
        return fac.kernelconv_fac_fused((ev, frame), ev, site, self.KPN.kernel_size, float(self.KernelConv.activation.negative_slope))

    def forward(self, FrameTensor, EventTensor):
        ev = self.Conv1(EventTensor)
        from . import f16scale, fac as facmod, weightbank
        book = f16scale.active_book()
        c = self.KernelConv.conv2d
        site = weightbank.lookup(c.weight, "id") if (book is not None and conv.get_compute_dtype() == "bf16x3") else None
        if torch.is_grad_enabled() and site is not None and FrameTensor.shape[0] == ev.shape[0] and \
                FrameTensor.shape[2:] == ev.shape[2:]:
            # training step with the fp16 backward: the filters and their gradient exist only as fp16 planes (ebfi_amd.fac), and
            # cat([ev, FrameTensor]) only as the fp16 image both passes of the 128 -> 1600 convolution read
            fuse = self.KernelConv.native(FrameTensor)          # (the same predicate on a part: device, dtype, layer form)
            if fuse is not None and fuse[0] == conv.ACT_LEAKY and \
                    facmod.kernelconv_fac_train_usable(site, book, FrameTensor, ev, self.KPN.kernel_size):
                ev1 = self.Conv3(facmod.KernelConvFacTrain.apply(FrameTensor, ev, site, fuse[1], self.KPN.kernel_size, c.weight, c.bias))
                return FrameTensor * ev1 + self.Conv2(ev1)
        fused = self._fused_filters_apply(ev, FrameTensor)
        if fused is not None:
            ev1 = self.Conv3(fused)
            return FrameTensor * ev1 + self.Conv2(ev1)
        cat = torch.cat([ev, FrameTensor], dim=1)
        fuse = self.KernelConv.native(cat)
        if fuse is not None and fuse[0] == conv.ACT_LEAKY and ev.is_cuda and N.dev_env("EBFI_NO_PREACT", "0") != "1":
            # the 1600-channel filter tensor has one consumer, the FAC op: its backward returns the gradient of the filters'
            # PRE-activation (kernel > 0 ? g : slope*g), so the 128 -> 1600 conv's weight / data gradient neither re-read the
            # 839 MB saved output for act' nor write / read a grad*act' side tensor of that size
            filters = self.KernelConv(cat, grad_is_preact=True)
            if site is not None and torch.is_grad_enabled():
                book.operand((site.key, "f"), filters)        # (calibration pass: the scale the fp16 filter planes will carry)
            ev1 = self.Conv3(self.KPN(ev, filters, kernel_leaky_slope=fuse[1]))
        else:
            ev1 = self.Conv3(self.KPN(ev, self.KernelConv(cat)))
        return FrameTensor * ev1 + self.Conv2(ev1)


# ---------------------------------------------------------------------------- detail branch (R3D-18 U-Net)
class identity(nn.Module):
    def __init__(self, *args):
        super().__init__()

    def forward(self, x):
        return x


class SEGating(nn.Module):
    """Channel gate from a global average (resnet_3D.py:89-105)."""

    def __init__(self, inplanes, reduction=16):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool3d(1)
        self.attn_layer = nn.Sequential(nn.Conv3d(inplanes, inplanes, kernel_size=1, stride=1, bias=True), nn.Sigmoid())

    def forward(self, x, res=None, post_act=None):
        """res / post_act (extensions used by this package's own blocks): add `res` and apply ReLU (post_act = 0.0) or
        LeakyReLU(post_act) after the gate, fused into the gate's kernels."""
        if fold3d.usable(x):
            return fold3d.se_gate(x, self.attn_layer[0], res, 0 if post_act is None else 1, post_act or 0.0)
        y = x * self.attn_layer(self.pool(x))
        if res is not None:
            y = y + res
        return y if post_act is None else F.leaky_relu(y, post_act)


def _norm3d(bn, ch):
    return nn.BatchNorm3d(ch) if bn else identity(ch)


class BasicBlock(nn.Module):
    """resnet_3D.py:108-141 (3x3x3 convs without bias, SE gate, residual)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, bn=False):
        super().__init__()
        c3 = lambda cin, s: nn.Conv3d(cin, planes, kernel_size=(3, 3, 3), stride=s, padding=1, bias=False)
        self.conv1 = nn.Sequential(c3(inplanes, stride), _norm3d(bn, planes), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(c3(planes, 1), _norm3d(bn, planes))
        self.fg = SEGating(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        if fold3d.usable(x) and isinstance(self.conv1[1], identity):
            out = fold3d.conv3d_d2(x, self.conv1[0], conv.ACT_LEAKY, 0.0)        # conv + ReLU fused
            res = x if self.downsample is None else fold3d.conv3d_d2(x, self.downsample[0])
            return self.fg(fold3d.conv3d_d2(out, self.conv2[0]), res, 0.0)      # gate, + residual, ReLU in one stage
        out = self.fg(self.conv2(self.conv1(x)))
        res = x if self.downsample is None else self.downsample(x)
        return self.relu(out + res)


class VideoResNet(nn.Module):
    """r3d_18 encoder returning all five scales (resnet_3D.py:218-292, 304-327)."""

    def __init__(self, channels, bn=False, layers=(2, 2, 2, 2)):
        super().__init__()
        self.stem = nn.Sequential(
            nn.Conv3d(3, channels[0], kernel_size=(3, 7, 7), stride=(1, 2, 2), padding=(1, 3, 3), bias=False),
            _norm3d(bn, channels[0]), nn.ReLU(inplace=True))
        inplanes = channels[0]
        spatial = [1, 2, 2, 1]   # temporal stride is 1 everywhere; layer4 keeps H, W
        for li in range(4):
            planes = channels[li]
            blocks = []
            for bi in range(layers[li]):
                ds, s = None, 1
                if bi == 0 and (spatial[li] != 1 or inplanes != planes):   # resnet_3D.py:259-266
                    s = (1, spatial[li], spatial[li])
                    ds = nn.Sequential(nn.Conv3d(inplanes, planes, kernel_size=1, stride=s, bias=False),
                                       _norm3d(bn, planes))
                blocks.append(BasicBlock(inplanes, planes, s, ds, bn))
                inplanes = planes
            setattr(self, "layer%d" % (li + 1), nn.Sequential(*blocks))
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        if fold3d.usable(x) and isinstance(self.stem[1], identity):
            x0 = fold3d.conv3d_d2(x, self.stem[0], conv.ACT_LEAKY, 0.0)          # 3x7x7 stride (1,2,2) + ReLU
        else:
            x0 = self.stem(x)
        x1 = self.layer1(x0)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        return x0, x1, x2, x3, self.layer4(x3)


def r3d_18(bn=False, channels=(32, 64, 96, 128), **_):
    return VideoResNet(list(channels), bn)


class Conv_3d(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size, stride=1, padding=0, bias=True, bn=False):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv3d(in_ch, out_ch, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias),
                                  SEGating(out_ch), _norm3d(bn, out_ch))

    def forward(self, x, post_act=None):
        """post_act: LeakyReLU slope applied after the block (UNet3d_18 does, model_singleframe.py:213-221), fused into the gate."""
        if fold3d.usable(x) and isinstance(self.conv[2], identity):
            return self.conv[1](fold3d.conv3d_d2(x, self.conv[0]), None, post_act)
        y = self.conv(x)
        return y if post_act is None else F.leaky_relu(y, post_act)


class upConv3D(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size, stride, padding, upmode="transpose", bn=False):
        super().__init__()
        self.upmode = upmode
        if upmode == "transpose":
            head = [nn.ConvTranspose3d(in_ch, out_ch, kernel_size=kernel_size, stride=stride, padding=padding)]
        else:
            head = [nn.Upsample(mode="trilinear", scale_factor=(1, 2, 2), align_corners=False),
                    nn.Conv3d(in_ch, out_ch, kernel_size=1, stride=1)]
        self.upconv = nn.Sequential(*head, SEGating(out_ch), _norm3d(bn, out_ch))

    def forward(self, x, post_act=None):
        if fold3d.usable(x) and self.upmode == "transpose" and isinstance(self.upconv[2], identity):
            return fold3d.conv_transpose3d_se(x, self.upconv[0], self.upconv[1].attn_layer[0], 0 if post_act is None else 1, post_act or 0.0)
        y = self.upconv(x)
        return y if post_act is None else F.leaky_relu(y, post_act)


class UNet3d_18(nn.Module):
    """(blurry frame, Sharp) stacked on a depth-2 axis -> residual detail (model_singleframe.py:170-223)."""

    def __init__(self, channels=[32, 64, 96, 128], bn=True):
        super().__init__()
        self.channels = channels
        c0, c1, c2, c3 = channels
        self.lrelu = nn.LeakyReLU(0.2, True)
        self.encoder = r3d_18(bn=bn, channels=channels)
        up = dict(kernel_size=(3, 4, 4), stride=(1, 2, 2), padding=(1, 1, 1), upmode="transpose", bn=bn)
        self.decoder = nn.Sequential(
            Conv_3d(c3, c2, kernel_size=3, padding=1, bias=True, bn=bn),
            upConv3D(2 * c2, c1, **up),
            upConv3D(2 * c1, c0, **up),
            Conv_3d(2 * c0, c0, kernel_size=3, padding=1, bias=True, bn=bn),
            upConv3D(2 * c0, c0, **up))
        self.feature_fuse = nn.Sequential(nn.Conv2d(2 * c0, c0, kernel_size=1, stride=1, bias=False),
                                          nn.BatchNorm2d(c0) if bn else identity())
        self.outconv = nn.Sequential(nn.ReflectionPad2d(3), nn.Conv2d(c0, 3, kernel_size=7, stride=1, padding=0))

    @staticmethod
    def _fuse_weight_on_depth_minor_channels(w):
        """feature_fuse's 1x1 weight [c0, 2*c0, 1, 1] on the channel order of cat(unbind(y, 2), 1) (index d*c0 + c) -> the same
        map on the channels of y's own memory, [B, c0, 2, H, W] == [B, 2*c0, H, W] with index c*2 + d: a permutation of the
        weight's input columns (a 0/1 fold like the Conv3d ones: the bank applies it, autograd routes the gradient back)."""
        co, c2 = w.shape[0], w.shape[1]
        return w.reshape(co, 2, c2 // 2, 1, 1).permute(0, 2, 1, 3, 4).reshape(co, c2, 1, 1)

    def _ebfi_bank_register(self, bank):
        ff = self.feature_fuse[0]
        if isinstance(self.feature_fuse[1], identity) and ff.bias is None and ff.kernel_size == (1, 1) and ff.in_channels % 2 == 0:
            bank.register(ff.weight, None, "fuse_d2", self._fuse_weight_on_depth_minor_channels)

    def forward(self, img0, img1):
        skips = self.encoder(torch.stack((img0, img1), dim=2))
        y = skips[4]
        slope = self.lrelu.negative_slope
        for stage, skip in zip(self.decoder[:4], (skips[3], skips[2], skips[1], skips[0])):
            y = torch.cat([stage(y, slope), skip], dim=1)          # LeakyReLU(0.2) fused into the stage's gate
        y = self.decoder[4](y, slope)
        ff, oc = self.feature_fuse[0], self.outconv[1]
        from . import weightbank
        site = weightbank.lookup(ff.weight, "fuse_d2") if N.dev_env("EBFI_NO_FUSE_D2", "0") != "1" else None
        if site is not None and y.is_contiguous() and conv.site_usable(site, y.reshape(y.shape[0], -1, y.shape[3], y.shape[4])) and \
                not torch.is_autocast_enabled():
            # cat(unbind(y, 2), 1) is a channel permutation of y's own memory: the 1x1 fuse reads y in place through the bank's
            # column-permuted weight image instead of a 67 MB copy forward and another backward
            y2 = y.reshape(y.shape[0], -1, y.shape[3], y.shape[4])
            y = conv.conv_site(y2, site, 0, conv.ACT_LEAKY, 0.2, [ff.weight], [])          # 1x1 fuse + LeakyReLU(0.2)
            pad = self.outconv[0].padding
            from . import fused
            yp = fused.reflect_pad2d(y, int(pad[0])) if isinstance(self.outconv[0], nn.ReflectionPad2d) and len(set(pad)) == 1 else self.outconv[0](y)
            return conv.conv_bias_act(yp, oc.weight, oc.bias, 1, 0, conv.ACT_NONE, 0.0)
        y = torch.cat(torch.unbind(y, 2), 1)
        if isinstance(self.feature_fuse[1], identity) and conv.supported(y, ff.weight, ff.stride, ff.padding) and \
                not torch.is_autocast_enabled():
            y = conv.conv_bias_act(y, ff.weight, None, 1, 0, conv.ACT_LEAKY, 0.2)  # 1x1 fuse + LeakyReLU(0.2)
            pad = self.outconv[0].padding
            if isinstance(self.outconv[0], nn.ReflectionPad2d) and len(set(pad)) == 1:
                from . import fused
                yp = fused.reflect_pad2d(y, int(pad[0]))            # (deterministic adjoint: no atomics in the step's backward)
            else:
                yp = self.outconv[0](y)
            return conv.conv_bias_act(yp, oc.weight, oc.bias, 1, 0, conv.ACT_NONE, 0.0)
        conv.left_native(y, ff.weight, ff.stride, ff.padding)        # (autocast / a fuse layer with a norm: torch convolutions)
        return self.outconv(self.lrelu(self.feature_fuse(y)))


# ---------------------------------------------------------------------------- the network
class EVFIAutoEx(BaseModel):
    def __init__(self, FrameBasech=64, EventBasech=64, InterCH=64, TB=16, norm=None, activation="LeakyReLU",
                 # exposure decision
                 BlurryFashion="DarkCh", BLInch=1, UseEvents=True, UseGTEx=False, FixEx=None, LoadPretrainEX=False,
                 PretrainedEXPath=None, FrozenEX=False,
                 # time-exposure control
                 step=32, DualPath=True,
                 # modification
                 residual=True,
                 # detail restoration
                 DetailEnabled=True, channels=[32, 64, 96, 128]):
        super().__init__()
        self.TB, self.UseGTEx, self.FixEx = TB, UseGTEx, FixEx
        self.BlurryFashion, self.DetailEnabled = BlurryFashion, DetailEnabled

        self.FrameFeatExtract = _conv(3, FrameBasech, 3, 2, 1, norm, activation)
        self.EventFeatExtract = _conv(2 * TB, EventBasech, 3, 2, 1, norm, activation)
        if not UseGTEx and not FixEx and UseEvents:
            self.ExposureDecision = ExposureDecision(EventInch=2 * TB, BLInch=BLInch, InterCH=InterCH, Group=4,
                                                     norm=norm, activation=activation, LoadPretrain=LoadPretrainEX,
                                                     PretrainedEXPath=PretrainedEXPath, Frozen=FrozenEX)
        if DualPath:
            self.ResidualControl = ResidualControl(BLinch=1, Tinch=1, Basech=EventBasech, step=step, norm=norm,
                                                   activation=activation)
        if residual:
            self.Modification = Modification(FrameBasech=FrameBasech, EventBasech=EventBasech, TB=TB, KernelSize=5,
                                             norm=norm, activation=activation)
        self.Reconstruction = nn.Sequential(
            nn.Sequential(_conv(FrameBasech, 4 * FrameBasech, 3, 1, 1, norm, None), nn.PixelShuffle(2),
                          nn.LeakyReLU(inplace=True)),
            _conv(FrameBasech, FrameBasech, 3, 1, 1, norm, activation),
            _conv(FrameBasech, 3, 3, 1, 1, norm, "Sigmoid"))
        if DetailEnabled:
            self.Detail = UNet3d_18(channels=channels, bn=False)
        initialize_weights([self.FrameFeatExtract, self.EventFeatExtract, self.Reconstruction], 0.1)

    def _reconstruct(self, x):
        """self.Reconstruction(x) (model_singleframe.py:257-266) with the LeakyReLU that follows the PixelShuffle applied in the
        epilogue of the 64 -> 256 conv instead (an elementwise activation commutes with the shuffle): two passes over the
        [B,64,H,W] map less in each direction."""
        head = self.Reconstruction[0]
        up = head[0]
        if isinstance(up, ConvLayer) and up.activation is None and isinstance(head[1], nn.PixelShuffle) and \
                isinstance(head[2], nn.LeakyReLU) and up.native(x) is not None:
            c = up.conv2d
            nxt = self.Reconstruction[1]
            if isinstance(nxt, ConvLayer) and head[1].upscale_factor == 2 and nxt.norm not in ("BN", "IN"):
                # round 6: the head's convolution stores THROUGH the shuffle and the next layer's data gradient stores through its
                # inverse (conv.SiteConvShufflePair): the [B,64,H,W] map crosses no PixelShuffle copy in either direction
                from . import weightbank
                cb, fuse_b = nxt.conv2d, conv.activation_code(nxt.activation)
                sa, sb = weightbank.lookup(c.weight, "id"), weightbank.lookup(cb.weight, "id")
                plain = lambda m: m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and m.dilation == (1, 1) and m.groups == 1
                if fuse_b is not None and fuse_b[0] in (conv.ACT_NONE, conv.ACT_LEAKY) and plain(c) and plain(cb) and \
                        conv.shuffle_pair_usable(x, sa, sb) and len(sa.w_shapes) == 1 and len(sb.w_shapes) == 1 and \
                        sa.has_bias == (c.bias is not None) and sb.has_bias == (cb.bias is not None):
                    pa = [c.weight] + ([c.bias] if c.bias is not None else [])
                    pb = [cb.weight] + ([cb.bias] if cb.bias is not None else [])
                    y = conv.conv_shuffle_pair(x, sa, float(head[2].negative_slope), pa, sb, fuse_b[0], fuse_b[1], pb)
                    return self.Reconstruction[2](y)
            y = conv.conv_bias_act(x, c.weight, c.bias, c.stride[0], c.padding[0], conv.ACT_LEAKY, float(head[2].negative_slope))
            y = F.pixel_shuffle(y, head[1].upscale_factor)
            return self.Reconstruction[2](self.Reconstruction[1](y))
        return self.Reconstruction(x)

    def LoadExposureDecision(self):
        self.ExposureDecision.load_pretrain()

    def _blurry_level(self, Frame):
        kind = self.BlurryFashion
        if kind == "DarkCh":
            return Frame2DCP(Frame)
        if kind == "Lap":
            return Frame2Lap(Frame)
        if kind == "RGB":
            return Frame
        if kind == "RGBDark":
            return torch.cat([Frame, Frame2DCP(Frame)], dim=1)
        if kind == "RGBLap":
            return torch.cat([Frame, Frame2Lap(Frame)], dim=1)
        raise Exception("Wrong blurry convertion fashion!!")

    def _exposure(self, Frame, Event, GTEx):
        if self.UseGTEx:
            assert self.FixEx is None, "set UseGTEx, but FixEx is given!"
            assert GTEx is not None, "set UseGTEx, but NO GTEx provided!"
            return GTEx
        if self.FixEx:
            assert 0 <= self.FixEx <= 1, "Wrong FixEx!"
            return torch.full((Frame.size(0), 1), float(self.FixEx), dtype=Frame.dtype, device=Frame.device)
        return self.ExposureDecision(Event, self._blurry_level(Frame))

    def encode(self, Frame, Event, GTEx=None):
        """The part of forward() that does not depend on the latent timestamp T: pad to a multiple of 8, both feature
        extractors and the exposure estimate (Frame2Lap + ExposureDecision).  The reference's inference and training loops
        call the model once per timestamp with IDENTICAL Frame / Event (infer_ours.py:113-118, train_ours.py:237-251);
        `decode(state, T)` finishes a forward from this state, and forward() is exactly decode(encode(...), T)."""
        H, W = Frame.size()[-2:]
        cropper = CropSize(W, H, {"h": 8, "w": 8}) if (H % 8 or W % 8) else None
        if cropper is not None:
            Frame, Event = cropper.pad(Frame), cropper.pad(Event)
        Event = Event.reshape(Event.size(0), -1, Event.size(3), Event.size(4))   # channel = tb*2 + polarity

        frame_feat = self.FrameFeatExtract(Frame)
        event_feat = self.EventFeatExtract(Event)
        ex = self._exposure(Frame, Event, GTEx)
        return Frame, frame_feat, event_feat, ex, cropper

    def decode(self, state, T):
        Frame, frame_feat, event_feat, ex, cropper = state
        event_feat = self.ResidualControl(event_feat, ex, T)
        Sharp = self._reconstruct(self.Modification(frame_feat, event_feat))
        if self.DetailEnabled:
            Final = Sharp + self.Detail(img0=Frame, img1=Sharp)
        else:
            Final = Sharp

        if cropper is not None:
            Sharp = cropper.crop(Sharp).contiguous()
            Final = cropper.crop(Final).contiguous() if self.DetailEnabled else Sharp
        return Sharp, Final

    def forward(self, Frame, Event, T, GTEx=None):
        return self.decode(self.encode(Frame, Event, GTEx), T)
