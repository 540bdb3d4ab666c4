"""Conv2d + bias + activation as ONE op on the gfx950 kernels of csrc/conv2d.hip.

The reference's ConvLayer (models/model_misc/submodules.py:159-200) is nn.Conv2d followed by an
activation module; here the forward is one fused kernel (`ebfi_conv2d_forward`), and the backward
runs the stride-1 data gradient on the same kernel with the activation derivative folded into its
staging (`ebfi_conv2d_backward_data`) plus a deterministic weight/bias gradient
(`ebfi_conv2d_backward_weight`).  Configurations the kernels do not cover (kernel sizes other than
1/3, dilation, groups, feature maps smaller than one tile) stay on PyTorch-ROCm's conv; that is a
GPU library path, never a CPU fallback.  Feature maps of any size >= 2 pixels take the native kernels (partial tiles are
masked): MIOpen would otherwise JIT-compile a kernel per small, unusual shape -- minutes on a fresh machine.
"""
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _native as N

ACT_NONE, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2

# Matrix-core operand precision of the 3x3 / 1x1 stride-1 convs (everything else always runs the exact fp32 kernels):
#   "fp32"    exact fp32 matrix cores everywhere
#   "bf16x3"  forward, data and weight gradient with split bf16 operands (hi + lo pairs, three MFMAs per product):
#             ~1e-5 of fp32, the parity-grade fast mode
#   "bf16"    single bf16 rounding of the operands for forward, data and weight gradient (~3e-3; parity bar 2e-2)
# fp32 tensors in memory and fp32 accumulation in every mode.
_COMPUTE = "fp32"
MODES = ("fp32", "bf16x3", "bf16")


def set_compute_dtype(mode):
    global _COMPUTE
    if mode not in MODES:
        raise ValueError("compute dtype must be one of %s" % (MODES,))
    _COMPUTE = mode


def get_compute_dtype():
    return _COMPUTE


def _bf16_ok(k, stride, mode=None):
    return (mode or _COMPUTE) == "bf16" and k in (1, 3) and stride == 1


def _x3_ok(k, stride, rows=None, mode=None):
    """rows: output rows of the product (Cout forward, Cin for the data gradient): 7x7 runs in split precision up to 16.
    mode: the compute mode to decide for (None = the current process-wide one)."""
    return (mode or _COMPUTE) == "bf16x3" and stride == 1 and (k in (1, 3) or (k == 7 and rows is not None and rows <= 16 and
                                                                               N.dev_env("EBFI_NO_CONV7X3") is None))


def _bf16_ws(lib, geo, device):
    need = int(lib.ebfi_conv2d_bf16_workspace(geo[1], geo[4], geo[5]))
    return torch.empty(max(need, 16), dtype=torch.uint8, device=device), need


def activation_code(module):
    """nn activation module -> (code, slope) or None when it cannot be fused."""
    if module is None:
        return ACT_NONE, 0.0
    if isinstance(module, torch.nn.LeakyReLU):
        return ACT_LEAKY, float(module.negative_slope)
    if isinstance(module, torch.nn.ReLU):
        return ACT_LEAKY, 0.0
    if isinstance(module, torch.nn.Sigmoid):
        return ACT_SIGMOID, 0.0
    return None


_LEFT_NATIVE = set()


def supported(x, weight, stride, padding, dilation=(1, 1), groups=1):
    """True when the convolution runs on libebfi_hip.so.  A pure predicate (model code asks it to choose a path): what happens
    when a GPU convolution is actually handed to torch is `left_native`'s business, at the one place that does it."""
    k = weight.shape[-1]
    return bool(x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and
                weight.shape[-2] == k and k in (1, 3, 7) and stride[0] == stride[1] and stride[0] in (1, 2) and
                not (k == 1 and stride[0] != 1) and padding[0] == padding[1] and 0 <= padding[0] <= k and
                tuple(dilation) == (1, 1) and groups == 1 and x.shape[-1] * x.shape[-2] >= 2)


def left_native(x, weight, stride, padding, dilation=(1, 1), groups=1):
    """Called where a GPU convolution is DISPATCHED to torch (dilation, groups through nn.Conv2d, k outside {1, 3, 7}, stride > 2,
    non-fp32 / autocast tensors: none occur in the default model) -- i.e. to MIOpen, which compiles a kernel per unusual shape
    for minutes on a fresh machine.  That must not happen silently: the first time a shape leaves the native path one line goes
    to stderr; with EBFI_STRICT_NATIVE=1 it is an EbfiNativeError instead."""
    if not x.is_cuda:
        return
    key = (tuple(weight.shape), tuple(stride), tuple(padding), tuple(dilation), int(groups), x.dtype, weight.dtype)
    msg = ("ebfi_amd.conv: convolution weight %s stride %s padding %s dilation %s groups %d (%s) is outside the native kernels' "
           "shapes and runs on torch / MIOpen" % (tuple(weight.shape), tuple(stride), tuple(padding), tuple(dilation),
                                                  int(groups), str(x.dtype).replace("torch.", "")))
    import os
    if os.environ.get("EBFI_STRICT_NATIVE") == "1":
        raise N.EbfiNativeError(msg + " (EBFI_STRICT_NATIVE=1)")
    if key not in _LEFT_NATIVE and len(_LEFT_NATIVE) < 256:
        _LEFT_NATIVE.add(key)
        import sys
        print(msg + "; set EBFI_STRICT_NATIVE=1 to make this an error", file=sys.stderr, flush=True)


def _thin_forward(x, weight, bias, out, geo, act, slope):
    """The tap-row forward of a 3x3 layer with <= 3 output channels (csrc/conv2d_thin.inc.hpp) when the shape is one: True if it
    ran.  (The C entry point decides; the test here only spares the call for the shapes that cannot qualify.)"""
    B, Cin, H, W, Cout, k, stride, pad = geo
    if not (k == 3 and stride == 1 and pad == 1 and Cout <= 3 and 16 <= Cin <= 64 and Cin % 16 == 0 and B * H * W >= 65536 and _COMPUTE == "bf16x3"):
        return False
    rc = N.lib().ebfi_conv2d_thin_forward(N.ptr(x), N.ptr(weight), N.ptr(bias), N.ptr(out), *geo, act, slope, N.stream_ptr(x.device))
    if rc == N.EBFI_ERR_UNSUPPORTED:
        return False
    N.check(rc, "ebfi_conv2d_thin_forward")
    return True


def _geo(x, weight, stride, pad):
    B, Cin, H, W = x.shape
    return [int(B), int(Cin), int(H), int(W), int(weight.shape[0]), int(weight.shape[-1]), int(stride), int(pad)]


class ConvBiasAct(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act, slope, grad_preact=False):
        """grad_preact: the gradient that will arrive in backward is already the gradient of the PRE-activation (its producer
        applied act' -- e.g. the FAC backward with kernel_leaky_slope): backward then runs as for act = none."""
        x, weight = x.contiguous(), weight.contiguous()
        geo = _geo(x, weight, stride, pad)
        k = geo[5]
        Ho, Wo = (geo[2] + 2 * pad - k) // stride + 1, (geo[3] + 2 * pad - k) // stride + 1
        out = torch.empty((geo[0], geo[4], Ho, Wo), dtype=x.dtype, device=x.device)
        lib = N.lib()
        bptr = N.ptr(bias.contiguous() if bias is not None else None)
        with torch.cuda.device_of(x):
            if _thin_forward(x, weight, bias.contiguous() if bias is not None else None, out, geo, act, slope):
                rc = 0
            elif _bf16_ok(k, stride) or _x3_ok(k, stride, geo[4]):
                ws, need = _bf16_ws(lib, geo, x.device)
                fn = lib.ebfi_conv2d_forward_bf16mma if _COMPUTE == "bf16" else lib.ebfi_conv2d_forward_bf16x3
                rc = fn(N.ptr(x), N.ptr(weight), bptr, N.ptr(out), *geo, act, slope, N.ptr(ws), need, N.stream_ptr(x.device))
            else:
                rc = lib.ebfi_conv2d_forward(N.ptr(x), N.ptr(weight), bptr, N.ptr(out), *geo, act, slope, N.EBFI_F32,
                                             N.stream_ptr(x.device))
        N.check(rc, "ebfi_conv2d_forward")
        if grad_preact:
            act, slope = ACT_NONE, 0.0
        # the compute mode is part of the node: backward runs in the mode its forward ran in, whatever the process-wide
        # switch says by then (a backward() issued outside the engine's precision context)
        ctx.cfg = (stride, pad, act, slope, bias is not None, _COMPUTE)
        ctx.save_for_backward(x, weight, out if act != ACT_NONE else None)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, weight, y = ctx.saved_tensors
        stride, pad, act, slope, has_bias, mode = ctx.cfg
        gout = gout.contiguous()
        geo = _geo(x, weight, stride, pad)
        k = geo[5]
        lib = N.lib()
        gx = gw = gb = None
        need_x = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])
        bf16 = _bf16_ok(k, stride, mode)
        with torch.cuda.device_of(x):
            st = N.stream_ptr(x.device)
            gpre = None       # grad_output * act'(y): by-product of the fp32 weight-gradient kernel
            if need_w:
                gw = torch.empty_like(weight)
                gb = torch.empty(geo[4], dtype=x.dtype, device=x.device) if has_bias else None
                need = int(lib.ebfi_conv2d_backward_weight_workspace(*geo, N.EBFI_F32))
                ws = torch.empty(max(need, 4), dtype=torch.uint8, device=x.device)
                if need_x and act != ACT_NONE and not bf16:
                    gpre = torch.empty_like(gout)
                # (the split-precision weight gradient also covers 7x7 stride 1: the detail branch's output conv)
                x3w = mode == "bf16x3" and k in (1, 3, 7) and stride == 1
                wdt = N.EBFI_F32_BF16MMA if bf16 else (N.EBFI_F32_BF16X3MMA if x3w else N.EBFI_F32)
                rc = lib.ebfi_conv2d_backward_weight_ex(N.ptr(x), N.ptr(gout), N.ptr(y), N.ptr(gw), N.ptr(gb), N.ptr(gpre),
                                                        *geo, act, slope, N.ptr(ws), need, wdt, st)
                N.check(rc, "ebfi_conv2d_backward_weight")
            if need_x:
                gx = torch.empty_like(x)
                if bf16:
                    ws, need = _bf16_ws(lib, geo, x.device)
                    rc = lib.ebfi_conv2d_backward_data_bf16mma(N.ptr(gout), N.ptr(y), N.ptr(weight), N.ptr(gx), *geo, act,
                                                               slope, N.ptr(ws), need, st)
                elif _x3_ok(k, stride, geo[1], mode) and (k != 7 or gpre is not None or act == ACT_NONE):
                    # split-precision data gradient (on grad * act' when the weight gradient left it)
                    ws, need = _bf16_ws(lib, geo, x.device)
                    src, sy, a = (gpre, None, ACT_NONE) if gpre is not None else (gout, y, act)
                    rc = lib.ebfi_conv2d_backward_data_bf16x3(N.ptr(src), N.ptr(sy), N.ptr(weight), N.ptr(gx), *geo, a,
                                                              slope if a != ACT_NONE else 0.0, N.ptr(ws), need, st)
                elif stride == 1:
                    if gpre is not None:    # derivative already folded in: plain transposed conv
                        rc = lib.ebfi_conv2d_backward_data(N.ptr(gpre), N.ptr(None), N.ptr(weight), N.ptr(gx), *geo, ACT_NONE,
                                                           0.0, N.EBFI_F32, st)
                    else:
                        rc = lib.ebfi_conv2d_backward_data(N.ptr(gout), N.ptr(y), N.ptr(weight), N.ptr(gx), *geo, act, slope,
                                                           N.EBFI_F32, st)
                else:
                    # stride 2: zero-insert grad_output (times act') and run the stride-1 data gradient on it.
                    # Only the small stems / down-sampling convs take this path.
                    if gpre is None:
                        gpre = gout if act == ACT_NONE else gout * _act_grad(y, act, slope)
                    if k == 7 and stride == 2 and _x3_ok(k, 1, geo[1], mode) and pad == 3:
                        # by output parity on the gradient itself: no zero-inserted tensor (csrc/conv2d.hip conv7s2_dgrad_x3)
                        rc = lib.ebfi_conv2d_backward_data_s2_bf16x3(N.ptr(gpre.contiguous()), N.ptr(weight), N.ptr(gx), geo[0], geo[1],
                                                                     geo[2], geo[3], geo[4], k, pad, st)
                        N.check(rc, "ebfi_conv2d_backward_data_s2_bf16x3")
                        return gx, gw, gb, None, None, None, None, None
                    uh, uw = geo[2] + 2 * pad - k + 1, geo[3] + 2 * pad - k + 1
                    up = gout.new_zeros((geo[0], geo[4], uh, uw))
                    up[:, :, ::stride, ::stride][:, :, :gout.shape[2], :gout.shape[3]] = gpre
                    geo1 = geo[:6] + [1, pad]
                    if k == 7 and _x3_ok(k, 1, geo[1], mode):
                        ws, need = _bf16_ws(lib, geo1, x.device)
                        rc = lib.ebfi_conv2d_backward_data_bf16x3(N.ptr(up), N.ptr(None), N.ptr(weight), N.ptr(gx), *geo1, ACT_NONE,
                                                                  0.0, N.ptr(ws), need, st)
                    else:
                        rc = lib.ebfi_conv2d_backward_data(N.ptr(up), N.ptr(None), N.ptr(weight), N.ptr(gx), *geo1, ACT_NONE, 0.0,
                                                           N.EBFI_F32, st)
                N.check(rc, "ebfi_conv2d_backward_data")
        return gx, gw, gb, None, None, None, None, None


def _act_grad(y, act, slope):
    if act == ACT_LEAKY:
        return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope))
    return y * (1 - y)


class SiteConvBiasAct(Function):
    """conv + bias + activation whose weight images live in the active WeightBank (ebfi_amd.weightbank): pre-split once
    per optimiser step, possibly folded (depth-2 Conv3d / ConvTranspose3d) or concatenated from several parameters.
    3x3 / 1x1, stride 1, split-precision kernels.  `params` = the site's source weights followed by its source biases:
    they are inputs only so that autograd routes the gradients; their values are read from the bank."""

    @staticmethod
    def forward(ctx, x, site, pad, act, slope, grad_preact, *params):
        x = x.contiguous()
        B, Cin, H, W = (int(v) for v in x.shape)
        if Cin != site.K:
            raise RuntimeError("input has %d channels, the packed weight expects %d" % (Cin, site.K))
        geo = [B, Cin, H, W, site.M, site.ks, 1, int(pad)]
        Ho, Wo = H + 2 * pad - site.ks + 1, W + 2 * pad - site.ks + 1
        out = torch.empty((B, site.M, Ho, Wo), dtype=x.dtype, device=x.device)
        with torch.cuda.device_of(x):
            # (a plain single-weight site of a thin layer: the tap-row kernel reads the fp32 weight itself)
            if site.kind == "id" and len(site.w_shapes) == 1 and site.groups == 1 and \
                    _thin_forward(x, params[0].contiguous(), site.bias(), out, geo, act, slope):
                rc = 0
            else:
                rc = N.lib().ebfi_conv2d_forward_bf16x3(N.ptr(x), N.ptr(None), N.ptr(site.bias()), N.ptr(out), *geo, act, slope,
                                                        site.fwd_ptr(), site.fwd_bytes, N.stream_ptr(x.device))
        N.check(rc, "ebfi_conv2d_forward_bf16x3 (packed)")
        if grad_preact:                      # the incoming gradient is already that of the pre-activation
            act, slope = ACT_NONE, 0.0
        ctx.site, ctx.cfg, ctx.geo = site, (act, slope), geo
        ctx.save_for_backward(x, out if act != ACT_NONE else None)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, y = ctx.saved_tensors
        site, (act, slope), geo = ctx.site, ctx.cfg, ctx.geo
        gx, pgrads = _site_backward(site, geo, act, slope, x, y, gout, ctx.needs_input_grad[0], any(ctx.needs_input_grad[6:]))
        return (gx, None, None, None, None, None) + tuple(pgrads)


def _site_backward(site, geo, act, slope, x, y, gout, need_x, need_p, unshuffle=None):
    """Backward of one bank-site convolution (SiteConvBiasAct): returns (grad_x or None, [gradients of the site's source weights, then
    of its source biases]).  unshuffle = (mask, mask_slope): the layer's input came out of PixelShuffle(2) + LeakyReLU(mask_slope)
    behind another convolution (mask = that input, `x`): grad_x leaves as the PRE-activation gradient of that convolution, in its
    layout [B, 4*Cin, H/2, W/2] -- the data-gradient kernel applies LeakyReLU'(mask) and stores through the inverse shuffle
    (ebfi_conv2d_packed_f16_shuffled), so neither the pixel-unshuffle copy nor a mask pass over the saved activation runs."""
    gout = gout.contiguous()
    lib = N.lib()
    gx, pgrads = None, [None] * (len(site.w_shapes) + len(site.b_shapes))
    # fp16 single-product backward (ebfi_amd.f16scale, csrc/conv2d_f16.inc.hpp) where the kernels apply: 3x3 same-padded
    # layers; weight gradient with 64-channel input blocks, data gradient with quad-aligned rows and >= 48 input channels
    from . import f16scale
    book = f16scale.active_book()
    B, Cin, H, W, M, ks, _, pad = geo
    f16 = book is not None and ks == 3 and pad == 1
    # (partial 64-channel blocks -- the 32 / 48-channel layers of the detail branch -- go through the pixel-major kernel, which
    # needs quad-aligned rows; below 32 channels the zero-filled half of the block would be most of the work)
    # (the pixel-major kernel -- the only one for ragged Cin -- stages 16-byte quads: operands that are unaligned views
    # take the split-precision kernel instead of failing with EBFI_ERR_UNSUPPORTED)
    al16 = all(t is None or t.data_ptr() % 16 == 0 for t in (x, gout, y))
    f16_w = f16 and (Cin % 64 == 0 or (Cin >= 32 and W % 4 == 0 and al16 and N.dev_env("EBFI_WGRAD_TR", "1") != "0"))
    f16_x = f16 and W % 4 == 0 and Cin >= 48 and site.tr16_ptr() is not None
    gpre = None

    def weight_gradients(st_w):
        """weight / bias gradient launches (+ slab reduction, + fold gathers) on stream `st_w`; returns (pgrads, gpre)."""
        gw2 = torch.empty((site.M, site.K, site.ks, site.ks), dtype=x.dtype, device=x.device)
        gb2 = torch.empty(site.M, dtype=x.dtype, device=x.device) if site.has_bias else None
        need = int(lib.ebfi_conv2d_backward_weight_workspace(*geo, N.EBFI_F32))
        ws = torch.empty(max(need, 4), dtype=torch.uint8, device=x.device)
        # grad * act' for the data gradient: as the fp16 image that kernel stages (ebfi_amd.c16) when both gradients run on fp16
        # operands and the pixel-major weight-gradient kernel applies -- half the bytes written and read again
        gpre16 = need_x and act != ACT_NONE and f16_w and f16_x and al16 and W % 4 == 0 and M % 16 == 0 and site.groups == 1 and \
            N.dev_env("EBFI_NO_GPRE16", "0") != "1" and N.dev_env("EBFI_WGRAD_TR", "1") != "0"
        gp = None
        if need_x and act != ACT_NONE:
            if gpre16:
                from . import c16
                gp = c16.empty(B, M, H, W, x.device)
            else:
                gp = torch.empty_like(gout)
        if f16_w:
            rc = lib.ebfi_conv2d_backward_weight_f16g_ex(N.ptr(x), N.ptr(gout), N.ptr(y), N.ptr(gw2), N.ptr(gb2), N.ptr(gp),
                                                         1 if gpre16 else 0, B, Cin, H, W, M, ks, pad, 1, act, slope,
                                                         book.operand((site.key, "x"), x), book.operand((site.key, "g"), gout),
                                                         N.ptr(ws), need, st_w)
        else:
            rc = lib.ebfi_conv2d_backward_weight_ex(N.ptr(x), N.ptr(gout), N.ptr(y), N.ptr(gw2), N.ptr(gb2), N.ptr(gp), *geo,
                                                    act, slope, N.ptr(ws), need, N.EBFI_F32_BF16X3MMA, st_w)
        N.check(rc, "ebfi_conv2d_backward_weight")
        return _route_site_grads(site, gw2, gb2, st_w), gp

    with torch.cuda.device_of(x):
        st = N.stream_ptr(x.device)
        if need_p:
            pgrads, gpre = weight_gradients(st)
        if need_x:
            src, sy, a = (gpre, None, ACT_NONE) if gpre is not None else (gout, y, act)
            img = gpre is not None and gpre.dtype == torch.float16
            # (the shuffled store is the fp16 data-gradient kernel's: ungrouped, even sizes, the mask 16-byte aligned like the rest)
            native_unshuffle = unshuffle is not None and (img or (f16_x and a == ACT_NONE)) and site.groups == 1 and H % 2 == 0 and \
                W % 4 == 0 and Cin > 32 and unshuffle[0].is_contiguous() and unshuffle[0].data_ptr() % 16 == 0 and \
                N.dev_env("EBFI_NO_CONV_SHUFFLE", "0") != "1"
            if native_unshuffle:
                gx = torch.empty((B, 4 * Cin, H // 2, W // 2), dtype=x.dtype, device=x.device)
                in_slot = book.ptr(book.slot((site.key, "g"))) if img else book.operand((site.key, "g"), src)
                rc = lib.ebfi_conv2d_packed_f16_shuffled(N.ptr(src), 1 if img else 0, site.tr16_ptr(), site.tr16_bytes, N.ptr(None), N.ptr(gx),
                                                         B, M, H, W, Cin, ACT_NONE, 0.0, N.ptr(unshuffle[0]), ACT_LEAKY, float(unshuffle[1]),
                                                         in_slot, site.w_slot_ptr(), 2, st)
                N.check(rc, "ebfi_conv2d_packed_f16_shuffled (data gradient through the inverse pixel shuffle)")
                return gx, pgrads
            gx = torch.empty_like(x)
            if img:
                # the image the weight gradient just wrote (scale: the slot it recorded into)
                rc = lib.ebfi_conv2d_packed_f16_c16(N.ptr(gpre), 1, site.tr16_ptr(), site.tr16_bytes, N.ptr(None), N.ptr(gx), B,
                                                    M // site.groups, H, W, Cin * site.groups, ks, pad, site.groups, ACT_NONE, 0.0,
                                                    N.ptr(None), N.ptr(None), 0, 0.0, book.ptr(book.slot((site.key, "g"))),
                                                    site.w_slot_ptr(), N.ptr(None), N.ptr(None), 0, 0, st)
                N.check(rc, "ebfi_conv2d_packed_f16_c16 (data gradient from the image of grad * act')")
            elif f16_x and a == ACT_NONE:
                # the data gradient as a convolution of the pre-activation gradient with the transposed fp16 image
                rc = lib.ebfi_conv2d_packed_f16(N.ptr(src), site.tr16_ptr(), site.tr16_bytes, N.ptr(None), N.ptr(gx), B,
                                                M // site.groups, H, W, Cin * site.groups, ks, pad, site.groups, ACT_NONE, 0.0,
                                                N.ptr(None), N.ptr(None), 0, 0.0, book.operand((site.key, "g"), src),
                                                site.w_slot_ptr(), st)
                N.check(rc, "ebfi_conv2d_packed_f16 (data gradient)")
            else:
                rc = lib.ebfi_conv2d_backward_data_bf16x3(N.ptr(src), N.ptr(sy), N.ptr(None), N.ptr(gx), *geo, a,
                                                          slope if a != ACT_NONE else 0.0, site.tr_ptr(), site.tr_bytes, st)
                N.check(rc, "ebfi_conv2d_backward_data_bf16x3 (packed)")
            if unshuffle is not None:            # (kernels without the shuffled store: mask and inverse shuffle as torch ops)
                m, ms = unshuffle
                gx = torch.nn.functional.pixel_unshuffle(gx * torch.where(m > 0, 1.0, float(ms)), 2)
    return gx, pgrads


class SiteConvShufflePair(Function):
    """conv A (3x3, bank site) -> PixelShuffle(2) -> LeakyReLU(slope_a) -> conv B (3x3, bank site, act_b): the reconstruction head
    (models/Ours/model_singleframe.py:257-262) as ONE autograd node, so that no tensor crosses the shuffle twice: A's epilogue
    applies the LeakyReLU and stores through the shuffle (ebfi_conv2d_packed_x3_shuffled); B's data gradient applies
    LeakyReLU'(B's input) and stores through the inverse shuffle (_site_backward, unshuffle=), handing A its PRE-activation
    gradient in A's own layout.  params = A's source weights and biases (`na` tensors), then B's."""

    @staticmethod
    def forward(ctx, x, site_a, slope_a, site_b, act_b, slope_b, na, *params):
        x = x.contiguous()
        B, Cin, H, W = (int(v) for v in x.shape)
        if Cin != site_a.K or site_a.M != 4 * site_b.K:
            raise RuntimeError("shuffled pair: %d -> %d channels, then %d -> %d" % (Cin, site_a.M, site_b.K, site_b.M))
        geo_a = [B, Cin, H, W, site_a.M, 3, 1, 1]
        geo_b = [B, site_b.K, 2 * H, 2 * W, site_b.M, 3, 1, 1]
        y1 = torch.empty((B, site_b.K, 2 * H, 2 * W), dtype=x.dtype, device=x.device)
        y2 = torch.empty((B, site_b.M, 2 * H, 2 * W), dtype=x.dtype, device=x.device)
        with torch.cuda.device_of(x):
            st = N.stream_ptr(x.device)
            rc = N.lib().ebfi_conv2d_packed_x3_shuffled(N.ptr(x), site_a.fwd_ptr(), site_a.fwd_bytes, N.ptr(site_a.bias()), N.ptr(y1), B, Cin,
                                                        H, W, site_a.M, ACT_LEAKY, slope_a, 1, st)
            N.check(rc, "ebfi_conv2d_packed_x3_shuffled")
            rc = N.lib().ebfi_conv2d_forward_bf16x3(N.ptr(y1), N.ptr(None), N.ptr(site_b.bias()), N.ptr(y2), *geo_b, act_b, slope_b,
                                                    site_b.fwd_ptr(), site_b.fwd_bytes, st)
            N.check(rc, "ebfi_conv2d_forward_bf16x3 (packed)")
        ctx.sites, ctx.cfg, ctx.geos, ctx.na = (site_a, site_b), (slope_a, act_b, slope_b), (geo_a, geo_b), na
        ctx.save_for_backward(x, y1, y2 if act_b != ACT_NONE else None)
        return y2

    @staticmethod
    def backward(ctx, g2):
        x, y1, y2 = ctx.saved_tensors
        (site_a, site_b), (slope_a, act_b, slope_b), (geo_a, geo_b), na = ctx.sites, ctx.cfg, ctx.geos, ctx.na
        need = ctx.needs_input_grad
        need_pa, need_pb = any(need[7:7 + na]), any(need[7 + na:])
        need_a = need[0] or need_pa
        ga, pg_b = _site_backward(site_b, geo_b, act_b, slope_b, y1, y2, g2, need_a, need_pb, unshuffle=(y1, slope_a))
        gx, pg_a = (None, [None] * na)
        if need_a:                               # (A's activation derivative is already in ga)
            gx, pg_a = _site_backward(site_a, geo_a, ACT_NONE, 0.0, x, None, ga, need[0], need_pa)
        return (gx, None, None, None, None, None, None) + tuple(pg_a) + tuple(pg_b)


def shuffle_pair_usable(x, site_a, site_b):
    """The shuffled store is the wave-specialised 3x3 kernel's: quad-aligned rows, more than 32 output channels (csrc/conv2d.hip)."""
    return (site_usable(site_a, x) and site_b is not None and site_a.ks == 3 and site_b.ks == 3 and site_a.kind == "id" and
            site_b.kind == "id" and site_a.groups == 1 and site_b.groups == 1 and site_a.M == 4 * site_b.K and site_a.M > 32 and
            x.shape[3] % 4 == 0 and x.data_ptr() % 16 == 0 and not torch.is_autocast_enabled() and
            N.dev_env("EBFI_NO_CONV_SHUFFLE", "0") != "1")


def conv_shuffle_pair(x, site_a, slope_a, params_a, site_b, act_b, slope_b, params_b):
    return SiteConvShufflePair.apply(x, site_a, float(slope_a), site_b, int(act_b), float(slope_b), len(params_a), *params_a, *params_b)


def _route_site_grads(site, gw2, gb2, st):
    """Gradient of the folded / concatenated weight [M,K,ks,ks] (and bias [M]) -> gradients of the source parameters."""
    lib = N.lib()
    out = []
    if site.kind == "id":                     # plain (or row-concatenated) Conv2d weights: the rows ARE the parameters
        row = 0
        for shp in site.w_shapes:
            out.append(gw2[row:row + shp[0]].view(shp))
            row += shp[0]
        row = 0
        for shp in site.b_shapes:
            out.append(gb2[row:row + shp[0]].view(shp))
            row += shp[0]
        return out
    for shp, inv, R in zip(site.w_shapes, site.w_inv, site.w_R):
        g = torch.empty(shp, dtype=gw2.dtype, device=gw2.device)
        N.check(lib.ebfi_gather_sum(N.ptr(gw2), N.ptr(inv), N.ptr(g), g.numel(), R, st), "ebfi_gather_sum")
        out.append(g)
    for shp, inv, R in zip(site.b_shapes, site.b_inv, site.b_R):
        g = torch.empty(shp, dtype=gb2.dtype, device=gb2.device)
        N.check(lib.ebfi_gather_sum(N.ptr(gb2), N.ptr(inv), N.ptr(g), g.numel(), R, st), "ebfi_gather_sum")
        out.append(g)
    return out


def site_usable(site, x, stride=1):
    return (site is not None and _COMPUTE == "bf16x3" and int(stride) == 1 and x.is_cuda and x.dtype == torch.float32 and
            x.dim() == 4 and x.shape[1] == site.K)


def conv_site(x, site, padding, act, slope, weights, biases, grad_preact=False):
    """The convolution of bank site `site` (weights / biases: its source parameters, for gradient routing)."""
    return SiteConvBiasAct.apply(x, site, int(padding), int(act), float(slope), bool(grad_preact), *weights, *biases)


def conv_bias_act(x, weight, bias, stride=1, padding=0, act=ACT_NONE, slope=0.0, grad_preact=False):
    from . import weightbank
    site = weightbank.lookup(weight, "id")
    if site_usable(site, x, stride) and (bias is not None) == site.has_bias and len(site.w_shapes) == 1:
        return conv_site(x, site, padding, act, slope, [weight], [bias] if bias is not None else [], grad_preact)
    return ConvBiasAct.apply(x, weight, bias, int(stride), int(padding), int(act), float(slope), bool(grad_preact))
