"""Filter-Adaptive Convolution -- host side.

Mirrors the reference's ``models/FAC/kernelconv2d/KernelConv2D.py``: ``KernelConv2DFunction``
(:12-58, ``apply(input_pad, kernel, kernel_size)`` -> output, backward returns
``(grad_input, grad_kernel, None)``) and the ``KernelConv2D`` module (:77-87, replicate-pad by
K//2 then the op).  Same argument checks, same error on CPU tensors (NotImplementedError,
:38-39,55-56); the compute is ``ebfi_fac_forward`` / ``ebfi_fac_backward`` of libebfi_hip.so.
"""
import torch
from torch import nn
from torch.autograd import Function

from . import _native as N


def fac_forward(input_pad, kernel, kernel_size, out=None):
    """Raw op on already padded input.  Any strides are accepted by the native side."""
    N.require_gpu(input_pad, kernel)
    B, C = input_pad.size(0), input_pad.size(1)
    Ho, Wo = kernel.size(2), kernel.size(3)
    if out is None:
        out = torch.empty((B, C, Ho, Wo), dtype=input_pad.dtype, device=input_pad.device)
    if out.numel() == 0:
        return out
    with torch.cuda.device_of(input_pad):
        rc = N.lib().ebfi_fac_forward(
            N.ptr(input_pad), N.i64x4(input_pad.shape), N.i64x4(input_pad.stride()),
            N.ptr(kernel), N.i64x4(kernel.shape), N.i64x4(kernel.stride()), int(kernel_size),
            N.ptr(out), N.i64x4(out.shape), N.i64x4(out.stride()),
            N.dtype_code(input_pad), N.stream_ptr(input_pad.device))
    N.check(rc, "ebfi_fac_forward")
    return out


def fac_backward(input_pad, kernel, kernel_size, grad_output, need_input=True, need_kernel=True, kernel_leaky_slope=None):
    """`kernel_leaky_slope`: the filters came out of a LeakyReLU(slope) layer and grad_kernel is wanted as the gradient of
    that layer's pre-activation (None = plain grad_kernel)."""
    N.require_gpu(input_pad, kernel, grad_output)
    gin = torch.empty_like(input_pad, memory_format=torch.contiguous_format) if need_input else None
    gk = torch.empty_like(kernel, memory_format=torch.contiguous_format) if need_kernel else None
    unit = N.i64x4((0, 0, 0, 1))
    if kernel.numel() == 0:
        return gin, gk
    with torch.cuda.device_of(input_pad):
        rc = N.lib().ebfi_fac_backward_ex(
            N.ptr(input_pad), N.i64x4(input_pad.shape), N.i64x4(input_pad.stride()),
            N.ptr(kernel), N.i64x4(kernel.shape), N.i64x4(kernel.stride()), int(kernel_size),
            N.ptr(grad_output), N.i64x4(grad_output.stride()),
            N.ptr(gin), N.i64x4(gin.stride()) if gin is not None else unit,
            N.ptr(gk), N.i64x4(gk.stride()) if gk is not None else unit,
            1.0 if kernel_leaky_slope is None else float(kernel_leaky_slope),
            N.dtype_code(input_pad), N.stream_ptr(input_pad.device))
    N.check(rc, "ebfi_fac_backward")
    return gin, gk


def fac_rows_fold_weight(w, ksize=5, tile=32):
    """[C*K*K, Cin, 3, 3] -> [C*tile, Cin, 3, 3]: one FAC channel per `tile`-row matrix tile (K*K filter rows + zero rows): the
    row layout `ebfi_kernelconv_fac_fused_x3` expects.  A 0/1 linear map (usable as a weight-bank fold)."""
    kk = ksize * ksize
    c = w.shape[0] // kk
    return torch.nn.functional.pad(w.reshape(c, kk, *w.shape[1:]), (0, 0) * (w.dim() - 1) + (0, tile - kk)).reshape(c * tile, *w.shape[1:])


def fac_rows_fold_bias(b, ksize=5, tile=32):
    kk = ksize * ksize
    return torch.nn.functional.pad(b.reshape(-1, kk), (0, tile - kk)).reshape(-1)


def kernelconv_fac_fused(cat, feat, site, kernel_size, slope):
    """filters = LeakyReLU(conv3x3(cat)) applied to the replicate-padded `feat` as ONE kernel (no filter tensor; inference
    only -- no autograd node).  `site`: the weight bank's "facrows" images of the KernelConv layer.  `cat`: the convolution's
    input, or the tuple of the two tensors whose channel concatenation it is -- on fp16 operands with an image input the
    concatenation is then written only as that image (ebfi_to_c16_cat2: 1.9 GB less traffic per HD timestamp)."""
    parts = None
    if isinstance(cat, (tuple, list)):
        parts = tuple(t.contiguous() for t in cat)
        B, H, W = int(parts[0].shape[0]), int(parts[0].shape[2]), int(parts[0].shape[3])
        Cin = sum(int(t.shape[1]) for t in parts)
        N.require_gpu(*parts, feat)
        feat = feat.contiguous()
    else:
        N.require_gpu(cat, feat)
        cat, feat = cat.contiguous(), feat.contiguous()
        B, Cin, H, W = (int(v) for v in cat.shape)
    C = int(feat.shape[1])
    if site.M != C * 32 or site.K != Cin or tuple(feat.shape) != (B, C, H, W):
        raise RuntimeError("fused KernelConv -> FAC: packed weight [%d,%d] does not match input [%d,%d,%d,%d] / feature %s"
                           % (site.M, site.K, B, Cin, H, W, tuple(feat.shape)))
    out = torch.empty_like(feat)
    book = site.bank.book
    f16 = book is not None and site.fwd16_ptr() is not None and N.dev_env("EBFI_NO_FAC_F16", "0") != "1"
    from_parts = parts is not None and len(parts) == 2 and f16 and Cin % 16 == 0 and all(int(t.shape[1]) % 8 == 0 for t in parts) and \
        N.dev_env("EBFI_NO_FAC_IMG", "0") != "1" and N.dev_env("EBFI_NO_CAT16", "0") != "1"
    if parts is not None and not from_parts:
        cat = torch.cat(parts, 1)
    if from_parts:
        # the scale from the two parts, the image from the two parts: no concatenated tensor
        from . import c16, f16scale
        i = book.slot((site.key, "x"))
        (lo0, hi0), (lo1, hi1) = torch.aminmax(parts[0]), torch.aminmax(parts[1])
        amax = torch.maximum(torch.maximum(-lo0, hi0), torch.maximum(-lo1, hi1)).float()
        e = torch.floor(torch.log2(amax.clamp_min(1e-37))) + 1.0
        scale = torch.where((amax > 0) & torch.isfinite(amax), torch.exp2(f16scale.TARGET_EXP - e), torch.ones_like(amax))
        book.slots[f16scale.SLOT_STRIDE * i:f16scale.SLOT_STRIDE * i + 1].copy_(scale.reshape(1))
        src = c16.to_c16_cat2(parts[0], parts[1], book.ptr(i))
        with torch.cuda.device_of(feat):
            rc = N.lib().ebfi_kernelconv_fac_fused_f16(N.ptr(src), 1, site.fwd16_ptr(), site.fwd16_bytes, N.ptr(site.bias()),
                                                       N.ptr(feat), N.ptr(out), B, Cin, H, W, C, int(kernel_size), float(slope), book.ptr(i),
                                                       site.w_slot_ptr(), N.stream_ptr(feat.device))
        N.check(rc, "ebfi_kernelconv_fac_fused_f16")
        return out
    if f16:
        # fp16 operands (round 6): ONE matrix-core product per tap.  The input's power-of-two scale is set from the tensor itself
        # right here (no delayed scale: an inference call has no previous step to trust) -- one min/max pass over `cat` and a
        # handful of scalar launches on the stream, all capturable; the weight image carries its own exact scale (bank refresh).
        from . import f16scale
        i = book.slot((site.key, "x"))
        lo, hi = torch.aminmax(cat)
        amax = torch.maximum(-lo, hi).float()
        e = torch.floor(torch.log2(amax.clamp_min(1e-37))) + 1.0          # amax = m * 2^e, m in [0.5, 1)
        scale = torch.where((amax > 0) & torch.isfinite(amax), torch.exp2(f16scale.TARGET_EXP - e), torch.ones_like(amax))
        book.slots[f16scale.SLOT_STRIDE * i:f16scale.SLOT_STRIDE * i + 1].copy_(scale.reshape(1))
        src, is_img = cat, 0
        if Cin % 16 == 0 and N.dev_env("EBFI_NO_FAC_IMG", "0") != "1":
            # the layer has C / 2 output-channel blocks and each of them stages the whole input: written once as the scaled fp16
            # image (one pass), every staging reads half the bytes and converts nothing
            from . import c16
            src, is_img = c16.to_c16(cat, book.ptr(i)), 1
        with torch.cuda.device_of(cat):
            rc = N.lib().ebfi_kernelconv_fac_fused_f16(N.ptr(src), is_img, site.fwd16_ptr(), site.fwd16_bytes, N.ptr(site.bias()),
                                                       N.ptr(feat), N.ptr(out), B, Cin, H, W, C, int(kernel_size), float(slope), book.ptr(i),
                                                       site.w_slot_ptr(), N.stream_ptr(cat.device))
        N.check(rc, "ebfi_kernelconv_fac_fused_f16")
        return out
    with torch.cuda.device_of(cat):
        rc = N.lib().ebfi_kernelconv_fac_fused_x3(N.ptr(cat), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(feat),
                                                  N.ptr(out), B, Cin, H, W, C, int(kernel_size), float(slope),
                                                  N.stream_ptr(cat.device))
    N.check(rc, "ebfi_kernelconv_fac_fused_x3")
    return out


class KernelConv2DFunction(Function):
    @staticmethod
    def forward(ctx, input, kernel, kernel_size, kernel_leaky_slope=None):
        """`kernel_leaky_slope` (extension of the reference signature, default None = reference behaviour): see fac_backward;
        only for a `kernel` whose ONLY consumer is this op and whose producer runs with conv.conv_bias_act(...,
        grad_is_preact=True)."""
        ctx.kernel_size = kernel_size
        ctx.kernel_leaky_slope = kernel_leaky_slope
        assert input.is_contiguous()
        assert kernel.is_contiguous()
        assert ctx.kernel_size == int((kernel.size(1) / input.size(1)) ** 0.5)
        assert input.size(2) - kernel_size == kernel.size(2) - 1
        assert input.size(3) - kernel_size == kernel.size(3) - 1
        if not input.is_cuda:
            raise NotImplementedError()  # as the reference: no CPU version
        ctx.save_for_backward(input, kernel)
        return fac_forward(input, kernel, kernel_size)

    @staticmethod
    def backward(ctx, grad_output):
        input, kernel = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        if not grad_output.is_cuda:
            raise NotImplementedError()
        gin, gk = fac_backward(input, kernel, ctx.kernel_size, grad_output,
                               ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.kernel_leaky_slope)
        return gin, gk, None, None


class KernelConv2D(nn.Module):
    def __init__(self, kernel_size):
        super().__init__()
        assert kernel_size % 2 == 1
        self.kernel_size = kernel_size
        r = (kernel_size - 1) // 2
        self.pad = nn.ReplicationPad2d([r, r, r, r])

    def forward(self, input, kernel, kernel_leaky_slope=None):
        # (self.pad stays the module the reference has; on the GPU its padding runs through the op with the deterministic adjoint)
        from . import fused
        r = (self.kernel_size - 1) // 2
        padded = fused.replicate_pad2d(input, r) if input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 else self.pad(input)
        return KernelConv2DFunction.apply(padded, kernel, self.kernel_size, kernel_leaky_slope)


# ------------------------------------------------------------------------------------------------------------------------
# Training form of KernelConv -> FAC: the two [B, 1600, h, w] tensors of the pair -- the filters and
# their gradient -- exist only as PLANAR fp16 tensors scaled by a power of two (csrc/fac.hip, include/ebfi_hip.h):
#   forward   conv 128 -> 1600 + LeakyReLU writes the filters as fp16 planes (no fp32 output), the FAC forward reads them;
#   backward  the FAC backward reads them again and writes grad_kernel (times the LeakyReLU derivative) as fp16 planes; the
#             weight gradient and the data gradient of the convolution stage those planes.
# 4.2 GB of fp32 traffic per step (B=8, 256x256) become 2.1 GB.  The forward sees filters rounded to 11 significant bits:
# Sharp moves by 4e-5 and the packed gradient by 1.7e-3 of its norm (oracle experiment) -- inside the parity
# bars (1e-3 / 5e-3), which tests/test_gpu_model.py::test_benchmarked_step_vs_oracle holds for the whole step.
def kernelconv_fac_train_usable(site, book, frame, ev, ksize):
    """frame, ev: the two parts of the KernelConv's input cat([ev, frame], 1) (the concatenation itself need not exist)"""
    if site is None or book is None or N.dev_env("EBFI_NO_C16", "0") == "1" or N.dev_env("EBFI_NO_P16", "0") == "1":
        return False
    B, Cf, H, W = frame.shape
    Cin = Cf + ev.shape[1]
    if not (frame.is_cuda and frame.dtype == torch.float32 and ev.dtype == torch.float32 and tuple(ev.shape[2:]) == (H, W) and
            ev.shape[0] == B and ksize == 5 and W % 4 == 0 and Cin % 16 == 0 and Cin == site.K and site.ks == 3 and
            site.groups == 1 and site.has_bias and site.tr16_ptr() is not None and B * ev.shape[1] <= 65535 and
            site.M == ev.shape[1] * ksize * ksize):
        return False
    return all(book.index.get((site.key, r)) in book.calibrated for r in ("x", "g", "f"))


class KernelConvFacTrain(Function):
    """apply(frame, ev, site, slope, ksize, weight, bias) -> FAC(ReplicationPad(ev), LeakyReLU(conv3x3(cat([ev, frame], 1)))).
    weight / bias: the KernelConv parameters (inputs only so that autograd routes their gradients; values come from the bank).
    The concatenation is written only as the fp16 image both convolution passes read (ebfi_to_c16_cat2) when the forward runs on
    fp16 operands; the gradient of `ev` returned here is the sum of its two paths (FAC input, first half of the concatenation)."""

    @staticmethod
    def forward(ctx, frame, ev, site, slope, ksize, weight, bias):
        from . import c16, f16scale
        book = f16scale.active_book()
        frame, ev = frame.contiguous(), ev.contiguous()
        B, C, H, W = (int(v) for v in ev.shape)
        Cin = C + int(frame.shape[1])
        lib = N.lib()
        sp = lambda role: book.ptr(book.slot((site.key, role)))
        filt16 = torch.empty((B, site.M, H, W), dtype=torch.float16, device=ev.device)
        f16_fwd = f16scale.forward_level(book) >= 1 and site.fwd16_ptr() is not None
        with torch.cuda.device_of(ev):
            st = N.stream_ptr(ev.device)
            cat = None
            if f16_fwd and C % 8 == 0 and frame.shape[1] % 8 == 0 and N.dev_env("EBFI_NO_CAT16", "0") != "1":
                cat16 = c16.to_c16_cat2(ev, frame, sp("x"))       # the image of the concatenation from its two parts
            else:
                cat = torch.cat([ev, frame], 1)
                cat16 = c16.to_c16(cat, sp("x"))      # the weight gradient's input operand (cat itself is not needed again)
            if f16_fwd:
                # fp16-operand forward (Engine(forward_f16=...)): the convolution reads the image the weight gradient will read
                # and the site's fp16 forward weight image -- one matrix-core product per tap instead of three
                rc = lib.ebfi_conv2d_packed_f16_c16(N.ptr(cat16), 1, site.fwd16_ptr(), site.fwd16_bytes, N.ptr(site.bias()), N.ptr(None),
                                                    B, Cin, H, W, site.M, 3, 1, 1, 1, float(slope), N.ptr(None), N.ptr(None), 0, 0.0,
                                                    sp("x"), site.w_slot_ptr(), N.ptr(filt16), sp("f"), 1, 0, st)
                N.check(rc, "ebfi_conv2d_packed_f16_c16 (planar fp16 filters)")
            else:
                rc = lib.ebfi_conv2d_packed_x3_c16(N.ptr(cat), site.fwd_ptr(), site.fwd_bytes, N.ptr(site.bias()), N.ptr(None), B, Cin, H,
                                                   W, site.M, 3, 1, 1, 1, float(slope), N.ptr(None), N.ptr(None), 0, 0.0, N.ptr(filt16),
                                                   sp("f"), 1, st)
                N.check(rc, "ebfi_conv2d_packed_x3_c16 (planar fp16 filters)")
            # (the replicate padding of KernelConv2D.py:82-86 happens inside the kernels: clamped reads of `ev`, no padded copy)
            out = torch.empty_like(ev)
            N.check(lib.ebfi_fac_forward_p16(N.ptr(ev), 1, N.ptr(filt16), sp("f"), N.ptr(out), B, C, H, W, int(ksize), st), "ebfi_fac_forward_p16")
        ctx.site, ctx.cfg = site, (float(slope), int(ksize), B, Cin, C, H, W)
        ctx.save_for_backward(cat16, ev, filt16)
        return out

    @staticmethod
    def backward(ctx, gout):
        from . import f16scale
        cat16, ev, filt16 = ctx.saved_tensors
        site, (slope, ksize, B, Cin, C, H, W) = ctx.site, ctx.cfg
        book = f16scale.active_book()
        if book is None:
            raise RuntimeError("KernelConv -> FAC ran its forward with fp16 filter storage: backward() must run inside the same "
                               "scale-book context (Engine._fwd_bwd)")
        gout = gout.contiguous()
        lib = N.lib()
        sp = lambda role: book.ptr(book.slot((site.key, role)))
        dev = gout.device
        with torch.cuda.device_of(gout):
            st = N.stream_ptr(dev)
            # grad wrt `ev` directly: the adjoint of the replicate padding is folded inside the kernel, in a fixed order
            gev = torch.empty_like(ev) if ctx.needs_input_grad[1] else None
            gk16 = torch.empty_like(filt16)
            N.check(lib.ebfi_fac_backward_p16(N.ptr(ev), 1, N.ptr(filt16), sp("f"), N.ptr(gout), N.ptr(gev), N.ptr(gk16), sp("g"), slope,
                                              B, C, H, W, ksize, st), "ebfi_fac_backward_p16")
            gw = torch.empty((site.M, Cin, 3, 3), dtype=torch.float32, device=dev)
            gb = torch.empty(site.M, dtype=torch.float32, device=dev)
            need = int(lib.ebfi_conv2d_backward_weight_workspace(B, Cin, H, W, site.M, 3, 1, 1, N.EBFI_F32))
            ws = torch.empty(max(need, 4), dtype=torch.uint8, device=dev)
            N.check(lib.ebfi_conv2d_backward_weight_f16c(N.ptr(cat16), N.ptr(gk16), 1, N.ptr(gw), N.ptr(gb), B, Cin, H, W, site.M, 1, sp("x"),
                                                         sp("g"), N.ptr(ws), need, st), "ebfi_conv2d_backward_weight_f16c (planar)")
            gcat = None
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                gcat = torch.empty((B, Cin, H, W), dtype=torch.float32, device=dev)
                N.check(lib.ebfi_conv2d_packed_f16_c16(N.ptr(gk16), 2, site.tr16_ptr(), site.tr16_bytes, N.ptr(None), N.ptr(gcat), B, site.M, H,
                                                       W, Cin, 3, 1, 1, 0, 0.0, N.ptr(None), N.ptr(None), 0, 0.0, sp("g"), site.w_slot_ptr(),
                                                       N.ptr(None), N.ptr(None), 0, 0, st), "ebfi_conv2d_packed_f16_c16 (planar)")
        # inputs (frame, ev): the second half of the concatenation's gradient; the first half joins the FAC path's gradient of ev
        gframe = gcat[:, C:] if gcat is not None and ctx.needs_input_grad[0] else None
        if gcat is not None and gev is not None:
            gev = gev + gcat[:, :C]
        return gframe, gev, None, None, None, gw, gb
