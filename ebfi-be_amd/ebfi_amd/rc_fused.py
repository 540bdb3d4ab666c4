"""ResidualControl as ONE hand-scheduled autograd node (reference models/Ours/model_singleframe.py:79-136).

The module is 48 % of the model's FLOPs in 60 small 3x3 convolutions.  Run layer by layer through autograd it costs, per
round, 5 forward + 5 data-gradient + 5 weight-gradient convolutions, two gradient-accumulation adds for the three
consumers of x, a saved-activation read in every weight / data gradient, and a side output of grad*act' per layer.  Here
one round is

    forward   ya = lrelu(conv(x; [W3a | W4a]))        one convolution to 128 channels (Conv3[i][0] and Conv4[i][0] share x)
              a  = lrelu(gconv(ya; [W3b | W4b]))      one grouped convolution (groups = 2) for Conv3[i][1] / Conv4[i][1]
              c  = cat(s_ex*a[:, :64] + x, s_t*a[:, 64:] + x)      fused stage, reads the halves of `a` in place
              x' = lrelu(conv(c; W5))
    backward  everything flows as PRE-activation gradients: the data gradient of each layer applies, in its epilogue, the
              derivative of the activation that produced its input (and adds the residual path), so no kernel re-reads a
              saved output for act', none writes a grad*act' side tensor, and no add kernel runs

on the weight images of the active weight bank (split-precision mode).  The per-round scales s_ex = Conv1[i](Ex),
s_t = Conv2[i](T) stay ordinary autograd tensors (computed for all rounds at once by the module).
Not active (fp32 / bf16 modes, no bank, CPU tensors): the module falls back to its layer-by-layer form.
"""
import torch
from torch.autograd import Function

from . import _native as N
from . import conv, weightbank

ACT = conv.ACT_LEAKY


def _plain_layer(m, ks, cin=None):
    """ConvLayer `m` is exactly conv(ks x ks, stride 1, same padding, bias) + LeakyReLU with no norm layer: the only form the
    fused node computes.  Anything else (norm='BN' drops the bias, norm='IN' adds a norm layer, another activation or
    kernel size) must run layer by layer."""
    c = m.conv2d
    return (getattr(m, "norm", None) is None and isinstance(m.activation, torch.nn.LeakyReLU) and c.bias is not None and
            tuple(c.kernel_size) == (ks, ks) and tuple(c.stride) == (1, 1) and tuple(c.padding) == (ks // 2, ks // 2) and
            tuple(c.dilation) == (1, 1) and c.groups == 1 and (cin is None or c.in_channels == cin))


def fusable(rc):
    """Every layer of the module has the plain form the fused node assumes (checked once per call: a few attribute reads)."""
    for i in range(rc.step):
        if not (_plain_layer(rc.Conv1[i][0], 1) and _plain_layer(rc.Conv2[i][0], 1)):
            return False
        for bank in (rc.Conv3, rc.Conv4):
            if len(bank[i]) != 2 or not all(_plain_layer(m, 3) for m in bank[i]):
                return False
        if len(rc.Conv5[i]) != 1 or not _plain_layer(rc.Conv5[i][0], 3):
            return False
    slopes = {float(m.activation.negative_slope) for bank in (rc.Conv1, rc.Conv2, rc.Conv3, rc.Conv4, rc.Conv5)
              for seq in bank for m in seq}
    return len(slopes) == 1


def register(bank, rc):
    """Sites of one ResidualControl module: concatenated first layers, grouped second layers (Conv5 is a plain site).
    A module that is not `fusable` registers nothing and keeps running layer by layer."""
    if not fusable(rc):
        return
    for i in range(rc.step):
        a3, a4 = rc.Conv3[i][0].conv2d, rc.Conv4[i][0].conv2d
        b3, b4 = rc.Conv3[i][1].conv2d, rc.Conv4[i][1].conv2d
        bank.register([a3.weight, a4.weight], [a3.bias, a4.bias], kind="rcA")
        bank.register([b3.weight, b4.weight], [b3.bias, b4.bias], kind="rcB", groups=2)


def sites_of(rc):
    """[(siteA, siteB, siteC)] per round under the active bank, or None when any is missing."""
    out = []
    for i in range(rc.step):
        sa = weightbank.lookup(rc.Conv3[i][0].conv2d.weight, "rcA")
        sb = weightbank.lookup(rc.Conv3[i][1].conv2d.weight, "rcB")
        sc = weightbank.lookup(rc.Conv5[i][0].conv2d.weight, "id")
        if sa is None or sb is None or sc is None or not sc.has_bias:
            return None
        out.append((sa, sb, sc))
    return out


def usable(rc, x):
    if conv.get_compute_dtype() != "bf16x3" or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        return False
    if (x.shape[2] * x.shape[3]) % 4 != 0 or torch.is_autocast_enabled():
        return False
    if x.shape[1] % 64 != 0:          # a workgroup's 64 output channels must lie inside one group of the grouped layers
        return False
    return weightbank.active_bank() is not None and fusable(rc)


def _conv(lib, st, x, packed, nbytes, bias, out, B, cin_g, H, W, cout, groups, act, slope, addend=None, mask=None):
    rc = lib.ebfi_conv2d_packed_x3(N.ptr(x), packed, nbytes, N.ptr(bias), N.ptr(out), B, cin_g, H, W, cout, 3, 1, groups, act, slope,
                                   N.ptr(addend), N.ptr(mask), ACT if mask is not None else 0, slope if mask is not None else 0.0, st)
    N.check(rc, "ebfi_conv2d_packed_x3")


def _wgrad(lib, st, x, g, B, cin_g, H, W, cout, groups, ws_cache, book=None, site=None):
    """Weight / bias gradient of a (grouped) layer from its pre-activation gradient; `book` (fp16 backward active): the
    single-product kernel with the operand scales of slots (site, 'x') / (site, 'g')."""
    gw = torch.empty((cout, cin_g, 3, 3), dtype=x.dtype, device=x.device)
    gb = torch.empty(cout, dtype=x.dtype, device=x.device)
    key = (cin_g, cout)
    if key not in ws_cache:
        need = int(lib.ebfi_conv2d_backward_weight_workspace(B, cin_g, H, W, cout, 3, 1, 1, N.EBFI_F32))
        ws_cache[key] = (torch.empty(max(need, 4), dtype=torch.uint8, device=x.device), need)
    ws, need = ws_cache[key]
    if book is not None and cin_g % 64 == 0:
        rc = lib.ebfi_conv2d_backward_weight_f16g(N.ptr(x), N.ptr(g), N.ptr(None), N.ptr(gw), N.ptr(gb), N.ptr(None), B, cin_g, H, W,
                                                  cout, 3, 1, groups, 0, 0.0, book.operand((site.key, "x"), x),
                                                  book.operand((site.key, "g"), g), N.ptr(ws), need, st)
        N.check(rc, "ebfi_conv2d_backward_weight_f16g")
        return gw, gb
    rc = lib.ebfi_conv2d_backward_weight_x3g(N.ptr(x), N.ptr(g), N.ptr(gw), N.ptr(gb), B, cin_g, H, W, cout, 3, 1, groups,
                                             N.ptr(ws), need, st)
    N.check(rc, "ebfi_conv2d_backward_weight_x3g")
    return gw, gb


def _dgrad(lib, st, g, site, out, B, cin_g, H, W, cout, groups, slope, addend=None, mask=None, book=None):
    """Data gradient of `site`'s layer as a convolution of the pre-activation gradient `g` with the transposed images
    (+ addend, * act'(mask): the epilogue extras that hand a PRE-activation gradient to the layer below)."""
    if book is not None and W % 4 == 0 and site.tr16_ptr() is not None:
        rc = lib.ebfi_conv2d_packed_f16(N.ptr(g), site.tr16_ptr(), site.tr16_bytes, N.ptr(None), N.ptr(out), B, cin_g, H, W, cout, 3, 1,
                                        groups, 0, slope, N.ptr(addend), N.ptr(mask), ACT if mask is not None else 0,
                                        slope if mask is not None else 0.0, book.operand((site.key, "g"), g), site.w_slot_ptr(), st)
        N.check(rc, "ebfi_conv2d_packed_f16")
        return
    _conv(lib, st, g, site.tr_ptr(), site.tr_bytes, None, out, B, cin_g, H, W, cout, groups, 0, slope, addend, mask)


# ---- fp16 operand storage (ebfi_amd.c16, csrc/c16.hpp): the tensors the backward convolutions stage are written as scaled fp16
# images by whoever produces them, the backward kernels copy 16-byte pieces instead of converting fp32 planes
def _slot_keys(sites):
    keys = []
    for sa, sb, sc in sites:
        keys += [(sa.key, "x"), (sa.key, "g"), (sb.key, "x"), (sb.key, "g"), (sc.key, "x"), (sc.key, "g")]
    return keys


def _tail_in_epilogue(book, sites):
    """Round 5: the round's tail (scale, residual, concat) runs in the epilogue of its grouped convolution (csrc/conv2d.hip
    EpiExtra post_scale / pre16) once the slot of the activation image `a` has been calibrated by the layer-wise form."""
    if N.dev_env("EBFI_NO_RC_EPILOGUE", "0") == "1":          # (development switch: same-box A/B against the separate stage)
        return False
    return all(book.index.get((sb.key, "a")) in book.calibrated for _, sb, _ in sites)


def images_usable(book, sites, W):
    """The image path needs every operand slot of the module calibrated (the layer-wise form below measures them just in time
    on the first eager pass and during the engine's calibration steps) and rows that split into 16-byte quads."""
    if N.dev_env("EBFI_NO_C16", "0") == "1":          # (development switch: same-box A/B of the two forms)
        return False
    if book is None or W % 4 != 0 or any(sa.tr16_ptr() is None or sb.tr16_ptr() is None or sc.tr16_ptr() is None for sa, sb, sc in sites):
        return False
    return all(book.index.get(k) in book.calibrated for k in _slot_keys(sites))


def _conv16(lib, st, x, site, bias, out, B, cin_g, H, W, cout, groups, slope, out16, slot16):
    """Split-precision forward convolution + LeakyReLU whose output also leaves as an fp16 image."""
    rc = lib.ebfi_conv2d_packed_x3_c16(N.ptr(x), site.fwd_ptr(), site.fwd_bytes, N.ptr(bias), N.ptr(out), B, cin_g, H, W, cout, 3, 1,
                                       groups, ACT, slope, N.ptr(None), N.ptr(None), 0, 0.0, N.ptr(out16),
                                       slot16 if out16 is not None else N.ptr(None), 0, st)
    N.check(rc, "ebfi_conv2d_packed_x3_c16")


def _fconv16(lib, st, x16, site, B, cin_g, H, W, cout, groups, slope, book, out=None, out16=None, slot16=None):
    """fp16-operand FORWARD convolution + bias + LeakyReLU from the image of its input (scale slot (site, 'x')) on the site's
    fp16 forward weight image; the result as fp32 (`out`) and / or as the next layer's input image (`out16`, `slot16`)."""
    rc = lib.ebfi_conv2d_packed_f16_c16(N.ptr(x16), 1, site.fwd16_ptr(), site.fwd16_bytes, N.ptr(site.bias()), N.ptr(out), B, cin_g, H, W,
                                        cout, 3, 1, groups, ACT, slope, N.ptr(None), N.ptr(None), 0, 0.0,
                                        book.ptr(book.slot((site.key, "x"))), site.w_slot_ptr(), N.ptr(out16),
                                        slot16 if out16 is not None else N.ptr(None), 0, 0, st)
    N.check(rc, "ebfi_conv2d_packed_f16_c16")


def _wgrad16(lib, st, x16, g16, B, cin_g, H, W, cout, groups, ws_cache, book, site):
    gw = torch.empty((cout, cin_g, 3, 3), dtype=torch.float32, device=x16.device)
    gb = torch.empty(cout, dtype=torch.float32, device=x16.device)
    key = (cin_g, cout)
    if key not in ws_cache:
        need = int(lib.ebfi_conv2d_backward_weight_workspace(B, cin_g, H, W, cout, 3, 1, 1, N.EBFI_F32))
        ws_cache[key] = (torch.empty(max(need, 4), dtype=torch.uint8, device=x16.device), need)
    ws, need = ws_cache[key]
    rc = lib.ebfi_conv2d_backward_weight_f16c(N.ptr(x16), N.ptr(g16), 0, N.ptr(gw), N.ptr(gb), B, cin_g, H, W, cout, groups,
                                              book.ptr(book.slot((site.key, "x"))), book.ptr(book.slot((site.key, "g"))),
                                              N.ptr(ws), need, st)
    N.check(rc, "ebfi_conv2d_backward_weight_f16c")
    return gw, gb


def _wgrad16_batch(lib, st, items, B, H, W, ws_cache, book):
    """The weight / bias gradients of several layers over the same pixels as ONE launch (+ one reduction):
    items = [(x16, g16, site, cin_g, cout, groups)], every layer two 64 x 64 blocks.  Returns [(gw, gb)]."""
    import ctypes
    n = len(items)
    dev = items[0][0].device
    gws = [torch.empty((cout, cin_g, 3, 3), dtype=torch.float32, device=dev) for _, _, _, cin_g, cout, _ in items]
    gbs = [torch.empty(cout, dtype=torch.float32, device=dev) for _, _, _, _, cout, _ in items]
    ints = lambda vals: (ctypes.c_int * n)(*vals)
    ptrs = lambda vals: (ctypes.c_void_p * n)(*vals)
    cin, cout, groups = ints([it[3] for it in items]), ints([it[4] for it in items]), ints([it[5] for it in items])
    key = ("batch",) + tuple((it[3], it[4]) for it in items)
    if key not in ws_cache:
        need = int(lib.ebfi_conv2d_backward_weight_f16c_batch_workspace(n, cin, cout))
        ws_cache[key] = (torch.empty(max(need, 4), dtype=torch.uint8, device=dev), need)
    ws, need = ws_cache[key]
    slot = lambda site, role: book.ptr(book.slot((site.key, role))).value
    rc = lib.ebfi_conv2d_backward_weight_f16c_batch(
        n, ptrs([it[0].data_ptr() for it in items]), ptrs([it[1].data_ptr() for it in items]), ptrs([t.data_ptr() for t in gws]),
        ptrs([t.data_ptr() for t in gbs]), cin, cout, groups, ptrs([slot(it[2], "x") for it in items]),
        ptrs([slot(it[2], "g") for it in items]), B, H, W, N.ptr(ws), need, st)
    N.check(rc, "ebfi_conv2d_backward_weight_f16c_batch")
    return list(zip(gws, gbs))


def _dgrad16(lib, st, g16, site, B, cin_g, H, W, cout, groups, slope, book, out=None, out16=None, slot16=None, addend=None, mask=None,
             mask16=None):
    """Data gradient of `site`'s layer from the IMAGE of its pre-activation gradient; the result as fp32 (`out`) and / or as
    the image of the next pre-activation gradient (`out16`, scale slot `slot16`)."""
    m = mask16 if mask16 is not None else mask            # (mask16: the c16 image of the mask tensor -- only its signs are read)
    rc = lib.ebfi_conv2d_packed_f16_c16(N.ptr(g16), 1, site.tr16_ptr(), site.tr16_bytes, N.ptr(None), N.ptr(out), B, cin_g, H, W, cout,
                                        3, 1, groups, 0, slope, N.ptr(addend), N.ptr(m), ACT if m is not None else 0,
                                        slope if m is not None else 0.0, book.ptr(book.slot((site.key, "g"))), site.w_slot_ptr(),
                                        N.ptr(out16), slot16 if out16 is not None else N.ptr(None), 0, 1 if mask16 is not None else 0, st)
    N.check(rc, "ebfi_conv2d_packed_f16_c16")


class ResidualControlFn(Function):
    """apply(data, s_ex [step,B,C], s_t [step,B,C], sites, slope, *params) -> x_step.
    params per round: W3a, b3a, W4a, b4a, W3b, b3b, W4b, b4b, W5, b5 (inputs only so that autograd routes their gradients)."""

    @staticmethod
    def forward(ctx, data, s_ex, s_t, sites_keep, slope, *params):
        x = data.contiguous()
        B, C, H, W = (int(v) for v in x.shape)
        HW = H * W
        s_ex, s_t = s_ex.contiguous(), s_t.contiguous()
        lib = N.lib()
        saved = []
        # inference (grad mode off at the call): nothing is kept -- at B=8 720x1280 the 12 rounds' intermediates are 40 GB
        keep = bool(sites_keep[1])
        sites = sites_keep[0]
        from . import c16, f16scale
        book = f16scale.active_book()
        images = keep and images_usable(book, sites, W)
        ctx.images, ctx.tail = images, False
        if images:
            with torch.cuda.device_of(x):
                st = N.stream_ptr(x.device)
                new = lambda ch: torch.empty((B, ch, H, W), dtype=x.dtype, device=x.device)
                img = lambda ch: c16.empty(B, ch, H, W, x.device)
                sp = lambda site, role: book.ptr(book.slot((site.key, role)))
                x16 = c16.to_c16(x, sp(sites[0][0], "x"))
                f16fwd = f16scale.forward_level(book) >= 2 and all(t.fwd16_ptr() is not None for r in sites for t in r)
                tail = not f16fwd and _tail_in_epilogue(book, sites)
                ctx.tail = tail
                s_cat = torch.cat([s_ex, s_t], 2).contiguous() if tail else None        # [step, B, 2C]: the epilogue's scales
                for i, (sa, sb, sc) in enumerate(sites):
                    nxt = sites[i + 1][0] if i + 1 < len(sites) else None
                    ya, xn = new(2 * C), new(C)
                    ya16, c16i = img(2 * C), img(2 * C)
                    xn16 = img(C) if nxt is not None else None
                    if tail:
                        # the grouped convolution's epilogue forms c = cat(s_ex * a0 + x, s_t * a1 + x) itself and leaves the
                        # images of c (Conv5's weight gradient) and of a (the fused backward stage): no fp32 `a`, no stage launch
                        c, a16 = new(2 * C), img(2 * C)
                        _conv16(lib, st, x, sa, sa.bias(), ya, B, C, H, W, 2 * C, 1, slope, ya16, sp(sb, "x"))
                        rc = lib.ebfi_conv2d_packed_x3_rc(N.ptr(ya), sb.fwd_ptr(), sb.fwd_bytes, N.ptr(sb.bias()), N.ptr(c), B, C, H, W,
                                                          2 * C, 2, slope, N.ptr(s_cat[i]), N.ptr(x), C, N.ptr(a16), sp(sb, "a"),
                                                          N.ptr(c16i), sp(sc, "x"), st)
                        N.check(rc, "ebfi_conv2d_packed_x3_rc")
                        _conv16(lib, st, c, sc, sc.bias(), xn, B, 2 * C, H, W, C, 1, slope, xn16, sp(nxt, "x") if nxt is not None else None)
                        saved += [a16, x16, ya16, c16i]
                        x, x16 = xn, xn16
                        continue
                    a = new(2 * C)
                    if f16fwd:
                        # fp16-operand forward: every convolution reads the image its producer wrote (the same images the
                        # weight gradients read later) -- one matrix-core product per tap instead of three, half the input bytes
                        _fconv16(lib, st, x16, sa, B, C, H, W, 2 * C, 1, slope, book, out=ya, out16=ya16, slot16=sp(sb, "x"))
                        _fconv16(lib, st, ya16, sb, B, C, H, W, 2 * C, 2, slope, book, out=a)
                        c = None
                    else:
                        c = new(2 * C)
                        _conv16(lib, st, x, sa, sa.bias(), ya, B, C, H, W, 2 * C, 1, slope, ya16, sp(sb, "x"))
                        _conv(lib, st, ya, sb.fwd_ptr(), sb.fwd_bytes, sb.bias(), a, B, C, H, W, 2 * C, 2, ACT, slope)
                        # the epilogue-tail form above needs the slot of the image of `a`, which only the layer-wise form used to
                        # measure: slots restored from an earlier state, or a first pass run with EBFI_NO_RC_EPILOGUE=1, left this
                        # path on the separate stage for good (round-5 advisory).  Measured here while the fp32 `a` still exists
                        # (eager passes only: a first use cannot be calibrated inside a graph capture).
                        if book.index.get((sb.key, "a")) not in book.calibrated and not torch.cuda.is_current_stream_capturing() \
                                and N.dev_env("EBFI_NO_RC_EPILOGUE", "0") != "1":
                            book.operand((sb.key, "a"), a)
                    rc = lib.ebfi_scale_residual_cat_forward_c16(N.ptr(a), N.ptr(s_ex[i]), N._vp(a.data_ptr() + 4 * C * HW), N.ptr(s_t[i]),
                                                                 N.ptr(x), N.ptr(c), N.ptr(c16i), sp(sc, "x"), B, C, H, W, 2 * C * HW, st)
                    N.check(rc, "ebfi_scale_residual_cat_forward_c16")
                    if f16fwd:
                        _fconv16(lib, st, c16i, sc, B, 2 * C, H, W, C, 1, slope, book, out=xn, out16=xn16,
                                 slot16=sp(nxt, "x") if nxt is not None else None)
                    else:
                        _conv16(lib, st, c, sc, sc.bias(), xn, B, 2 * C, H, W, C, 1, slope, xn16, sp(nxt, "x") if nxt is not None else None)
                    # (neither x, ya nor c is needed again in fp32: their images feed the weight gradients and give the
                    # data gradients the signs of the LeakyReLU derivatives)
                    saved += [a, x16, ya16, c16i]
                    x, x16 = xn, xn16
            ctx.sites, ctx.slope, ctx.dims = sites, slope, (B, C, H, W)
            ctx.save_for_backward(s_ex, s_t, x, *saved)
            return x
        with torch.cuda.device_of(x):
            st = N.stream_ptr(x.device)
            new = lambda ch: torch.empty((B, ch, H, W), dtype=x.dtype, device=x.device)
            # inference: the round's tail in the grouped convolution's epilogue (scale + residual only, no images): one launch and
            # one [B, 2C, H, W] tensor less per round -- 0.39 ms and 3.8 GB of traffic each at B=8 720x1280
            tail = not keep and W % 4 == 0 and C % 64 == 0 and x.data_ptr() % 16 == 0 and N.dev_env("EBFI_NO_RC_EPILOGUE", "0") != "1"
            s_cat = torch.cat([s_ex, s_t], 2).contiguous() if tail else None
            for i, (sa, sb, sc) in enumerate(sites):
                if tail:
                    ya, c, xn = new(2 * C), new(2 * C), new(C)
                    _conv(lib, st, x, sa.fwd_ptr(), sa.fwd_bytes, sa.bias(), ya, B, C, H, W, 2 * C, 1, ACT, slope)
                    rc = lib.ebfi_conv2d_packed_x3_rc(N.ptr(ya), sb.fwd_ptr(), sb.fwd_bytes, N.ptr(sb.bias()), N.ptr(c), B, C, H, W,
                                                      2 * C, 2, slope, N.ptr(s_cat[i]), N.ptr(x), C, N.ptr(None), N.ptr(None),
                                                      N.ptr(None), N.ptr(None), st)
                    N.check(rc, "ebfi_conv2d_packed_x3_rc (inference)")
                    _conv(lib, st, c, sc.fwd_ptr(), sc.fwd_bytes, sc.bias(), xn, B, 2 * C, H, W, C, 1, ACT, slope)
                    x = xn
                    continue
                ya, a, c, xn = new(2 * C), new(2 * C), new(2 * C), new(C)
                _conv(lib, st, x, sa.fwd_ptr(), sa.fwd_bytes, sa.bias(), ya, B, C, H, W, 2 * C, 1, ACT, slope)
                _conv(lib, st, ya, sb.fwd_ptr(), sb.fwd_bytes, sb.bias(), a, B, C, H, W, 2 * C, 2, ACT, slope)
                if keep and book is not None:
                    book.operand((sb.key, "a"), a)          # (scale of the image of `a` the epilogue form writes: measured here)
                rc = lib.ebfi_scale_residual_cat_forward_ex(N.ptr(a), N.ptr(s_ex[i]), N._vp(a.data_ptr() + 4 * C * HW), N.ptr(s_t[i]),
                                                            N.ptr(x), N.ptr(c), B, C, HW, 2 * C * HW, st)
                N.check(rc, "ebfi_scale_residual_cat_forward_ex")
                _conv(lib, st, c, sc.fwd_ptr(), sc.fwd_bytes, sc.bias(), xn, B, 2 * C, H, W, C, 1, ACT, slope)
                if keep:
                    saved += [x, ya, a, c]
                x = xn
        ctx.sites, ctx.slope, ctx.dims = sites, slope, (B, C, H, W)
        ctx.save_for_backward(s_ex, s_t, x, *saved)
        return x

    @staticmethod
    def backward(ctx, gout):
        s_ex, s_t, xlast, *saved = ctx.saved_tensors
        sites, slope, (B, C, H, W) = ctx.sites, ctx.slope, ctx.dims
        HW = H * W
        lib = N.lib()
        nstep = len(sites)
        from . import f16scale
        book = f16scale.active_book()
        gs_ex, gs_t = torch.empty_like(s_ex), torch.empty_like(s_t)
        pgrads = [None] * (10 * nstep)
        ws_cache = {}
        if ctx.images:
            from . import c16
            if book is None:
                raise RuntimeError("ResidualControl ran its forward with fp16 operand images: backward() must run inside the same "
                                   "scale-book context (Engine._fwd_bwd)")
            with torch.cuda.device_of(gout):
                st = N.stream_ptr(gout.device)
                dev = gout.device
                new = lambda ch: torch.empty((B, ch, H, W), dtype=gout.dtype, device=dev)
                img = lambda ch: c16.empty(B, ch, H, W, dev)
                sp = lambda site, role: book.ptr(book.slot((site.key, role)))
                S = int(lib.ebfi_scale_residual_cat_backward_slices())
                parts = torch.empty((2, nstep, S, B, C), dtype=torch.float32, device=dev)
                # image of the last round's pre-activation gradient (later rounds get theirs from the data gradient's epilogue)
                g5 = c16.to_c16(gout.contiguous(), sp(sites[-1][2], "g"), xlast, slope)
                # (every layer of a round is two 64 x 64 blocks exactly when C == 64; enough 4 x 32 pixel tiles for the 40 splits)
                batch = C == 64 and B * ((H + 3) // 4) * ((W + 31) // 32) >= 40 and N.dev_env("EBFI_NO_WGRAD_BATCH", "0") != "1"
                gdata = None
                for i in range(nstep - 1, -1, -1):
                    sa, sb, sc = sites[i]
                    a, x16, ya16, c16i = saved[4 * i:4 * i + 4]
                    g5_img = g5
                    if not batch:
                        gw5, gb5 = _wgrad16(lib, st, c16i, g5, B, 2 * C, H, W, C, 1, ws_cache, book, sc)
                    gc = new(2 * C)
                    _dgrad16(lib, st, g5, sc, B, C, H, W, 2 * C, 1, 0.0, book, out=gc)
                    gb16, gxres = img(2 * C), new(C)
                    if ctx.tail:        # `a` is the image the forward epilogue wrote
                        rc = lib.ebfi_scale_residual_cat_backward_c16a(
                            N.ptr(gc), N.ptr(a), sp(sb, "a"), N.ptr(s_ex[i]), N.ptr(s_t[i]), N.ptr(gb16), sp(sb, "g"), N.ptr(gxres),
                            N.ptr(parts[0, i]), N.ptr(parts[1, i]), B, C, H, W, slope, st)
                    else:
                        rc = lib.ebfi_scale_residual_cat_backward_c16(
                            N.ptr(gc), N.ptr(a), N.ptr(s_ex[i]), N._vp(a.data_ptr() + 4 * C * HW), N.ptr(s_t[i]), N.ptr(gb16), sp(sb, "g"),
                            N.ptr(gxres), N.ptr(parts[0, i]), N.ptr(parts[1, i]), B, C, H, W, 2 * C * HW, slope, st)
                    N.check(rc, "ebfi_scale_residual_cat_backward_c16")
                    if not batch:
                        gwb, gbb = _wgrad16(lib, st, ya16, gb16, B, C, H, W, 2 * C, 2, ws_cache, book, sb)
                    ga16 = img(2 * C)
                    _dgrad16(lib, st, gb16, sb, B, C, H, W, 2 * C, 2, slope, book, out16=ga16, slot16=sp(sa, "g"), mask16=ya16)
                    if batch:       # the round's three weight gradients as one launch, once the last of their operands exists
                        (gwa, gba), (gwb, gbb), (gw5, gb5) = _wgrad16_batch(
                            lib, st, [(x16, ga16, sa, C, 2 * C, 1), (ya16, gb16, sb, C, 2 * C, 2), (c16i, g5_img, sc, 2 * C, C, 1)],
                            B, H, W, ws_cache, book)
                    else:
                        gwa, gba = _wgrad16(lib, st, x16, ga16, B, C, H, W, 2 * C, 1, ws_cache, book, sa)
                    if i > 0:       # leaves as the image of the previous round's Conv5 pre-activation gradient
                        g5 = img(C)
                        _dgrad16(lib, st, ga16, sa, B, 2 * C, H, W, C, 1, slope, book, out16=g5, slot16=sp(sites[i - 1][2], "g"),
                                 addend=gxres, mask16=x16)
                    else:
                        gdata = new(C)
                        _dgrad16(lib, st, ga16, sa, B, 2 * C, H, W, C, 1, slope, book, out=gdata, addend=gxres)
                    pgrads[10 * i:10 * i + 10] = [gwa[:C], gba[:C], gwa[C:], gba[C:], gwb[:C], gbb[:C], gwb[C:], gbb[C:], gw5, gb5]
                sums = parts.sum(2)                    # per-slice partials of the scale gradients, in slice order
                gs_ex, gs_t = sums[0], sums[1]
            need = ctx.needs_input_grad
            return (gdata if need[0] else None, gs_ex if need[1] else None, gs_t if need[2] else None, None, None) + tuple(pgrads)
        with torch.cuda.device_of(gout):
            st = N.stream_ptr(gout.device)
            new = lambda ch: torch.empty((B, ch, H, W), dtype=gout.dtype, device=gout.device)
            # gradient of the last round's pre-activation (later rounds get theirs from the epilogue below)
            gpre5 = torch.where(xlast > 0, gout, gout * slope)
            gdata = None
            for i in range(nstep - 1, -1, -1):
                sa, sb, sc = sites[i]
                x, ya, a, c = saved[4 * i:4 * i + 4]
                gw5, gb5 = _wgrad(lib, st, c, gpre5, B, 2 * C, H, W, C, 1, ws_cache, book, sc)
                gc = new(2 * C)
                _dgrad(lib, st, gpre5, sc, gc, B, C, H, W, 2 * C, 1, 0.0, book=book)
                gpre_b, gxres = new(2 * C), new(C)
                rc = lib.ebfi_scale_residual_cat_backward_ex(
                    N.ptr(gc), N.ptr(a), N.ptr(s_ex[i]), N._vp(a.data_ptr() + 4 * C * HW), N.ptr(s_t[i]), N.ptr(gpre_b),
                    N._vp(gpre_b.data_ptr() + 4 * C * HW), N.ptr(gxres), N.ptr(gs_ex[i]), N.ptr(gs_t[i]), B, C, HW, 2 * C * HW,
                    2 * C * HW, 1, slope, st)
                N.check(rc, "ebfi_scale_residual_cat_backward_ex")
                gwb, gbb = _wgrad(lib, st, ya, gpre_b, B, C, H, W, 2 * C, 2, ws_cache, book, sb)
                gpre_a = new(2 * C)
                _dgrad(lib, st, gpre_b, sb, gpre_a, B, C, H, W, 2 * C, 2, slope, None, ya, book)
                gwa, gba = _wgrad(lib, st, x, gpre_a, B, C, H, W, 2 * C, 1, ws_cache, book, sa)
                gx = new(C)
                # grad wrt x of this round = data gradient of the merged first layers + the residual path; for i > 0 it
                # leaves as the pre-activation gradient of the previous round's Conv5 (x is that layer's LeakyReLU output)
                _dgrad(lib, st, gpre_a, sa, gx, B, 2 * C, H, W, C, 1, slope, gxres, x if i > 0 else None, book)
                if i > 0:
                    gpre5 = gx
                else:
                    gdata = gx
                pgrads[10 * i:10 * i + 10] = [gwa[:C], gba[:C], gwa[C:], gba[C:], gwb[:C], gbb[:C], gwb[C:], gbb[C:], gw5, gb5]
        need = ctx.needs_input_grad
        return (gdata if need[0] else None, gs_ex if need[1] else None, gs_t if need[2] else None, None, None) + tuple(pgrads)


def residual_control(rc, data, Ex, T):
    """Fused ResidualControl.forward; the caller checked `usable` and found the sites."""
    sites = sites_of(rc)
    if sites is None:
        return None
    slope = float(rc.Conv5[0][0].activation.negative_slope)
    # per-round channel scales Conv1[i](Ex), Conv2[i](T) for all rounds at once: [step, B, C] -- one native launch per bank
    # (csrc/fuse.hip scalar_conv_*; before round 6: stack + einsum + add + leaky_relu through torch, ~30 tiny launches per step)
    def scales(bank_modules, v):
        from . import fused
        return fused.scalar_conv_bank(v, [m[0].conv2d.weight for m in bank_modules], [m[0].conv2d.bias for m in bank_modules],
                                      float(bank_modules[0][0].activation.negative_slope))
    s_ex, s_t = scales(rc.Conv1, Ex), scales(rc.Conv2, T)
    params = []
    for i in range(rc.step):
        for m in (rc.Conv3[i][0], rc.Conv4[i][0], rc.Conv3[i][1], rc.Conv4[i][1], rc.Conv5[i][0]):
            params += [m.conv2d.weight, m.conv2d.bias]
    keep = torch.is_grad_enabled()      # (inside Function.forward grad mode is always off: decide here)
    return ResidualControlFn.apply(data, s_ex, s_t, (sites, keep), slope, *params)
