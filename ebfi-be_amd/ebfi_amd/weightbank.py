"""Weight bank: the MFMA operand images of EVERY convolution weight of a model, refreshed by ONE launch per step.

The split-precision conv kernels (csrc/conv2d.hip) read their weights as two bf16 images (hi = bf16(w), lo = bf16(w - hi))
in the layouts [tap][Cout][Cin16] (forward) and [tap][Cin][Cout16] with flipped taps (data gradient).  Round 1 rebuilt
those images inside every conv call: 186 packing launches of ~5 us per training step for weights that change once per
optimiser step.  Here they are produced for all layers at once by `ebfi_pack_table_bf16`, a gather driven by an index table
built once on the host: entry e names the source element (in the flat parameter buffer) of packed element e.  Because the
table is arbitrary, it also carries the layouts that used to need extra launches of their own:

  * the depth-2 Conv3d / ConvTranspose3d folds of the detail branch (ebfi_amd.fold3d) -- table = fold(ids),
  * biases repeated over the folded channels (one `ebfi_gather_sum` launch for all of them),
  * several convolutions reading the same input as ONE convolution over concatenated output channels
    (`register_concat`: ResidualControl's Conv3[i][0] | Conv4[i][0], reference model_singleframe.py:127-131).

A bank reads the parameters from one flat fp32 buffer: `FlatAdam.flat` in training (no copy), its own concatenation
for inference.  Ops find their images through `lookup(param, kind)` while a bank is active (`with bank.active():`); without
an active bank, or in the fp32 / bf16 conv modes, every op packs per call exactly as before.
"""
import contextlib

import numpy as np
import torch

from . import _native as N

_ACTIVE = None
LO_FLAG = 1 << 30


def active_bank():
    return _ACTIVE


def lookup(param, kind="id"):
    """Site of `param` under the active bank, or None."""
    if _ACTIVE is None or param is None:
        return None
    return _ACTIVE.sites.get((param.data_ptr(), kind))


class Site:
    """Packed images of one (possibly folded / concatenated) conv weight [M = Cout, K = Cin, ks, ks]."""
    __slots__ = ("bank", "kind", "M", "K", "ks", "groups", "fwd_off", "tr_off", "fwd_bytes", "tr_bytes", "bias_off", "has_bias",
                 "w_inv", "w_R", "b_inv", "b_R", "w_shapes", "b_shapes", "key", "tr16_off", "tr16_bytes", "w_slot", "weights",
                 "fwd16_off", "fwd16_bytes", "fwd16_tag")

    def fwd_ptr(self):
        return N._vp(self.bank.packed.data_ptr() + self.fwd_off)

    def tr_ptr(self):
        return N._vp(self.bank.packed.data_ptr() + self.tr_off)

    def tr16_ptr(self):
        """fp16 transposed (data-gradient) image, scaled by the site's weight slot (None while the bank has no scale book)."""
        if self.bank.packed16 is None or self.tr16_bytes == 0:
            return None
        return N._vp(self.bank.packed16.data_ptr() + 2 * self.tr16_off)

    def fwd16_ptr(self):
        """fp16 forward image [tap][Cout][Cin16] scaled by the same weight slot (the fp16-operand FORWARD of a training step)."""
        if self.bank.packed16 is None or self.fwd16_bytes == 0:
            return None
        return N._vp(self.bank.packed16.data_ptr() + 2 * self.fwd16_off)

    def w_slot_ptr(self):
        return self.bank.book.ptr(self.w_slot)

    def bias(self):
        """fp32 [M] view of the (folded) bias inside the bank, or None."""
        return self.bank.bias_buf[self.bias_off:self.bias_off + self.M] if self.has_bias else None


def _ids_like(p, flat_off):
    """float64 tensor shaped like p holding (flat index + 1) of every element: pushed through the (0/1 linear) fold
    functions it yields, per folded element, the source element + 1, or 0 for a structural zero."""
    return (torch.arange(p.numel(), dtype=torch.float64) + (flat_off + 1)).view(p.shape)


def _adjoint_table(src_ids, n_src, first):
    """For a fold given as `src_ids` (flat index of the source of every folded element, -1 = none), the table that routes a
    gradient of the folded tensor back: [n_src, R] positions in the folded tensor per source element (-1 padded),
    source elements counted from `first`."""
    fwd = src_ids.reshape(-1).astype(np.int64)
    dest = np.nonzero(fwd >= 0)[0]
    order = np.argsort(fwd[dest], kind="stable")
    srcs, dest = fwd[dest][order] - first, dest[order]
    counts = np.bincount(srcs, minlength=n_src)
    R = int(max(1, counts.max()))
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    inv = np.full((n_src, R), -1, dtype=np.int32)
    inv[srcs, np.arange(len(srcs)) - starts[srcs]] = dest
    return inv, R


class WeightBank:
    def __init__(self, params, flat=None, inference=False, fwd16=None):
        """`params`: the tensors the sites may draw from.  `flat`: one fp32 buffer the params are views of, in this order
        (FlatAdam.flat); None = the bank keeps its own concatenation and re-copies it when a parameter's version changes.
        `inference`: modules may register forward-only sites that a training step would not use (the fused KernelConv -> FAC
        layout of Modification) -- they would only lengthen the per-step pack launch of a training bank."""
        self.inference = bool(inference)
        # which sites also get an fp16 FORWARD image (Engine(forward_f16=...), f16scale.FORWARD_LEVELS): the convolutions whose
        # module carries a tag `_ebfi_fwd16` ("filters": Modification.KernelConv) up to this level; "all": every 3x3 site
        self.fwd16 = fwd16
        self.params = list(params)
        self.owns_flat = flat is None
        self._offsets, off = {}, 0
        for p in self.params:
            self._offsets[p.data_ptr()] = (off, p)
            off += p.numel()
        self.numel = off
        self.device = self.params[0].device
        self.flat = torch.cat([p.detach().reshape(-1).float() for p in self.params]) if flat is None else flat
        if self.flat.numel() != self.numel:
            raise ValueError("flat buffer has %d elements, parameters %d" % (self.flat.numel(), self.numel))
        if self.numel >= LO_FLAG:
            raise ValueError("more than 2^30 parameters")
        self.sites = {}
        self._tables, self._n_packed = [], 0
        self._bias_tables, self._n_bias = [], 0
        self.packed = self.table = self.bias_buf = self.bias_table = None
        self._stamp = None
        # fp16 data-gradient images (ebfi_amd.f16scale / csrc/conv2d_f16.inc.hpp): built once a scale book is attached
        self.book = None
        self._tables16, self._n16, self._segs = [], 0, []
        self.packed16 = self.table16 = self.block_slot = None

    # ------------------------------------------------------------------ registration (host side, once)
    def _offset(self, p):
        ent = self._offsets.get(p.data_ptr())
        if ent is None or ent[1].shape != p.shape:
            raise KeyError("parameter is not part of this bank")
        return ent[0]

    def register(self, weights, biases=None, kind="id", fold_w=None, fold_b=None, groups=1, need_tr=True, fwd16=False):
        """`weights`: one parameter, or a list concatenated along the (folded) output-channel axis.  `fold_w` / `fold_b`:
        0/1 linear maps from the parameter's shape to [M, K, ks, ks] / [M] (None = identity).  Keyed by the FIRST weight.
        `groups` > 1: the M rows form that many groups, each convolving its own K input channels (a grouped convolution:
        the data-gradient image is then [tap][(group, ci)][co within the group]).  `need_tr` = False: forward images only
        (a site no data gradient will ever be taken through).  `fwd16`: also keep an fp16 forward image (with the scale book)."""
        weights = list(weights) if isinstance(weights, (list, tuple)) else [weights]
        biases = list(biases) if isinstance(biases, (list, tuple)) else ([biases] if biases is not None else [])
        if any(b is None for b in biases):
            if not all(b is None for b in biases):
                raise ValueError("a concatenated site needs a bias on every part or on none")
            biases = []
        key = (weights[0].data_ptr(), kind)
        if key in self.sites:
            return self.sites[key]
        ident = lambda t: t
        fw, fb = fold_w or ident, fold_b or ident
        ids = torch.cat([fw(_ids_like(w, self._offset(w))) for w in weights], dim=0).round().to(torch.int64) - 1   # [M,K,ks,ks]
        M, K, ks, ks2 = ids.shape
        if ks != ks2 or ks not in (1, 3):
            raise ValueError("weight bank holds 1x1 / 3x3 kernels, got %dx%d" % (ks, ks2))
        KK, K16, M16 = ks * ks, (K + 15) // 16 * 16, (M + 15) // 16 * 16
        pad = lambda t, n: torch.cat([t, t.new_full(t.shape[:-1] + (n - t.shape[-1],), -1)], dim=-1)
        fwd = pad(ids.permute(2, 3, 0, 1).reshape(KK, M, K), K16).reshape(-1)                    # [tap][co][ci16]
        if M % groups != 0:
            raise ValueError("%d output channels do not split into %d groups" % (M, groups))
        Mg = M // groups
        tr = pad(ids.flip((2, 3)).view(groups, Mg, K, ks, ks).permute(3, 4, 0, 2, 1).reshape(KK, groups * K, Mg),
                 (Mg + 15) // 16 * 16).reshape(-1)                                               # [tap][(g, ci)][co16], taps flipped
        s = Site()
        s.bank, s.kind, s.M, s.K, s.ks, s.groups = self, kind, int(M), int(K), int(ks), int(groups)
        s.key, s.weights = key, weights
        s.fwd16_tag = fwd16 if isinstance(fwd16, str) else None
        s.tr_off, s.tr_bytes, s.tr16_off, s.tr16_bytes, s.w_slot, s.fwd16_off, s.fwd16_bytes = 0, 0, 0, 0, -1, 0, 0
        # fp16 images (scaled by the site's weight slot; images start on 256-element bounds): the transposed one for the fp16 data
        # gradient of a training bank, a forward one where asked for -- an inference bank keeps forward images only (round 6: the
        # fused KernelConv -> FAC kernel on fp16 operands)
        images16 = []
        if ks == 3 and need_tr and not self.inference:
            images16.append(("tr16", tr))
        if ks == 3 and (fwd16 or (self.fwd16 == "all" and need_tr)):
            images16.append(("fwd16", fwd))
        if images16:
            for name, img in images16:
                setattr(s, name + "_off", self._n16)
                setattr(s, name + "_bytes", 2 * img.numel())
                padn = (-img.numel()) % 256
                self._tables16.append(torch.cat([img.to(torch.int32), torch.full((padn,), -1, dtype=torch.int32)]))
                self._segs.append((self._n16, s))
                self._n16 += img.numel() + padn
        for name, img in (("fwd", fwd), ("tr", tr)) if need_tr else (("fwd", fwd),):
            lo = torch.where(img >= 0, img | LO_FLAG, img)
            setattr(s, name + "_off", 2 * self._n_packed)
            setattr(s, name + "_bytes", 4 * img.numel())
            self._tables += [img.to(torch.int32), lo.to(torch.int32)]
            self._n_packed += 2 * img.numel()
        # gradient routes back to the parameters (per source parameter: positions in the folded gradient)
        idn = ids.numpy()
        s.w_inv, s.w_R, s.w_shapes = [], [], [tuple(w.shape) for w in weights]
        for w in weights:
            o = self._offset(w)
            inv, R = _adjoint_table(np.where((idn >= o) & (idn < o + w.numel()), idn, -1), w.numel(), o)
            s.w_inv.append(torch.from_numpy(inv).to(self.device))
            s.w_R.append(R)
        s.has_bias = len(biases) > 0
        s.bias_off, s.b_inv, s.b_R, s.b_shapes = self._n_bias, [], [], [tuple(b.shape) for b in biases]
        if s.has_bias:
            bids = torch.cat([fb(_ids_like(b, self._offset(b))) for b in biases], dim=0).round().to(torch.int64) - 1
            if bids.numel() != M:
                raise ValueError("folded bias has %d entries for %d output channels" % (bids.numel(), M))
            self._bias_tables.append(bids.to(torch.int32))
            self._n_bias += int(M)
            bn = bids.numpy()
            for b in biases:
                o = self._offset(b)
                inv, R = _adjoint_table(np.where((bn >= o) & (bn < o + b.numel()), bn, -1), b.numel(), o)
                s.b_inv.append(torch.from_numpy(inv).to(self.device))
                s.b_R.append(R)
        self.sites[key] = s
        self.packed = None            # (re)built at the next refresh
        return s

    def _finalize(self):
        self.table = torch.cat(self._tables).to(self.device) if self._tables else None
        self.packed = torch.empty(max(self._n_packed, 8), dtype=torch.bfloat16, device=self.device)
        self.bias_table = torch.cat(self._bias_tables).to(self.device) if self._bias_tables else None
        self.bias_buf = torch.empty(max(self._n_bias, 1), dtype=torch.float32, device=self.device)
        if self.book is not None and self._tables16:
            self.table16 = torch.cat(self._tables16).to(self.device)
            self.packed16 = torch.empty(self._n16, dtype=torch.float16, device=self.device)
            counts = []
            for (off, site), nxt in zip(self._segs, [o for o, _ in self._segs[1:]] + [self._n16]):
                site.w_slot = self.book.slot((site.key, "w"))
                counts.append((nxt - off) // 256)
            self.block_slot = torch.repeat_interleave(torch.tensor([site.w_slot for _, site in self._segs], dtype=torch.int32),
                                                      torch.tensor(counts)).to(self.device)

    def attach_scale_book(self, book):
        """Enable the fp16 data-gradient images: every 3x3 site gets a transposed fp16 image scaled by its own slot of `book`."""
        self.book = book
        self.packed = None            # (re)built, with the fp16 tables, at the next refresh

    # ------------------------------------------------------------------ per step
    def _current_stamp(self):
        if self.owns_flat:
            return sum(p._version for p in self.params)
        return self.flat._version

    def refresh(self):
        """Re-derive every image from the current parameter values: one pack launch (+ one gather for the folded biases;
        + one concatenation when the bank keeps its own flat copy)."""
        if self.packed is None:
            self._finalize()
        if self.owns_flat:
            torch.cat([p.detach().reshape(-1) for p in self.params], out=self.flat)
        lib = N.lib()
        with torch.cuda.device(self.device):
            st = N.stream_ptr(self.device)
            if self.table is not None:
                N.check(lib.ebfi_pack_table_bf16(N.ptr(self.flat), N.ptr(self.table), self.table.numel(), N.ptr(self.packed), st),
                        "ebfi_pack_table_bf16")
            if self.bias_table is not None:
                N.check(lib.ebfi_gather_sum(N.ptr(self.flat), N.ptr(self.bias_table), N.ptr(self.bias_buf), self._n_bias, 1, st),
                        "ebfi_gather_sum")
            if self.packed16 is not None:
                for _, site in self._segs:       # first refresh: the weight scales from the weights themselves (later: delayed)
                    if self.inference:           # (no optimiser step follows that would move a delayed scale along: exact every time)
                        self.book.calibrated.discard(site.w_slot)
                    self.book.calibrate(site.w_slot, *site.weights)
                N.check(lib.ebfi_pack_table_f16(N.ptr(self.flat), N.ptr(self.table16), self.table16.numel(), N.ptr(self.packed16),
                                                N.ptr(self.block_slot), N.ptr(self.book.slots), st), "ebfi_pack_table_f16")
        self._stamp = self._current_stamp()

    def ensure_fresh(self):
        if self.packed is None or self._stamp != self._current_stamp():
            self.refresh()

    @contextlib.contextmanager
    def active(self):
        global _ACTIVE
        prev, _ACTIVE = _ACTIVE, self
        try:
            yield self
        finally:
            _ACTIVE = prev


def build_for(model, flat=None, params=None, inference=False, fwd16=None, book=None):
    """A bank over every eligible convolution of `model` (nn.Conv2d 1x1 / 3x3 stride 1; the depth-2 Conv3d /
    ConvTranspose3d of the detail branch), plus the concatenations modules declare through `_ebfi_bank_register(bank)`.
    `book` (inference banks): a ScaleBook attached before the sites register, so that sites asking for an fp16 forward image
    (the fused KernelConv -> FAC layout) get one."""
    import torch.nn as nn

    from . import fold3d
    params = list(params) if params is not None else [p for p in model.parameters()]
    bank = WeightBank(params, flat, inference=inference, fwd16=fwd16)
    if book is not None:
        bank.attach_scale_book(book)
    known = set(p.data_ptr() for p in params)
    ok = lambda *ts: all(t is None or t.data_ptr() in known for t in ts)
    for m in model.modules():
        if isinstance(m, nn.Conv2d) and ok(m.weight, m.bias):
            k = m.kernel_size
            if k[0] == k[1] and k[0] in (1, 3) and m.stride == (1, 1) and m.dilation == (1, 1) and m.groups == 1:
                tag = getattr(m, "_ebfi_fwd16", None)
                from .f16scale import FORWARD_LEVELS
                want = tag is not None and fwd16 is not None and FORWARD_LEVELS.index(tag) <= FORWARD_LEVELS.index(fwd16)
                bank.register(m.weight, m.bias, "id", fwd16=tag if want else False)
        elif isinstance(m, nn.Conv3d) and ok(m.weight, m.bias):
            kd, kh, kw = m.kernel_size
            if kd in (1, 3) and kh == kw and kh in (1, 3) and (kh == 1 or m.stride[1] == 1) and m.stride[0] == 1 and \
                    m.padding[0] == kd // 2 and m.dilation == (1, 1, 1) and m.groups == 1:
                bank.register(m.weight, m.bias, "conv3d", fold3d.fold_conv3d_weight, fold3d._rep2)
        elif isinstance(m, nn.ConvTranspose3d) and ok(m.weight, m.bias):
            if m.kernel_size == (3, 4, 4) and m.stride == (1, 2, 2) and m.padding == (1, 1, 1) and m.groups == 1:
                bank.register(m.weight, m.bias, "convT3d", fold3d.fold_conv_transpose3d_weight, fold3d._rep8)
    for m in model.modules():
        hook = getattr(m, "_ebfi_bank_register", None)
        if hook is not None:
            hook(bank)
    return bank
