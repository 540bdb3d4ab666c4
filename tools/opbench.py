#!/usr/bin/env python3
"""Op-level roofline microbench for the hand-written kernels at the BASELINE.json sizes
(B=8, 64 channels, feature resolution 128x128; `--hd` = 360x640).  Prints one JSON line per op
with device time (hipEvent pairs recorded by libebfi_hip.so on the launch stream), algorithmic
bytes / flops and the achieved fraction of the MI355X roofline.

    python tools/opbench.py [--iters 20] [--B 8] [--hd]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))

import torch  # noqa: E402

HBM_PEAK, F32_MFMA_PEAK = 8000.0, 157.3     # GB/s, TFLOP/s (MI355X_MICROARCH.md)


def timed(fn, iters, names):
    from ebfi_amd import _native as N
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    N.prof_reset()
    N.prof_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    N.prof_enable(False)
    prof = N.prof_collect()
    return {k: v[1] / max(v[0], 1) for k, v in prof.items() if k in names}, prof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--hd", action="store_true")
    ap.add_argument("--ops", default="fac,dcn,conv")
    ap.add_argument("--bf16", action="store_true", help="bf16 matrix-core operands for the conv kernels")
    ap.add_argument("--x3", action="store_true", help="split-precision (bf16 hi+lo, 3 MFMAs) forward / data gradient")
    a = ap.parse_args()
    from ebfi_amd.dcn import dcn_v2_backward, dcn_v2_forward
    from ebfi_amd.fac import fac_backward, fac_forward
    B, C, K = a.B, 64, 5
    h, w = (360, 640) if a.hd else (128, 128)
    P = B * h * w
    dev = "cuda"
    torch.manual_seed(123)
    lines = []
    if "fac" in a.ops:
        x = torch.randn(B, C, h + 4, w + 4, device=dev)
        k = torch.randn(B, C * 25, h, w, device=dev)
        g = torch.randn(B, C, h, w, device=dev)
        out = torch.empty(B, C, h, w, device=dev)
        t, _ = timed(lambda: fac_forward(x, k, K, out=out), a.iters, {"fac_fwd_tile_f32"})
        by = 4 * P * C * 27
        for n, ms in t.items():
            lines.append({"op": "fac_forward", "kernel": n, "ms": round(ms, 4), "algorithmic_bytes": by,
                          "GBps": round(by / ms / 1e6, 1), "frac_hbm": round(by / ms / 1e6 / HBM_PEAK, 4)})
        t, _ = timed(lambda: fac_backward(x, k, K, g), a.iters, {"fac_bwd_rows_f32"})
        by = 4 * P * C * 53
        for n, ms in t.items():
            lines.append({"op": "fac_backward", "kernel": n, "ms": round(ms, 4), "algorithmic_bytes": by,
                          "GBps": round(by / ms / 1e6, 1), "frac_hbm": round(by / ms / 1e6 / HBM_PEAK, 4)})
        del x, k, g, out
        torch.cuda.empty_cache()
    if "dcn" in a.ops:
        dg = 8
        x = torch.randn(B, C, h, w, device=dev)
        off = torch.randn(B, dg * 18, h, w, device=dev) * 2
        msk = torch.sigmoid(torch.randn(B, dg * 9, h, w, device=dev))
        wt = torch.randn(C, C, 3, 3, device=dev) / 24
        bias = torch.randn(C, device=dev)
        g = torch.randn(B, C, h, w, device=dev)
        cfg = ((1, 1), (1, 1), (1, 1), dg)
        prod = "bf16x3" if a.x3 else "fp32"                         # --x3: the product of the forward in split precision
        t, _ = timed(lambda: dcn_v2_forward(x, wt, bias, off, msk, *cfg, product=prod), a.iters, {"dcn_fwd_f32", "dcn_fwd_bf16x3"})
        by = 4 * (P * (C + 2 * dg * 9 + dg * 9 + C) + C * C * 9)
        fl = 2.0 * P * C * 9 * (4 + C)
        for n, ms in t.items():
            lines.append({"op": "dcn_forward", "kernel": n, "ms": round(ms, 4), "algorithmic_bytes": by,
                          "GBps": round(by / ms / 1e6, 1), "frac_hbm": round(by / ms / 1e6 / HBM_PEAK, 4),
                          "TFLOPs": round(fl / ms / 1e9, 2), "frac_f32_mfma": round(fl / ms / 1e9 / F32_MFMA_PEAK, 4)})
        names = {"dcn_bwd_data_f32", "dcn_bwd_weight_f32", "dcn_bwd_reduce_f32"}
        t, _ = timed(lambda: dcn_v2_backward(x, wt, bias, off, msk, g, *cfg), a.iters, names)
        for n, ms in t.items():
            lines.append({"op": "dcn_backward", "kernel": n, "ms": round(ms, 4)})
    if "conv" in a.ops:
        from ebfi_amd import conv as convmod
        from ebfi_amd.conv import conv_bias_act
        sfx = "bf16" if a.bf16 else "f32"
        convmod.set_compute_dtype("bf16" if a.bf16 else ("bf16x3" if a.x3 else "fp32"))
        names = {"conv_fwd_%s/fwd" % sfx, "conv_fwd_%s/dgrad" % sfx, "conv_wgrad_" + sfx, "conv_wgrad_reduce_f32", "conv_pack_w_bf16",
                 "conv_fwd_bf16x3_db/fwd", "conv_fwd_bf16x3_db/dgrad", "conv_wgrad_x3", "conv_wgrad_x3_ws",
                 "conv_fwd_bf16x3_ws/fwd", "conv_fwd_bf16x3_ws/dgrad"}
        peak = 2500.0 if a.bf16 else F32_MFMA_PEAK
        for (cin, cout, hh, ww, tag) in [(64, 64, h, w, "ResidualControl 64->64"), (128, 64, h, w, "Conv5 128->64"),
                                         (128, 1600, h, w, "KernelConv 128->1600"), (64, 64, 2 * h, 2 * w, "Recon 64->64 @2x")]:
            x = torch.randn(B, cin, hh, ww, device=dev).requires_grad_()
            wt = (torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5).requires_grad_()
            bs = torch.zeros(cout, device=dev, requires_grad=True)
            g = torch.randn(B, cout, hh, ww, device=dev)

            def step():
                x.grad = wt.grad = bs.grad = None
                conv_bias_act(x, wt, bs, 1, 1, 1, 0.01).backward(g)
            t, _ = timed(step, max(3, a.iters // 4), names)
            fl = 2.0 * B * hh * ww * cin * cout * 9
            for n, ms in sorted(t.items()):
                e = {"op": "conv3x3 " + tag, "kernel": n, "ms": round(ms, 4)}
                if n not in ("conv_wgrad_reduce_f32", "conv_pack_w_bf16"):
                    e.update(TFLOPs=round(fl / ms / 1e9, 2), frac_mfma_peak=round(fl / ms / 1e9 / peak, 4))
                lines.append(e)
            del x, wt, bs, g
            torch.cuda.empty_cache()
    for l in lines:
        l.update(B=B, h=h, w=w)
        print(json.dumps(l), flush=True)
    fwd = [l for l in lines if l["op"] in ("fac_forward", "dcn_forward")]
    if len(fwd) == 2:
        ms = sum(l["ms"] for l in fwd)
        by = sum(l["algorithmic_bytes"] for l in fwd)
        print(json.dumps({"op": "dcn+fac forward (BASELINE target >= 0.30); DCN product on %s matrix cores" % ("bf16x3 split-precision" if a.x3 else "exact fp32"),
                          "ms": round(ms, 4), "algorithmic_bytes": by,
                          "GBps": round(by / ms / 1e6, 1), "frac_hbm": round(by / ms / 1e6 / HBM_PEAK, 4)}), flush=True)


if __name__ == "__main__":
    main()
