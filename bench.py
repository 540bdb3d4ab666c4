#!/usr/bin/env python3
"""Headline benchmark: interpolated frames/s of the EVFIAutoEx training step (forward + Lap/census
loss + backward + flat RCCL gradient all-reduce + Adam) at B=8 per GPU, 256x256, synthetic data.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its N ranks itself, as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; weak scaling (B=8 per rank); ONE collective per step (the flat gradient buffer with the fp16 overflow
flag as its last element).  Rank 0 prints ONE JSON line.  For N > 1 a failed hipGraph capture is an error (--strict-graph is
the default there).  Default precision: bf16x3
(split-precision conv operands, fp32-grade accuracy, parity-tested at 1e-3 like the exact fp32 mode); forward + loss +
backward + gradient packing are replayed from one captured hipGraph (--no-graph launches eagerly).  Besides the contract
fields the line carries
  roofline         the dominant hand-written kernel of the step by device time: its bound (matrix cores or HBM, whichever
                   floor is higher for the work it was given), achieved vs peak, PMC traffic per launch; measured with
                   hipEvent pairs the library records on the launch stream during a second, eager pass of the same K steps
  kernels          the same figures for every hand-written kernel of the step
  fp32_exact_mode  the same step with every conv on the exact fp32 matrix cores (second leg of the same run)
  ops              FAC fwd/bwd, DCNv2 fwd/bwd and the composite DCNv2+FAC forward (the north_star target, >= 0.30 of the HBM
                   roofline) at B=8, 128x128 features, timed in the same process (N=1 only)
  inference        BASELINE configs 2 (B=4 256x256, fp32 and bf16x3) and 5 (B=8 720x1280): frames/s, peak memory, and the roofline
                   of the fused KernelConv -> FAC kernel (N=1 only)
  cpu_baseline     the CPU oracle (oracle/model_ref.py + loss_ref.py, a port of the reference path) timed on this box's
                   host cores on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
B_PER_GPU, H, W, TB = 8, 256, 256, 16


F32_MFMA_PEAK_TFS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
BF16_MFMA_PEAK_TFS = 2500.0  # dense bf16 matrix peak (spec, without sparsity)


def host_threads():
    """Threads for the CPU leg: the cores this process may actually run on, capped at the GPU box's
    per-GPU CPU share (16); os.cpu_count() reports the whole 256-thread host there."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(model_args, budget_s=20.0):
    """Oracle fwd+bwd on the host cores, same architecture and input statistics as the GPU run, on a
    bounded sample of ~10-20 s of CPU work: as many B=1 iterations at 256x256 as fit the budget (at most 24; one
    iteration is ~0.8 s on the GPU box's 16 threads), else B=1 at 128x128 (stated in `sample`; the conv-dominated cost
    scales with the pixel count)."""
    from oracle import loss_ref, model_ref
    from ebfi_amd.engine import synthetic_batch
    from ebfi_amd.model import EVFIAutoEx
    cores = host_threads()
    torch.set_num_threads(cores)
    torch.manual_seed(123)
    sd = {k: v.clone().requires_grad_(v.is_floating_point())
          for k, v in EVFIAutoEx(**model_args).state_dict().items()}

    def run(h, w, iters):
        # same branch as the GPU run (RGBLap exposure decision); Frame2Lap via the oracle's numpy restatement
        frame, event, t, gtex, target = synthetic_batch(1, h, w, TB, device="cpu")
        t0 = time.perf_counter()
        for _ in range(iters):
            s, f = model_ref.evfi_forward(sd, model_args, frame, event, t, gtex)
            loss_ref.train_loss(s, f, target).backward()
        return time.perf_counter() - t0

    run(64, 64, 1)                                  # thread pool / allocator warm-up
    t128 = run(128, 128, 1)
    note("cpu baseline: B=1 128x128 fwd+bwd %.2f s on %d threads" % (t128, cores))
    if 4 * t128 * 2 <= budget_s:
        h = w = 256
        iters = max(1, min(24, int(budget_s / (4 * t128)) - 1))
    else:
        h = w = 128
        iters = max(1, min(24, int(budget_s / max(t128, 1e-3)) - 1))
    dt = run(h, w, iters)
    fps = iters / dt
    scale = (h * w) / float(H * W)
    return {"value": round(fps * scale, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d iteration(s) of B=1 %dx%d fwd+bwd (Lap/census loss) of the same model through "
                      "oracle/model_ref.py + oracle/loss_ref.py; measured %.4f it/s, reported as 256x256 frames/s "
                      "(x%.2f pixel-count scaling)" % (iters, h, w, fps, scale)}


T_START = time.perf_counter()


def note(msg):
    """progress on stderr (the JSON line is the only thing on stdout)"""
    print("[bench %7.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


def symbol_of(label):
    """Profiler labels are "<kernel symbol>" or "<kernel symbol>/<role>" (one kernel serving several ops, e.g. the
    forward and the data gradient): device time is ranked per SYMBOL."""
    return label.split("/")[0]


def kernel_table(kernels, elapsed, steps):
    """Per hand-written kernel (grouped by kernel symbol): device time from the library's hipEvent pairs, algorithmic
    work from the launchers (SURVEY.md 8(d) formulas, un-padded).  The bound of a kernel is whichever floor is higher for
    the work it was given: matrix time (matrix flops EXECUTED / dense peak of the operand type; the split-precision x3
    kernels execute 3 matrix flops per algorithmic flop) or HBM time (algorithmic bytes / 8 TB/s).  For matrix-bound
    kernels `achieved` / `frac` are quoted in ALGORITHMIC flops (the useful work); the executed rate is given beside them
    as `executed` / `frac_executed`."""
    grouped = {}
    for label, (launches, total_ms, flops, nbytes) in kernels.items():
        if label.startswith("__") or launches == 0:
            continue
        e = grouped.setdefault(symbol_of(label), {"launches": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0, "roles": {}})
        e["launches"] += launches
        e["total_ms"] += total_ms
        e["flops"] += flops
        e["bytes"] += nbytes
        if "/" in label:
            e["roles"][label.split("/", 1)[1]] = {"launches": launches, "total_ms": round(total_ms, 3)}
    per_kernel = {}
    for name, e in grouped.items():
        launches, total_ms, flops, nbytes = e["launches"], e["total_ms"], e["flops"], e["bytes"]
        entry = {"launches": launches, "launches_per_step": launches / steps, "avg_ms": round(total_ms / launches, 5),
                 "total_ms": round(total_ms, 3)}
        if e["roles"]:
            entry["roles"] = e["roles"]
        secs = total_ms * 1e-3
        mult = 3 if "x3" in name else 1                 # split precision: three MFMAs per product; fp16 / fp32 kernels: one
        half = "bf16" in name or "x3" in name or "f16" in name
        mfma_peak = BF16_MFMA_PEAK_TFS if half else F32_MFMA_PEAK_TFS      # (the fp16 MFMA runs at the bf16 rate)
        t_mfma = mult * flops / (mfma_peak * 1e12)
        t_hbm = nbytes / (HBM_PEAK_GBS * 1e9)
        if nbytes > 0:
            entry["algorithmic_bytes_per_launch"] = nbytes / launches        # (also for matrix-bound kernels: traffic_ratio)
        if flops > 0 and t_mfma >= t_hbm:
            entry.update(bound="mfma", algorithmic_flops_per_launch=flops / launches,
                         achieved=round(flops / secs / 1e12, 2), peak=mfma_peak, unit="TFLOP/s")
            if mult != 1:
                entry.update(matrix_flops_per_algorithmic_flop=mult, executed=round(mult * flops / secs / 1e12, 2),
                             frac_executed=round(mult * flops / secs / 1e12 / mfma_peak, 4))
        elif nbytes > 0:
            entry.update(bound="hbm", algorithmic_bytes_per_launch=nbytes / launches,
                         achieved=round(nbytes / secs / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s")
            if flops > 0:
                entry["algorithmic_flops_per_launch"] = flops / launches
        if "achieved" in entry:
            entry["frac"] = round(entry["achieved"] / entry["peak"], 4)
        per_kernel[name] = entry
    # HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE
    # in separate passes over `bench.py --no-graph`; FETCH_SIZE doubled for the kernels that stage with 16-byte-per-lane
    # loads, MI355X_MICROARCH.md section HBM), next to the algorithmic bytes of the same launch
    pmc = {}
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
    except (OSError, ValueError):
        pass
    for name, entry in per_kernel.items():
        t = pmc.get(name)
        if isinstance(t, dict) and t.get("bytes_per_launch") and entry.get("algorithmic_bytes_per_launch"):
            entry["traffic"] = t["bytes_per_launch"]
            entry["traffic_ratio"] = round(t["bytes_per_launch"] / entry["algorithmic_bytes_per_launch"], 3)
    ranked = sorted((k for k in per_kernel if "frac" in per_kernel[k]), key=lambda k: -per_kernel[k]["total_ms"])
    roofline = None
    if ranked:
        d = per_kernel[ranked[0]]
        roofline = {"kernel": ranked[0], "bound": d["bound"], "achieved": d["achieved"], "peak": d["peak"],
                    "unit": d["unit"], "frac": d["frac"], "traffic": d.get("traffic"), "launches_per_step": d["launches_per_step"],
                    "avg_launch_ms": d["avg_ms"], "share_of_step": round(d["total_ms"] / (1e3 * elapsed), 4)}
        for k in ("executed", "frac_executed", "matrix_flops_per_algorithmic_flop", "roles", "algorithmic_bytes_per_launch",
                  "algorithmic_flops_per_launch", "traffic_ratio"):
            if k in d:
                roofline[k] = d[k]
        if isinstance(pmc.get(ranked[0]), dict):
            roofline["traffic_detail"] = pmc[ranked[0]]
            # (HBM bytes per launch come from the COMMITTED rocprofv3 PMC passes of the same command on the builder's box,
            #  not from this run: counters need their own passes, MI355X_MICROARCH.md)
            roofline["traffic_source"] = "profiles/pmc_traffic.json"
    return per_kernel, roofline


def ops_block(device, iters=20):
    """The two extension ops of the path at the north_star size (B=8, 64 channels, 128x128 features: SURVEY.md 8(d)),
    timed in THIS process with the library's hipEvent pairs: FAC forward / backward, DCNv2 forward / backward, and the
    composite "DCNv2+FAC forward" the BASELINE target (>= 30 % of the HBM roofline) is stated on."""
    from ebfi_amd import _native as N
    from ebfi_amd.dcn import dcn_v2_backward, dcn_v2_forward
    from ebfi_amd.fac import fac_backward, fac_forward
    B, C, K, h, w, dg = B_PER_GPU, 64, 5, H // 2, W // 2, 8
    P = B * h * w
    g = torch.Generator(device="cpu").manual_seed(123)
    rn = lambda *shape: torch.randn(*shape, generator=g).to(device)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(device)
        N.prof_reset()
        N.prof_enable(True)
        for _ in range(iters):
            fn()
        torch.cuda.synchronize(device)
        N.prof_enable(False)
        return {k: v[1] / v[0] for k, v in N.prof_collect().items() if v[0]}

    out = {"shape": {"B": B, "C": C, "h": h, "w": w, "K": K, "deformable_groups": dg}, "iters": iters}
    xp, kern, go = rn(B, C, h + 4, w + 4), rn(B, C * K * K, h, w), rn(B, C, h, w)
    res = torch.empty(B, C, h, w, device=device)
    t = timed(lambda: fac_forward(xp, kern, K, out=res))
    fac_fwd_ms, fac_fwd_bytes = sum(t.values()), 4.0 * P * C * (1 + K * K + 1)
    t = timed(lambda: fac_backward(xp, kern, K, go))
    fac_bwd_ms, fac_bwd_bytes = sum(t.values()), 4.0 * P * C * (K * K + 1 + 1 + 1 + K * K)
    del xp, kern, res
    torch.cuda.empty_cache()
    x, off = rn(B, C, h, w), rn(B, dg * 18, h, w) * 2
    msk, wt, bias = torch.sigmoid(rn(B, dg * 9, h, w)), rn(C, C, 3, 3) / 24, rn(C)
    cfg = ((1, 1), (1, 1), (1, 1), dg)
    t = timed(lambda: dcn_v2_forward(x, wt, bias, off, msk, *cfg))
    dcn_fwd_ms = sum(t.values())
    dcn_fwd_bytes = 4.0 * (P * (C + 2 * dg * 9 + dg * 9 + C) + C * C * 9)
    dcn_flops = 2.0 * P * C * 9 * (4 + C)
    tb = timed(lambda: dcn_v2_backward(x, wt, bias, off, msk, go, *cfg))
    hbm = lambda by, ms: {"ms": round(ms, 4), "algorithmic_bytes": by, "GBps": round(by / ms / 1e6, 1),
                          "frac_hbm": round(by / ms / 1e6 / HBM_PEAK_GBS, 4)}
    out["fac_forward"] = hbm(fac_fwd_bytes, fac_fwd_ms)
    out["fac_backward"] = hbm(fac_bwd_bytes, fac_bwd_ms)
    out["dcn_forward"] = dict(hbm(dcn_fwd_bytes, dcn_fwd_ms), TFLOPs=round(dcn_flops / dcn_fwd_ms / 1e9, 2),
                              frac_f32_mfma=round(dcn_flops / dcn_fwd_ms / 1e9 / F32_MFMA_PEAK_TFS, 4))
    # backward, algorithmic: reads x, offset, mask, grad_out and the weight; writes grad_input, grad_offset, grad_mask, grad_weight,
    # grad_bias (no column tensor, no slabs); matrix work = colgrad (W^T . grad_out) + the weight gradient, 2 * P * C * 9 * C each,
    # plus the sampling walk's 4 FMAs per sample forward and 4 per coordinate gradient
    dcn_bwd_ms = sum(tb.values())
    dcn_bwd_bytes = 4.0 * (P * (C + 3 * dg * 9 + C) + P * (C + 3 * dg * 9) + 2 * C * C * 9 + C)
    dcn_bwd_flops = 2.0 * P * C * 9 * (2 * C + 8)
    out["dcn_backward"] = dict(hbm(dcn_bwd_bytes, dcn_bwd_ms), TFLOPs=round(dcn_bwd_flops / dcn_bwd_ms / 1e9, 2),
                               frac_f32_mfma=round(dcn_bwd_flops / dcn_bwd_ms / 1e9 / F32_MFMA_PEAK_TFS, 4),
                               bound="mfma (exact fp32 matrix cores: the floor of the two products is %.0f us, of the bytes %.0f us)"
                                     % (1e6 * dcn_bwd_flops / (F32_MFMA_PEAK_TFS * 1e12), 1e6 * dcn_bwd_bytes / (HBM_PEAK_GBS * 1e9)),
                               kernels_ms={k: round(v, 4) for k, v in sorted(tb.items())})
    out["dcn_fac_forward"] = dict(hbm(fac_fwd_bytes + dcn_fwd_bytes, fac_fwd_ms + dcn_fwd_ms),
                                  target_frac_hbm=0.30, note="BASELINE.json north_star target: DCNv2+FAC forward at "
                                  "B=8 256x256 (128x128 features), exact fp32 kernels, sum of the two launches")
    return out


def inference_block(device):
    """BASELINE.json configs 2 and 5 (inference through the reference's per-timestamp loop, infer_ours.py:113-118) in THIS
    process: B=4 256x256 with the exact fp32 kernels and in the default split-precision mode, and B=8 720x1280 (HD).  One
    timed clip = the timestamp-independent prefix once + `num_ts` replays of the per-timestamp hipGraph
    (ebfi_amd.engine.ClipInterpolator); O(1)-gain random weights (the x0.1 initialisation outputs the constant 0.5).  The
    roofline entry is the fused KernelConv(128->1600) -> FAC kernel, the dominant launch of these configs, timed with the
    library's hipEvent pairs during one extra eager (no-graph) timestamp."""
    from ebfi_amd import _native as N
    from ebfi_amd.engine import DEFAULT_MODEL_ARGS, ClipInterpolator, synthetic_batch
    from ebfi_amd.model import EVFIAutoEx
    torch.manual_seed(123)
    model = EVFIAutoEx(**DEFAULT_MODEL_ARGS).to(device).eval()
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() > 1:
                p.copy_(torch.randn_like(p) * (1.2 / p[0].numel() ** 0.5))
            else:
                p.add_(0.05 * torch.randn_like(p))
    out = {"note": "inference, one clip = prefix (feature extractors + exposure decision) once + per-timestamp hipGraph replays "
                   "(`timestamps_per_pass` timestamps of the clip per replay, as a batch) into a preallocated [B, num_ts, 3, H, W] result; frames/s = B * num_ts / wall time of the MEDIAN of "
                   "`clips_timed` clips; prefix / per-timestamp device time from event pairs on the stream"}
    # (tag, B, height, width, precision, timestamps per clip, timed clips): every leg times >= 3 clips of >= 8 timestamps into ONE
    # preallocated result tensor and reports the MEDIAN clip (round 5 timed a single 4-timestamp clip: one stall on a fresh
    # box -- 47.95 frames/s against 120-127 elsewhere -- was the whole measurement)
    legs = [("config2_fp32", 4, 256, 256, "fp32", 8, 5), ("config2_bf16x3", 4, 256, 256, "bf16x3", 16, 5),
            ("config5_hd_bf16x3", 8, 720, 1280, "bf16x3", 8, 3)]
    for tag, B, h, w, prec, num_ts, clips in legs:
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(device)
        interp = ClipInterpolator(model, precision=prec, graph=True, hoist=True)
        frame, event, _, gtex, _ = synthetic_batch(B, h, w, TB, device=device, seed=123)
        stamps = [i / float(num_ts) for i in range(num_ts)]
        res = torch.empty(B, num_ts, 3, h, w, device=device)
        for _ in range(2):                                       # untimed: allocator, capture (keyed by the timestamps per pass), first replays
            interp(frame, event, gtex, stamps, out=res)
        torch.cuda.synchronize(device)
        wall, enc, dec = [], [], []
        for _ in range(clips):
            t0 = time.perf_counter()
            interp(frame, event, gtex, stamps, out=res, timing=True)      # (timing=True ends with an event synchronise)
            torch.cuda.synchronize(device)
            wall.append(time.perf_counter() - t0)
            enc.append(interp.last_timing["encode_ms"])
            dec.append(interp.last_timing["decode_ms"])
        med = lambda v: sorted(v)[len(v) // 2]
        dt = med(wall)
        entry = {"B": B, "height": h, "width": w, "precision": prec, "num_ts": num_ts, "clips_timed": clips,
                 "frames_per_s": round(B * num_ts / dt, 2), "statistic": "median clip",
                 "frames_per_s_per_clip": [round(B * num_ts / v, 2) for v in wall],
                 "ms_per_timestamp": round(1e3 * dt / num_ts, 3),
                 "timestamps_per_pass": interp.last_group,
                 "prefix_encode_ms": round(med(enc), 3), "decode_ms_per_timestamp": round(med(dec) / num_ts, 3),
                 "wall_over_device_time": round(1e3 * dt / (med(enc) + med(dec)), 3),
                 "peak_memory_GB": round(torch.cuda.max_memory_allocated(device) / 1e9, 2),
                 "output_mean": round(float(res.mean().item()), 5), "finite": bool(torch.isfinite(res).all().item())}
        if prec == "bf16x3":
            # one eager pass of the same size as the timed replays (`timestamps_per_pass` timestamps as one batch), with event pairs
            gk = interp.last_group
            eager = ClipInterpolator(model, precision=prec, graph=False, hoist=True, group=gk)
            eager(frame, event, gtex, stamps[:gk])
            torch.cuda.synchronize(device)
            N.prof_reset()
            N.prof_enable(True)
            eager(frame, event, gtex, stamps[:gk])
            torch.cuda.synchronize(device)
            N.prof_enable(False)
            k = N.prof_collect()
            per_kernel, _ = kernel_table(k, 1.0, gk)
            top = sorted(per_kernel.items(), key=lambda kv: -kv[1]["total_ms"])[:5]
            entry["top_kernels_ms_per_timestamp"] = {n: round(v["total_ms"] / gk, 4) for n, v in top}
            # the fused KernelConv -> FAC launch: fp16 operands (one matrix-core product per tap; round 6) or split precision (three)
            label = next((l for l in ("conv_fwd_f16_ws/kernelconv_fac_img", "conv_fwd_f16_ws/kernelconv_fac",
                                      "conv_fwd_bf16x3_ws/kernelconv_fac") if k.get(l) and k[l][0]), None)
            if label is not None:
                n, ms, flops, nbytes = k[label]
                secs = ms * 1e-3
                mult = 3 if "x3" in label else 1
                t_mfma, t_hbm = mult * flops / (BF16_MFMA_PEAK_TFS * 1e12), nbytes / (HBM_PEAK_GBS * 1e9)
                entry["kernelconv_fac_fused"] = {
                    "kernel": label, "launches": n, "avg_ms": round(ms / n, 4),
                    "bound": "mfma" if t_mfma >= t_hbm else "hbm", "algorithmic_flops_per_launch": flops / n,
                    "algorithmic_bytes_per_launch": nbytes / n, "achieved": round(flops / secs / 1e12, 2), "peak": BF16_MFMA_PEAK_TFS,
                    "unit": "TFLOP/s", "frac": round(flops / secs / 1e12 / BF16_MFMA_PEAK_TFS, 4),
                    "matrix_flops_per_algorithmic_flop": mult, "executed": round(mult * flops / secs / 1e12, 2),
                    "frac_executed": round(mult * flops / secs / 1e12 / BF16_MFMA_PEAK_TFS, 4),
                    "GBps": round(nbytes / secs / 1e9, 1), "frac_hbm": round(nbytes / secs / 1e9 / HBM_PEAK_GBS, 4)}
            del eager
        out[tag] = entry
        note("inference %s: %.1f frames/s, peak %.1f GB" % (tag, entry["frames_per_s"], entry["peak_memory_GB"]))
        del interp, frame, event, res
    del model
    torch.cuda.empty_cache()
    return out


LINE_LIMIT = 4096          # the driver parses ONE JSON line from stdout; round 5's 20 KB line came back unparsed


def compact_line(full, detail_path=None):
    """The ONE line rank 0 prints: the contract fields, a short `config`, the dominant kernel's `roofline`, `cpu_baseline`, the
    north_star op figure and three inference scalars -- under LINE_LIMIT bytes whatever the per-kernel tables hold.  Everything
    else (every kernel of both legs, the op block, the inference block) goes to `bench_detail.json` and to stderr."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    line = {k: full.get(k) for k in keep}
    cfg = full.get("config") or {}
    guard = cfg.get("fp16_overflow_guard") or {}
    line["config"] = {
        "workload": cfg.get("workload"), "global_batch": cfg.get("global_batch"), "parallelism": cfg.get("parallelism"),
        "world_size": cfg.get("world_size"), "collective_backend": cfg.get("collective_backend"),
        "collectives_per_step": cfg.get("collectives_per_step"), "launch": (cfg.get("launch") or "")[:120],
        "graph_capture_failed": cfg.get("graph_capture_failed"), "untimed_steps": cfg.get("untimed_steps"),
        "precision": cfg.get("precision_short"), "fp16_steps_skipped": guard.get("optimiser_steps_skipped"),
        "loss": cfg.get("loss")}
    rf = full.get("roofline")
    if rf is not None:
        fields = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_ratio",
                  "launches_per_step", "avg_launch_ms", "share_of_step", "algorithmic_bytes_per_launch",
                  "algorithmic_flops_per_launch", "executed", "frac_executed", "matrix_flops_per_algorithmic_flop", "timing")
        rf = {k: rf[k] for k in fields if k in rf}
    line["roofline"] = rf
    line["cpu_baseline"] = full.get("cpu_baseline")
    if "dcn_fac_forward_frac_hbm" in full:
        line["dcn_fac_forward_frac_hbm"] = full["dcn_fac_forward_frac_hbm"]
    if full.get("fp32_exact_mode"):
        line["fp32_exact_ms_per_step"] = full["fp32_exact_mode"].get("ms_per_step")
    inf = full.get("inference") or {}
    scal = {tag: inf[tag].get("frames_per_s") for tag in ("config2_fp32", "config2_bf16x3", "config5_hd_bf16x3") if tag in inf}
    if scal:
        line["inference_frames_per_s"] = scal
    if detail_path:
        line["detail"] = os.path.relpath(detail_path, ROOT)
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:            # never print a line the driver cannot take: shed the optional parts, longest first
        for victim in ("inference_frames_per_s", "fp32_exact_ms_per_step", "detail"):
            line.pop(victim, None)
        if line.get("cpu_baseline") and len(json.dumps(line)) >= LINE_LIMIT:
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:160]
        line["config"]["workload"] = (line["config"]["workload"] or "")[:200]
        text = json.dumps(line)
    assert len(text) < LINE_LIMIT, len(text)
    return text


def write_detail(full, path):
    """The full record (per-kernel tables of both legs, op block, inference block) next to the line: `bench_detail.json` at the
    repo root (and under gpurun_out/ when that directory exists, so that it travels back from a GPU box), and on stderr."""
    text = json.dumps(full, indent=1)
    written = None
    for target in (path, os.path.join(ROOT, "gpurun_out", os.path.basename(path))):
        if target != path and (os.path.dirname(os.path.abspath(path)) != ROOT or not os.path.isdir(os.path.dirname(target))):
            continue
        try:
            with open(target, "w") as fh:
                fh.write(text)
            written = written or target
        except OSError as err:
            note("could not write %s: %s" % (target, err))
    # (stderr copy on ONE prefixed line: nothing but the compact line may look like the bench's JSON line to a log parser)
    print("[bench detail] " + json.dumps(full), file=sys.stderr, flush=True)
    return written


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher's environment: run
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same flags>`
    as a child process and return its exit code.  Nothing here touches the GPU (a process that has initialised HIP must not
    be replaced or forked into ranks); stdout / stderr are inherited, so rank 0's JSON line is this command's JSON line."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as s:                 # a free port on the loopback interface
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("MASTER_PORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    note("--gpus %d without WORLD_SIZE: launching the ranks as a child process: %s" % (n, " ".join(cmd[1:])))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16x3", choices=["fp32", "bf16x3", "bf16"],
                    help="matrix-core operands of the 3x3/1x1 convs: bf16x3 = split bf16 hi+lo pairs, fp32-grade accuracy "
                         "(default; parity-checked at 1e-3 like fp32); fp32 = exact fp32 matrix cores; bf16 = single rounding")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel of the step eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ops", action="store_true", help="skip the FAC / DCNv2 op block (north_star target figure)")
    ap.add_argument("--no-inference", action="store_true", help="skip the inference block (BASELINE configs 2 and 5)")
    ap.add_argument("--strict-graph", dest="strict_graph", action="store_true", default=None,
                    help="exit non-zero when the hipGraph capture of the step failed and the engine fell back to eager launches "
                         "(default for --gpus > 1: one rank silently 20 %% slower would drag every rank; with one GPU the default "
                         "is to report it as config.graph_capture_failed and in config.launch)")
    ap.add_argument("--no-strict-graph", dest="strict_graph", action="store_false")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full record goes (per-kernel tables of both legs, op block, inference block); the ONE line on "
                         "stdout carries the contract fields, roofline and cpu_baseline only (< 4 KB)")
    ap.add_argument("--no-bf16-leg", "--no-extra-legs", dest="no_extra_legs", action="store_true",
                    help="skip the secondary measurement of the same step in the exact-fp32 mode")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as a CHILD process (never an exec), before anything
        # in this process has touched the GPU; relay the child's output (rank 0 prints the one JSON line) and its exit code
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU: python bench.py --gpus N, or "
                         "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N)" % (args.gpus, world))
    # the torch.distributed path at world size 1 (EBFI_BENCH_FORCE_DIST=1): RCCL communicator, its watchdog thread next to the
    # hipGraph capture, the collective on the packed bucket -- everything a one-GPU box can exercise of the N > 1 path
    force_dist = os.environ.get("EBFI_BENCH_FORCE_DIST", "0") == "1"
    strict_graph = args.strict_graph if args.strict_graph is not None else (world > 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback for the product path)")
    # Rehearsal hook for a one-GPU box: EBFI_BENCH_REHEARSAL=1 puts every rank on device 0 and uses gloo, so the
    # multi-rank code path (barriers, flat-bucket all-reduce, max-over-ranks timing) can be exercised without 8 GPUs.
    rehearsal = os.environ.get("EBFI_BENCH_REHEARSAL", "0") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        if force_dist:
            os.environ["EBFI_FORCE_COLLECTIVES"] = "1"          # (ebfi_amd.dp: collectives also at world size 1)
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from ebfi_amd import _native as N
    from ebfi_amd.engine import DEFAULT_MODEL_ARGS, Engine, synthetic_batch

    # (strictness is handled HERE: a failed capture still yields the JSON line, flagged, and then a non-zero exit code)
    eng = Engine(DEFAULT_MODEL_ARGS, device=device, precision=args.precision, lr=1e-4, seed=123, graph=not args.no_graph,
                 strict_graph=False)
    batch = synthetic_batch(B_PER_GPU, H, W, TB, device=device, seed=123, rank=rank)   # resident in HBM

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(device)

    untimed = {}

    def timed_run(tag):
        """W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize (max over ranks);
        afterwards the same K steps once more, eagerly, with the library's hipEvent pairs switched on, for the
        per-kernel table (a hipGraph replay cannot carry per-kernel event pairs, and the pairs cost a little time,
        so they stay out of the timed region)."""
        # the engine's first steps are not the steady state: with the fp16 backward it takes `calibration_steps` eager steps
        # (operand scales measured just in time) and captures the graph on the next one.  If W is too small to cover
        # them, extra untimed steps are added so that the timed region never holds a capture (reported as untimed_steps).
        settle = 0
        if eng.use_graph:
            cal = max(0, eng.calibration_steps - eng._steps_run) if (eng.book is not None and eng.precision == "bf16x3") else 0
            capture = 0 if any(k[0] == eng.precision for k in eng._graphs) else 1
            settle = max(0, cal + capture - args.warmup)
        untimed[tag] = args.warmup + settle
        for i in range(args.warmup + settle):
            eng.train_step(*batch)
            torch.cuda.synchronize(device)
            note("%s warm-up step %d done" % (tag, i))
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = eng.train_step(*batch)
        sync()
        elapsed = time.perf_counter() - t0
        t_max = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else device)
        if dist.is_initialized():
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        note("%s timed region: %d steps in %.3f s (max over ranks %.3f s)" % (tag, args.steps, elapsed, t_max.item()))
        loss_value = float(loss.item())
        graph_mode, eng.use_graph = eng.use_graph, False
        eng.train_step(*batch)                      # eager path warm (allocator) before the profiled pass
        sync()
        N.prof_reset()
        N.prof_enable(True)
        prof_elapsed = 0.0
        for _ in range(args.steps):
            t0 = time.perf_counter()
            eng.train_step(*batch)
            sync()
            prof_elapsed += time.perf_counter() - t0
            N.prof_fold()                           # once per step: the pending list never overflows
        N.prof_enable(False)
        eng.use_graph = graph_mode
        kernels = N.prof_collect()                  # raises if a single launch went untimed
        note("%s profiled eager pass: %d steps in %.3f s" % (tag, args.steps, prof_elapsed))
        return t_max.item(), kernels, loss_value, prof_elapsed

    note("engine + batch ready on %s (rank %d/%d)" % (device, rank, world))
    elapsed, kernels, loss, prof_elapsed = timed_run(args.precision)
    extra_leg = None
    if args.precision == "bf16x3" and not args.no_extra_legs:
        eng.precision = "fp32"
        e2, k2, l2, pe2 = timed_run("fp32-leg")
        eng.precision = args.precision
        pk2, rf2 = kernel_table(k2, pe2, args.steps)
        extra_leg = {"note": "the same step with every conv on the exact fp32 matrix cores (v_mfma_f32_32x32x2_f32); informational",
                     "value": round(world * B_PER_GPU * args.steps / e2, 3), "unit": "frames/s",
                     "ms_per_step": round(1e3 * e2 / args.steps, 3), "loss": l2, "roofline": rf2,
                     "kernels": {k: {f: v[f] for f in ("launches", "launches_per_step", "total_ms", "bound", "achieved", "unit", "frac", "frac_executed") if f in v}
                                 for k, v in pk2.items()}}

    # replicas must still be identical after the timed steps (same init, every rank applied the same averaged gradient):
    # every rank contributes a checksum of its parameters, rank 0 compares them bit for bit
    skipped = eng.book.skipped_steps() if eng.book is not None else None      # fp16 backward: steps the overflow guard skipped
    flat = eng.optimizer.flat.detach().double()
    check = torch.stack([flat.sum(), flat.square().sum()]).to("cpu" if rehearsal or world == 1 else device)
    checks = [check]
    if world > 1:
        checks = [torch.empty_like(check) for _ in range(world)]
        dist.all_gather(checks, check)
    checks = [tuple(c.cpu().tolist()) for c in checks]
    if any(c != checks[0] for c in checks):
        raise SystemExit("bench.py: replicas diverged, per-rank parameter checksums %r" % (checks,))

    # a capture that could not be taken silently costs ~20 % of the step: every rank reports, the line carries the flag
    capture_failed, capture_err = bool(eng.graph_capture_failed), eng.graph_capture_error
    if world > 1:
        flag = torch.tensor([1.0 if capture_failed else 0.0], device="cpu" if rehearsal else device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        capture_failed = bool(flag.item() > 0)
    if capture_failed:
        note("WARNING: hipGraph capture failed (%s): the timed region ran eager launches" % (capture_err or "on another rank"))

    if rank == 0:
        per_kernel, roofline = kernel_table(kernels, prof_elapsed, args.steps)
        if roofline is not None:
            roofline["measured"] = ("hipEvent pairs around every launch of the kernel during a second, eager pass of the same "
                                    "%d steps right after the timed region (%.3f ms/step with the pairs on)"
                                    % (args.steps, 1e3 * prof_elapsed / args.steps))
            roofline["timing"] = "hipEvent pairs on the launch stream, eager pass of the same %d steps" % args.steps
        out = {
            "metric": "interpolated frames/sec (train fwd+bwd) at B=8 256x256",
            "value": round(world * B_PER_GPU * args.steps / elapsed, 3),
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x3": ("bf16x3 fwd%s / f16 bwd" % {None: "", "filters": " (f16 KernelConv)",
                                                                   "all": " (f16 KernelConv, ResidualControl)"}[eng.book.forward_f16])
                                if eng.book is not None else "bf16x3", "bf16": "bf16"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "EVFIAutoEx (config/train_ours.yml defaults, 5.69 M params) train step: fwd + "
                                   "Lap/census loss + bwd + flat grad all-reduce + Adam; B=%d per GPU, %dx%d frames, "
                                   "TB=%d event bins, Poisson(0.35) event counts" % (B_PER_GPU, H, W, TB),
                       "global_batch": world * B_PER_GPU, "parallelism": "dp%d" % world, "loss": loss,
                       "world_size": world, "collective_backend": (dist.get_backend() if dist.is_initialized() else None),
                       "collectives_per_step": 1 if dist.is_initialized() else 0,
                       "replica_param_checksum": {"sum": checks[0][0], "sum_sq": checks[0][1], "ranks_identical": True},
                       "precision_short": {"fp32": "fp32 storage, exact fp32 MFMA",
                                           "bf16x3": "fp32 storage+accumulate; fwd convs bf16 hi+lo (3 MFMA/product)" +
                                                     ("; bwd convs fp16 operands, delayed pow2 scales" if eng.book is not None else ""),
                                           "bf16": "fp32 storage+accumulate; conv operands rounded to bf16"}[args.precision],
                       "precision": {"fp32": "fp32 tensors, exact fp32 matrix cores",
                                     "bf16x3": "fp32 tensors and accumulation; forward conv operands split into bf16 hi+lo pairs, 3 MFMAs "
                                               "per product (~1e-5 of fp32, parity-tested at 1e-3 like the fp32 mode)" +
                                               ("; data / weight gradients of the 3x3 layers with fp16 operands, 1 MFMA per product, "
                                                "delayed power-of-two operand scales (packed gradient within 5e-3 of the oracle's: "
                                                "tests/test_gpu_model.py::test_benchmarked_step_vs_oracle)" if eng.book is not None else ""),
                                     "bf16": "fp32 tensors and accumulation; conv operands rounded once to bf16"}[args.precision],
                       "launch": "hipGraph replay of fwd+loss+bwd+grad packing; all-reduce and Adam eager" if eng.use_graph else
                       ("eager (hipGraph capture FAILED on at least one rank: %s)" % capture_err if capture_failed else "eager"),
                       "graph_capture_failed": capture_failed,
                       "untimed_steps": untimed[args.precision],
                       "fp16_overflow_guard": None if eng.book is None else
                       {"optimiser_steps_skipped": skipped, "operand_scale_slots": len(eng.book.index)},
                       "rehearsal_single_device_gloo": rehearsal},
            "roofline": roofline,
            "kernels": per_kernel,
            "fp32_exact_mode": extra_leg,
        }
        if world == 1 and not args.no_ops:
            note("op block: FAC / DCNv2 at B=8, 128x128 features ...")
            del eng
            torch.cuda.empty_cache()
            out["ops"] = ops_block(device)
            out["dcn_fac_forward_frac_hbm"] = out["ops"]["dcn_fac_forward"]["frac_hbm"]
        if world == 1 and not args.no_inference:
            note("inference block: configs 2 (B=4 256x256) and 5 (B=8 720x1280) ...")
            out["inference"] = inference_block(device)
        if world == 1 and not args.no_cpu_baseline:
            note("cpu baseline (oracle on host cores) ...")
            out["cpu_baseline"] = cpu_baseline(dict(DEFAULT_MODEL_ARGS))
            note("cpu baseline done")
        else:
            out["cpu_baseline"] = None
        detail_path = write_detail(out, args.detail)
        print(compact_line(out, detail_path), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if strict_graph and capture_failed:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
