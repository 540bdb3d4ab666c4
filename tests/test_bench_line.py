"""bench.py prints ONE JSON line the driver must be able to parse: the contract fields, the dominant kernel's roofline and the
CPU baseline, under 4 KB, whatever the per-kernel tables hold (round 5's 20 KB line came back `parsed: null`).  CPU only: the
line is built from a canned kernel table shaped like a real run's (profiles/r05/bench_final.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def canned_kernels(n=90):
    """label -> (launches, total_ms, flops, bytes) like ebfi_amd._native.prof_collect()"""
    k = {"conv_fwd_f16_ws/img_img": (230, 10.8, 230 * 31.5e9, 230 * 94.4e6), "conv_fwd_f16_ws/img_f32": (370, 21.1, 370 * 31.5e9, 370 * 94.4e6),
         "conv_fwd_bf16x3_ws/fwd": (440, 31.4, 440 * 19.3e9, 440 * 124e6), "fac_fwd_tile_f32/p16": (10, 0.9, 10 * 0.4e9, 10 * 5e8)}
    for i in range(n):
        k["some_long_kernel_symbol_name_%03d/role_with_a_long_name" % i] = (10 + i, 0.1 * (i + 1), 1e9 * i, 1e6 * (i + 1))
    return k


def full_record():
    per_kernel, roofline = bench.kernel_table(canned_kernels(), 0.29, 10)
    assert roofline["kernel"] == "conv_fwd_f16_ws" and roofline["launches_per_step"] == 60.0
    roofline["measured"] = "x" * 300
    roofline["timing"] = "hipEvent pairs on the launch stream, eager pass of the same 10 steps"
    cfg = {"workload": "EVFIAutoEx (config/train_ours.yml defaults, 5.69 M params) train step: fwd + Lap/census loss + bwd + flat "
                       "grad all-reduce + Adam; B=8 per GPU, 256x256 frames, TB=16 event bins, Poisson(0.35) event counts",
           "global_batch": 8, "parallelism": "dp1", "loss": 465607.9, "world_size": 1, "collective_backend": None,
           "collectives_per_step": 0, "replica_param_checksum": {"sum": 1.0, "sum_sq": 2.0, "ranks_identical": True},
           "precision": "y" * 600, "precision_short": "fp32 storage+accumulate; fwd convs bf16 hi+lo (3 MFMA/product); bwd convs fp16",
           "launch": "hipGraph replay of fwd+loss+bwd+grad packing; all-reduce and Adam eager", "graph_capture_failed": False,
           "untimed_steps": 5, "fp16_overflow_guard": {"optimiser_steps_skipped": 0, "operand_scale_slots": 321},
           "rehearsal_single_device_gloo": False}
    return {"metric": "interpolated frames/sec (train fwd+bwd) at B=8 256x256", "value": 544.2, "unit": "frames/s", "n_gpus": 1,
            "steps": 20, "warmup": 5, "ms_per_step": 14.7, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3 fwd (f16 KernelConv) / f16 bwd", "data": "synthetic", "config": cfg, "roofline": roofline,
            "kernels": per_kernel, "fp32_exact_mode": {"ms_per_step": 45.25, "kernels": per_kernel, "roofline": roofline},
            "ops": {"dcn_fac_forward": {"frac_hbm": 0.447}, "blob": ["z" * 50] * 100}, "dcn_fac_forward_frac_hbm": 0.447,
            "inference": {"config2_fp32": {"frames_per_s": 700.1}, "config2_bf16x3": {"frames_per_s": 1351.0},
                          "config5_hd_bf16x3": {"frames_per_s": 126.7, "top_kernels_ms_per_timestamp": {"k%d" % i: i for i in range(40)}}},
            "cpu_baseline": {"value": 1.2336, "unit": "frames/s", "cores": 16, "kind": "port",
                             "sample": "4 iteration(s) of B=1 256x256 fwd+bwd (Lap/census loss) of the same model through oracle/"
                                       "model_ref.py + oracle/loss_ref.py; measured 1.2336 it/s, reported as 256x256 frames/s (x1.00 "
                                       "pixel-count scaling)"}}


def test_line_is_small_and_round_trips(tmp_path):
    full = full_record()
    assert len(json.dumps(full)) > 20000                      # (the record itself is as large as round 5's line)
    text = bench.compact_line(full, str(tmp_path / "bench_detail.json"))
    assert "\n" not in text and len(text) < 4096
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == 544.2 and d["config"]["workload"].startswith("EVFIAutoEx") and "model" not in d["config"]
    assert d["config"]["global_batch"] == 8 and d["config"]["untimed_steps"] == 5 and d["config"]["fp16_steps_skipped"] == 0
    rf = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rf) and rf["kernel"] == "conv_fwd_f16_ws"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 16
    assert d["dcn_fac_forward_frac_hbm"] == 0.447 and d["inference_frames_per_s"]["config5_hd_bf16x3"] == 126.7
    assert "kernels" not in d and "ops" not in d and "fp32_exact_mode" not in d


def test_line_sheds_optional_parts_rather_than_overflow():
    full = full_record()
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    d = json.loads(bench.compact_line(full))
    assert len(json.dumps(d)) < 4096 and d["roofline"]["kernel"] == "conv_fwd_f16_ws" and d["cpu_baseline"]["value"] == 1.2336


def test_detail_file_holds_the_full_record(tmp_path, capsys):
    full = full_record()
    path = bench.write_detail(full, str(tmp_path / "bench_detail.json"))
    assert path == str(tmp_path / "bench_detail.json")
    assert json.load(open(path))["kernels"].keys() == full["kernels"].keys()
    err = capsys.readouterr().err
    assert all(not ln.startswith("{") for ln in err.splitlines())      # nothing on stderr looks like the bench's JSON line
