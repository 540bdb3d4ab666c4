"""infer_ours.py with the reference's command line (/root/reference/infer_ours.py:193-220, scripts/infer_ours.sh:2-16).

CPU: the flag set and the defaults / overrides of the dataset section; GPU: the script run end to end on the clip of the
reference-generated clipdata fixture, the restored frames it writes compared with the CPU oracle at 1e-3; and BASELINE config 2
(B=4 256x256, fp32) through ClipInterpolator, one sample against the oracle."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ebfi-be_amd")

# scripts/infer_ours.sh of the reference, first block, argument for argument
REFERENCE_ARGS = ["--model_path", "/path/to/model", "--data_list", "/path/to/test.txt", "--output_path", "/path/to/output",
                  "--scale", "2", "--ori_scale", "down2", "--time_bins", "16", "--num_frame_per_period", "16",
                  "--num_frame_per_blurry", "3", "--num_period_per_seq", "2", "--sliding_window_seq", "2",
                  "--num_period_per_load", "1", "--sliding_window_load", "1", "--exposure_method", "Fixed", "--noise_enabled"]
# ... and its RealBlur block
REFERENCE_ARGS_REAL = ["--model_path", "/path/to/model", "--data_list", "/path/to/test.txt", "--output_path", "/path/to/output",
                       "--scale", "2", "--ori_scale", "down2", "--time_bins", "16", "--interp_num", "256", "--num_period_per_seq", "2",
                       "--sliding_window_seq", "2", "--num_period_per_load", "1", "--sliding_window_load", "1", "--noise_enabled",
                       "--real_blur"]


@pytest.fixture(scope="module")
def cli():
    spec = importlib.util.spec_from_file_location("ebfi_infer_ours", os.path.join(PKG, "infer_ours.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_command_line_is_accepted(cli):
    f = cli.get_flags(REFERENCE_ARGS)
    ds, notes = cli.dataset_settings(f)
    assert notes == []
    assert (ds["scale"], ds["ori_scale"], ds["time_bins"]) == (2, "down2", 16)
    assert (ds["NumFramePerPeriod"], ds["NumFramePerBlurry"], ds["ExposureMethod"]) == (16, 3, "Fixed")
    assert (ds["NumPeriodPerSeq"], ds["SlidingWindowSeq"], ds["NumPeriodPerLoad"], ds["SlidingWindowLoad"]) == (2, 2, 1, 1)
    assert ds["noise"]["enabled"] is False            # --noise_enabled is store_false in the reference: the flag switches noise OFF
    assert f.device == "cuda:0" and f.center_crop_size is None
    f = cli.get_flags(REFERENCE_ARGS_REAL)
    ds, notes = cli.dataset_settings(f)
    assert f.real_blur and any("real_blur" in n for n in notes)


def test_reference_defaults_and_overrides(cli):
    ds, notes = cli.dataset_settings(cli.get_flags(["--data_list", "x.txt", "--output_path", "o"]))
    # infer_ours.py:223-236: down4 / 16 frames per period / 9 per blurry / noise on with std 1.0 at 5 % of the cells
    assert (ds["scale"], ds["ori_scale"], ds["NumFramePerPeriod"], ds["NumFramePerBlurry"]) == (4, "down4", 16, 9)
    assert ds["noise"] == dict(enabled=True, noise_std=1.0, noise_fraction=0.05)
    assert ds["NumPeriodPerLoad"] == 1 and any("num_period_per_load" in n for n in notes)     # its default of 2 cannot feed the model
    ds, notes = cli.dataset_settings(cli.get_flags(["--data_list", "x.txt", "--output_path", "o", "--noise_std", "2.5",
                                                    "--center_crop_size", "128", "160", "--exposure_method", "Custom",
                                                    "--exposure_time", "9 11 13", "--scale", "2", "--ori_scale", "down4",
                                                    "--num_period_per_load", "1"]))
    assert ds["noise"] == dict(enabled=True, noise_std=2.5, noise_fraction=0.05) and ds["center_crop"] == [128, 160]
    assert ds["ExposureTime"] == [9, 11, 13] and len(notes) == 1 and "ori_scale" in notes[0]


def _small_checkpoint(tmp_path, cfg):
    from ebfi_amd.model import EVFIAutoEx
    torch.manual_seed(3)
    net = EVFIAutoEx(**cfg)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() > 1:
                p.copy_(torch.randn_like(p) * (1.2 / p[0].numel() ** 0.5))
            else:
                p.add_(0.05 * torch.randn_like(p))
    path = str(tmp_path / "checkpoint-iteration9.pth")
    torch.save({"model": {"name": "EVFIAutoEx", "states": net.state_dict()}, "config": {"model": {"name": "EVFIAutoEx", "args": cfg}},
                "trainer": {"training_mode": "iteration_based_train", "iteration": 9, "monitor_best": None}}, path)
    return path, {k: v.clone() for k, v in net.state_dict().items()}


@pytest.mark.gpu
@pytest.mark.parametrize("noise", [False, True])
def test_infer_ours_writes_what_the_oracle_computes(cli, golden_dir, tmp_path, noise):
    """The reference's flags on the fixture clip (3 periods of 8 frames at 24x32, 4 event bins): sequences of 2 loads, 8 latent
    timestamps per load, blurry = mean of the first 3 frames.  Every restored frame in restored.npz equals the CPU oracle's
    `Final` for the same (Frame, Event, T, GTEx) within 1e-3, with and without the dataset's event noise."""
    from oracle import model_ref
    from ebfi_amd import clipdata
    from ebfi_amd.engine import DEFAULT_MODEL_ARGS
    z = np.load(os.path.join(golden_dir, "clipdata_small.npz"))
    clip = str(tmp_path / "clip0.npz")
    np.savez(clip, **{k[5:]: z[k] for k in z.files if k.startswith("clip.")})
    lst = str(tmp_path / "test.txt")
    open(lst, "w").write(clip + "\n")
    cfg = dict(DEFAULT_MODEL_ARGS, FrameBasech=16, EventBasech=16, InterCH=16, TB=4, step=2, channels=[4, 4, 8, 8])
    ckpt, sd = _small_checkpoint(tmp_path, cfg)
    out = str(tmp_path / "out")
    args = ["--model_path", ckpt, "--data_list", lst, "--output_path", out, "--scale", "1", "--ori_scale", "ori", "--time_bins", "4",
            "--num_frame_per_period", "8", "--num_frame_per_blurry", "3", "--num_period_per_seq", "2", "--sliding_window_seq", "2",
            "--num_period_per_load", "1", "--sliding_window_load", "1", "--exposure_method", "Fixed", "--png"]
    if not noise:
        args.append("--noise_enabled")
    cli.main(args)
    res = np.load(os.path.join(out, "clip0.npz", "restored.npz"))
    assert res["restored"].shape == (2, 8, 3, 24, 32) and res["period"].tolist() == [0, 1]      # the third period: no full sequence
    assert np.allclose(res["exposure_duty"], 3 / 8) and np.array_equal(res["timestamps"][0], np.arange(8) / 8)
    img = os.path.join(out, "clip0.npz", "img")
    assert sorted(os.listdir(os.path.join(img, "restored_frame")))[:2] == ["000000000_0.png", "000000001_0.png"]
    assert len(os.listdir(os.path.join(img, "restored_frame"))) == 16 and len(os.listdir(os.path.join(img, "blurry_frame"))) == 2
    # the same items, rebuilt here with the seeds the script uses, through the CPU oracle
    data = clipdata.ClipDataset(clip, time_bins=4, frames_per_period=8, frames_per_blurry=3, exposure_method="Fixed", crop=None,
                                crop_mode="center", device="cuda", seed=123, noise=(1.0, 0.05) if noise else None)
    worst = 0.0
    for load, period in enumerate((0, 1)):
        item = data.__getitem__(period, seed=123 + 7919 * 0 + period)
        frame, event, duty = item["SeqBlurryF"][0].cpu(), item["SeqHREv"].cpu(), item["SeqExposureDuty"][0].cpu()
        assert np.array_equal(res["blurry"][load], frame[0].numpy())
        if noise:
            assert event.sum() > torch.from_numpy(z["fixed.%d.SeqHREv" % period]).sum()       # (noise only adds counts)
        for k in (0, 3, 7):
            t = torch.full((1, 1), k / 8)
            ref = model_ref.evfi_forward(sd, cfg, frame, event, t, duty)[-1]
            err = (torch.from_numpy(res["restored"][load, k]) - ref[0]).abs().max().item() / ref.abs().max().item()
            worst = max(worst, err)
            assert err < 1e-3, (load, k, err)
    assert res["restored"].std() > 1e-3
    # a second run into the same output directory refuses to overwrite, like the reference's os.makedirs(exist_ok=False)
    with pytest.raises(FileExistsError):
        cli.main(args)


@pytest.mark.gpu
def test_config2_clip_interpolator_vs_oracle():
    """BASELINE config 2 -- B=4 256x256 fp32 inference -- through the path infer_ours.py runs (ClipInterpolator: hoisted prefix +
    hipGraph replays, default widths): sample 2 of the batch at two timestamps against the CPU oracle at 1e-3, and the
    split-precision default against the fp32 result."""
    from oracle import model_ref
    from ebfi_amd.engine import DEFAULT_MODEL_ARGS, ClipInterpolator, synthetic_batch
    from ebfi_amd.model import EVFIAutoEx
    torch.manual_seed(11)
    net = EVFIAutoEx(**DEFAULT_MODEL_ARGS)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() > 1:
                p.copy_(torch.randn_like(p) * (1.2 / p[0].numel() ** 0.5))
            else:
                p.add_(0.05 * torch.randn_like(p))
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda().eval()
    frame, event, _, gtex, _ = synthetic_batch(4, 256, 256, 16, device="cuda", seed=21)
    stamps = [0.125, 0.6875]
    got = {p: ClipInterpolator(net, precision=p, graph=True, hoist=True)(frame, event, gtex, stamps) for p in ("fp32", "bf16x3")}
    assert got["fp32"].shape == (4, 2, 3, 256, 256)
    rel = lambda a, b: ((a.float().cpu() - b).abs().max() / b.abs().max()).item()
    for i, ts in enumerate(stamps):
        ref = model_ref.evfi_forward(sd, DEFAULT_MODEL_ARGS, frame[2:3].cpu(), event[2:3].cpu(), torch.full((1, 1), ts), gtex[2:3].cpu())[-1]
        assert ref.std() > 0.01
        for p in ("fp32", "bf16x3"):
            assert rel(got[p][2:3, i], ref) < 1e-3, (p, ts, rel(got[p][2:3, i], ref))
