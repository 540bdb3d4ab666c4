"""ebfi_amd.clipdata (the real-data entry) against a fixture produced by the reference's own H5Dataset
(tests/golden/make_golden_clipdata.py: dataloader/h5dataset.py:118-366 run on an in-memory clip).

CPU tests: period / exposure indexing, frame assembly, event slice + timestamp normalisation (binned here by the oracle's
events_to_stack, itself pinned bit-exactly to the reference function), crop window, rank sharding, the trainer's per-latent-frame
loop.  GPU test: the full item through the device events_to_stack kernel, bit for bit."""
import os

import numpy as np
import pytest
import torch

from ebfi_amd import clipdata

CFGS = {"fixed": dict(frames_per_period=8, frames_per_blurry=5, exposure_method="Fixed", exposure_time=[1], crop=None),
        "custom": dict(frames_per_period=6, frames_per_blurry=6, exposure_method="Custom", exposure_time=[3, 4, 6], crop=[16, 16]),
        "noise": dict(frames_per_period=8, frames_per_blurry=3, exposure_method="Fixed", exposure_time=[1], crop=[16, 24],
                      noise=(1.0, 0.05))}


@pytest.fixture(scope="module")
def fixture(golden_dir, tmp_path_factory):
    z = np.load(os.path.join(golden_dir, "clipdata_small.npz"))
    path = str(tmp_path_factory.mktemp("clip") / "clip0.npz")
    np.savez(path, **{k[5:]: z[k] for k in z.files if k.startswith("clip.")})
    return z, path


def _dataset(path, tag, device="cpu"):
    c = CFGS[tag]
    return clipdata.ClipDataset(path, time_bins=4, crop_mode="center", flips=False, device=device, **c)


@pytest.mark.parametrize("tag", ["fixed", "custom", "noise"])
def test_items_match_the_reference_dataset_on_the_host(fixture, tag):
    from oracle import events_ref
    z, path = fixture
    ds = _dataset(path, tag)
    assert len(ds) == int(z["%s.len" % tag]) > 1
    H, W = ds.clips[0].resolution
    for i in range(len(ds)):
        sharp, blur, (xs, ys, ts, ps), duty = ds.host_item(i)
        stack = torch.from_numpy(events_ref.events_to_stack(xs, ys, ts, ps.astype(np.float32), 4, (H, W))).transpose(0, 1)
        sharp, blur, stack = ds.augment([sharp, blur, stack], (H, W), seed=5)
        if ds.noise is not None:
            assert tag == "noise"
            clean = stack
            stack = clipdata.add_noise(stack[None], 5 + 3, *ds.noise)[0]
            assert 0 < (stack != clean).float().mean() <= 0.05        # (|N(0,1)| truncates to 0 for two thirds of the drawn cells)
        item = ds.assemble(sharp, blur, stack, duty)
        for k in ("SeqLatentF", "SeqBlurryF", "SeqHREv", "RelativeLatentTs", "SeqExposureDuty"):
            ref = z["%s.%d.%s" % (tag, i, k)]
            assert tuple(item[k].shape) == ref.shape, (k, item[k].shape, ref.shape)
            assert np.array_equal(item[k].numpy(), ref), (tag, i, k)


def test_period_rules():
    # the trailing period is dropped even when complete (candidates_indices[:-1], h5dataset.py:132-135)
    assert len(clipdata.period_items(32, 16, 16)) == 1 and len(clipdata.period_items(33, 16, 16)) == 2
    items = clipdata.period_items(40, 8, exposure_method="Custom", exposure_time=[3, 8])
    assert [len(b) for _, b, _ in items] == [3, 8, 3, 8] and [d for _, _, d in items] == [3 / 8, 1.0, 3 / 8, 1.0]
    assert items[2][0] == list(range(16, 24)) and items[2][1] == [16, 17, 18]
    with pytest.raises(AssertionError):
        clipdata.period_items(40, 8, exposure_method="Custom", exposure_time=[9])
    auto = clipdata.period_items(100, 10, exposure_method="Auto", seed=3)
    assert all(1 <= len(b) < 10 for _, b, _ in auto)


def test_sequence_items_follow_set_items():
    """h5dataset.py:166-186 by hand: 5 periods, sequences of 2 stepping by 2, loads of 1 stepping by 1 (scripts/infer_ours.sh)."""
    assert clipdata.sequence_items(5, 2, 2, 1, 1) == [[(0, 0), (1, 1)], [(2, 2), (3, 3)]]      # the start at 4 would end at 5 > 4
    assert clipdata.sequence_items(5, 1, 1, 1, 1) == [[(i, i)] for i in range(5)]
    assert clipdata.sequence_items(6, 3, 1, 2, 1) == [[(s, s + 1), (s + 1, s + 2)] for s in range(4)]   # the load at s+2 would cross
    assert clipdata.sequence_items(4, 2, 2, 2, 2) == [[(0, 1)], [(2, 3)]]
    assert clipdata.sequence_items(1, 2, 2, 1, 1) == []
    with pytest.raises(AssertionError):
        clipdata.sequence_items(4, 1, 1, 2, 1)
    with pytest.raises(ValueError):
        clipdata.sequence_items(4, 2, 0, 1, 1)


def test_event_normalisation_edge_cases():
    xs, ys, ts, ps = clipdata.normalise_events([], [], [], [])
    assert xs.tolist() == ys.tolist() == ts.tolist() == ps.tolist() == [0.0]
    _, _, ts, _ = clipdata.normalise_events([1, 2, 3], [0, 0, 0], [10.0, 10.5, 12.0], [1, -1, 1])
    assert ts[0] == 0.0 and ts[-1] == 2.0 / (2.0 + 1e-6) and ts.dtype == np.float64
    _, _, ts, _ = clipdata.normalise_events([1], [1], [7.0], [1])          # a single event: 0 / 1e-6
    assert ts.tolist() == [0.0]


def test_crop_windows():
    assert clipdata.crop_window(24, 32, (16, 16), "center") == (4, 8, 16, 16)
    assert clipdata.crop_window(24, 32, (24, 16), "center") is None and clipdata.crop_window(24, 32, (8, 32), "random", seed=1) is None
    a, b = clipdata.crop_window(64, 64, (32, 32), "random", seed=9), clipdata.crop_window(64, 64, (32, 32), "random", seed=9)
    assert a == b and 0 <= a[0] <= 32 and 0 <= a[1] <= 32
    assert clipdata.crop_window(64, 64, (32, 32), "random", scale=4, seed=9)[2:] == (8, 8)


def test_batches_shard_over_ranks_and_feed_the_trainer_loop(fixture):
    _, path = fixture

    class Host(clipdata.ClipDataset):           # the binning kernel needs a GPU: an all-zero stack stands in here
        def __getitem__(self, index, seed=None):
            sharp, blur, _, duty = self.host_item(index)
            return self.assemble(sharp, blur, torch.zeros(4, 2, *sharp.shape[-2:]), duty)

    ds = Host(path, time_bins=4, frames_per_period=4, frames_per_blurry=2, device="cpu")
    assert len(ds) == 6                          # 26 frames: period starts 0, 4, ..., 24, the trailing one dropped
    seen = []
    for rank in range(2):
        got = list(clipdata.batches(ds, 2, rank=rank, world=2, seed=1, epochs=1))
        assert len(got) == 1                     # 6 items -> 3 per rank -> one full batch of 2 each (drop_last)
        b = got[0]
        assert b["SeqLatentF"].shape == (2, 1, 1, 4, 3, 24, 32) and b["SeqHREv"].shape == (2, 1, 4, 2, 24, 32)
        passes = list(clipdata.model_inputs(b))
        assert len(passes) == 4
        for k, (frame, event, t, duty, latent) in enumerate(passes):
            assert frame.shape == (2, 3, 24, 32) and event.shape == (2, 4, 2, 24, 32) and t.shape == duty.shape == (2, 1)
            assert torch.equal(t, torch.full((2, 1), k / 4)) and torch.equal(duty, torch.full((2, 1), 0.5))
            assert torch.equal(latent, b["SeqLatentF"][:, 0, 0, k])
        seen.append(b["SeqBlurryF"])
    assert not torch.equal(seen[0], seen[1])     # different periods on the two ranks


def test_h5_clip_without_h5py_is_a_clear_error(tmp_path):
    try:
        import h5py  # noqa: F401
        pytest.skip("h5py is installed here")
    except ImportError:
        pass
    with pytest.raises(ImportError, match="needs h5py"):
        clipdata.open_clip(str(tmp_path / "x.h5"))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["fixed", "custom", "noise"])
def test_items_match_the_reference_dataset_on_the_device(fixture, tag):
    z, path = fixture
    ds = _dataset(path, tag, device="cuda")
    for i in range(len(ds)):
        item = ds.__getitem__(i, seed=5)
        for k in ("SeqLatentF", "SeqBlurryF", "SeqHREv", "RelativeLatentTs", "SeqExposureDuty"):
            assert item[k].is_cuda
            assert np.array_equal(item[k].cpu().numpy(), z["%s.%d.%s" % (tag, i, k)]), (tag, i, k)


@pytest.mark.gpu
def test_train_ours_runs_on_recorded_clips(tmp_path):
    """`train_ours.py --data <dir>`: two optimiser steps on a small synthetic clip directory (reduced-width model)."""
    import subprocess
    import sys
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for k in range(2):
        clipdata.write_synthetic_clip(str(tmp_path / ("clip%d.npz" % k)), num_imgs=17, H=32, W=32, events_per_frame=300, seed=k)
    cfg = yaml.safe_load(open(os.path.join(root, "ebfi-be_amd", "config", "train_ours.yml")))
    cfg["model"]["args"].update(FrameBasech=16, EventBasech=16, InterCH=16, TB=4, step=2, channels=[4, 4, 8, 8])
    cfg["trainer"].update(batch_size=2, output_path=str(tmp_path / "out"))
    cfg["train_dataloader"] = {"dataset": {"time_bins": 4, "NumFramePerPeriod": 4, "NumFramePerBlurry": 3, "ExposureMethod": "Fixed"}}
    cfg_path = str(tmp_path / "cfg.yml")
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    out = subprocess.run([sys.executable, os.path.join(root, "ebfi-be_amd", "train_ours.py"), "-c", cfg_path, "--data", str(tmp_path),
                          "--iterations", "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Iteration: 2/3" in out.stdout and "saved" in out.stdout


def test_dataset_config_keys_are_honoured_or_refused():
    """clipdata.dataset_args_from_config: the reference's dataset keys (config/train_ours.yml:115-150) either change what the
    reader does (flip probabilities, crop order) or are refused -- never ignored (round-4 advisory)."""
    import copy
    from ebfi_amd import clipdata
    # the `train_dataloader.dataset` section with the reference's key names and shipped values
    cfg = {"scale": 2, "ori_scale": "down2", "time_bins": 16, "NumFramePerPeriod": 16, "NumFramePerBlurry": 16,
           "ExposureMethod": "Custom", "ExposureTime": [9, 10, 11, 12, 13, 14, 15],
           "data_augment": {"enabled": True,
                            "augment": ["RandomCrop", "CenterCrop", "HorizontalFlip", "VertivcalFlip", "Noise", "HotPixel"],
                            "random_crop": {"enabled": True, "size": [128, 128]},
                            "center_crop": {"enabled": False, "size": [128, 128]},
                            "flip": {"enabled": True, "horizontal_prob": 0.5, "vertical_prob": 0.5},
                            "noise": {"enabled": False, "noise_std": 1.0, "noise_fraction": 0.05},
                            "hot_pixel": {"enabled": False, "hot_pixel_std": 2.0, "hot_pixel_fraction": 0.001}}}
    a = clipdata.dataset_args_from_config(cfg)
    assert a["crop"] == [128, 128] and a["crop_mode"] == "random" and a["center_crop"] is None
    assert a["flips"] is True and a["flip_probs"] == (0.5, 0.5)
    c = copy.deepcopy(cfg)
    c["data_augment"]["flip"].update(horizontal_prob=1.0, vertical_prob=0.0)
    c["data_augment"]["center_crop"].update(enabled=True, size=[64, 64])
    b = clipdata.dataset_args_from_config(c)
    assert b["flip_probs"] == (1.0, 0.0) and b["center_crop"] == [64, 64]
    c = copy.deepcopy(cfg)
    c["data_augment"]["noise"].update(enabled=True)
    assert clipdata.dataset_args_from_config(c)["noise"] == (1.0, 0.05) and a["noise"] is None
    for breaker in (lambda d: d["data_augment"].update(augment=["HorizontalFlip", "RandomCrop"]),
                    lambda d: d.update(scale=2, ori_scale="down4"),
                    lambda d: d.update(scale=1, ori_scale="down2")):
        c = copy.deepcopy(cfg)
        breaker(c)
        with pytest.raises(NotImplementedError):
            clipdata.dataset_args_from_config(c)
    c = copy.deepcopy(cfg)
    c.update(scale=1, ori_scale="ori")
    clipdata.dataset_args_from_config(c)
    c["data_augment"]["hot_pixel"]["enabled"] = True            # the reference never applies it either (h5dataset.py:436)
    clipdata.dataset_args_from_config(c)
    # flips with probability 1 / 0 are deterministic: horizontal always, vertical never
    t = torch.arange(24.0).reshape(1, 4, 6)
    ds = clipdata.ClipDataset.__new__(clipdata.ClipDataset)
    ds.crop, ds.crop_mode, ds.center_crop, ds.flips, ds.flip_probs = None, "random", None, True, (1.0, 0.0)
    for seed in range(5):
        assert torch.equal(ds.augment([t], (4, 6), seed)[0], t.flip(-1))
