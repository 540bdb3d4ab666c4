#!/usr/bin/env python3
"""Golden fixture for ebfi_amd.clipdata: RUNS THE REFERENCE'S OWN H5Dataset (/root/reference/dataloader/h5dataset.py;
build container only) on a small random clip and stores what `__getitem__` returned.

    python tests/golden/make_golden_clipdata.py        # rewrites tests/golden/clipdata_small.npz

How the reference class is run without HDF5: `h5py` is absent from the image and the dataset only uses the open file as a
nested mapping (`f['ori_images'][name][:]`, `.attrs[...]`, `f['ori_events/xs'][a:b]`), so the module is imported with
empty placeholder modules for h5py / cv2 (never called: frames already have the sensor resolution) and the instance gets a
plain in-memory mapping of numpy arrays in place of the open file.  Every index rule, the event normalisation, the
timestamp arithmetic, the crop and the (CPU) events_to_stack are the reference's code.  Only data is written.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ebfi-be_amd"))


class _Attrs(dict):
    pass


class _Arr:
    """numpy array with an `.attrs` mapping (what an h5py dataset looks like to the reference code)."""

    def __init__(self, a, **attrs):
        self.a, self.attrs = a, _Attrs(attrs)

    def __getitem__(self, k):
        return self.a[k]


class _Group(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.attrs = _Attrs()

    def __getitem__(self, key):
        node = self
        for part in key.split("/"):
            node = dict.__getitem__(node, part)
        return node


def as_file(clip):
    f = _Group()
    H, W = clip["images"].shape[1:3]
    f.attrs["sensor_resolution"] = np.array([H, W])
    f["ori_images"] = _Group({"image%09d" % i: _Arr(clip["images"][i], ori_event_idx=int(clip["event_idx"][i]))
                              for i in range(len(clip["images"]))})
    f["ori_events"] = _Group({k: _Arr(clip[k]) for k in ("xs", "ys", "ts", "ps")})
    return f


def import_h5dataset():
    for n in ("h5py", "cv2"):
        sys.modules[n] = types.ModuleType(n)
    import matplotlib.pyplot as plt
    orig = plt.style.use

    def tolerant(style):               # 'seaborn-whitegrid' left matplotlib in 3.8
        try:
            orig(style)
        except Exception:
            pass
    plt.style.use = tolerant
    # the reference's `dataloader` package, loaded by file path (this repo ships an import-path shim of the same name)
    import importlib.util
    for name in ("dataloader", "dataloader.encodings", "dataloader.h5dataset"):
        sys.modules.pop(name, None)
    pkg = types.ModuleType("dataloader")
    pkg.__path__ = [os.path.join(REF, "dataloader")]
    sys.modules["dataloader"] = pkg
    mods = {}
    for name in ("encodings", "h5dataset"):
        spec = importlib.util.spec_from_file_location("dataloader." + name, os.path.join(REF, "dataloader", name + ".py"))
        mods[name] = importlib.util.module_from_spec(spec)
        sys.modules["dataloader." + name] = mods[name]
        spec.loader.exec_module(mods[name])
    return mods["h5dataset"]


def main():
    from ebfi_amd import clipdata
    h5d = import_h5dataset()
    path = os.path.join(HERE, "_clip_tmp.npz")
    clipdata.write_synthetic_clip(path, num_imgs=26, H=24, W=32, events_per_frame=60, seed=11)
    clip = dict(np.load(path))
    os.remove(path)
    # a gap without events between two frames, and a period whose event slice is empty, are part of the fixture
    out = {"clip." + k: v for k, v in clip.items()}
    cfgs = {
        "fixed": dict(NumFramePerPeriod=8, NumFramePerBlurry=5, ExposureMethod="Fixed", ExposureTime=[1], crop=None),
        "custom": dict(NumFramePerPeriod=6, NumFramePerBlurry=6, ExposureMethod="Custom", ExposureTime=[3, 4, 6], crop=[16, 16]),
        # event-stack noise (AugmentData 'Noise' -> add_noise, h5dataset.py:432-433, :455-463) after a centre crop: what
        # infer_ours.py runs with unless --noise_enabled is given (the flag is store_false there)
        "noise": dict(NumFramePerPeriod=8, NumFramePerBlurry=3, ExposureMethod="Fixed", ExposureTime=[1], crop=[16, 24],
                      noise=dict(enabled=True, noise_std=1.0, noise_fraction=0.05)),
    }
    for tag, c in cfgs.items():
        config = dict(scale=1, ori_scale="ori", time_bins=4, NumFramePerPeriod=c["NumFramePerPeriod"],
                      NumFramePerBlurry=c["NumFramePerBlurry"], NumPeriodPerSeq=1, SlidingWindowSeq=1, NumPeriodPerLoad=1,
                      SlidingWindowLoad=1, ExposureMethod=c["ExposureMethod"], ExposureTime=c["ExposureTime"],
                      data_augment=dict(enabled=c["crop"] is not None, augment=["CenterCrop", "Noise"],
                                        center_crop=dict(enabled=True, size=c["crop"] or [0, 0]),
                                        noise=c.get("noise", dict(enabled=False))))
        ds = h5d.H5Dataset.__new__(h5d.H5Dataset)
        ds.config, ds.h5_file_path = config, "<memory>"
        ds.h5_file = as_file(clip)
        ds.sensor_resolution = ds.h5_file.attrs["sensor_resolution"].tolist()
        ds.scale, ds.ori_scale = 1, "ori"
        # (a tuple: GetFrames compares `frame.shape[:-1] != self.gt_sensor_resolution` -- with the list set_data_scale stores
        # that is always true and every frame goes through a same-size cv2.resize, an identity OpenCV is not here to perform)
        ds.inp_sensor_resolution = ds.gt_sensor_resolution = tuple(ds.sensor_resolution)
        ds.inp_prex = ds.gt_prex = "ori"
        ds.load_metadata()
        ds.set_period_items()
        ds.set_items()
        out["%s.len" % tag] = np.array(len(ds))
        for i in range(len(ds)):
            item = ds.__getitem__(i, seed=5)
            for k in ("SeqLatentF", "SeqBlurryF", "SeqHREv", "RelativeLatentTs", "SeqExposureDuty"):
                out["%s.%d.%s" % (tag, i, k)] = item[k].numpy()
    np.savez_compressed(os.path.join(HERE, "clipdata_small.npz"), **out)
    print("wrote clipdata_small.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
