"""Real-data entry of the hot path: recorded clips (frames + a raw event list) -> the tensors the model is fed.

What the reference's H5Dataset does between the file and the network (dataloader/h5dataset.py), minus HDF5 itself:

  * periods of `NumFramePerPeriod` consecutive sharp frames; the blurry input of a period is the MEAN of its first
    `exposure` frames and `ExposureDuty = exposure / NumFramePerPeriod` (set_period_items, h5dataset.py:118-166:
    Fixed / Custom exposure; 'Auto' draws the exposure with numpy's global generator there and with a seeded one here),
  * the events between the first and the last latent frame of the period, timestamps normalised to
    `(t - t0) / (tN - t0 + 1e-6)` (GetEventsIndex, :327-336; an empty slice becomes the single all-zero event), binned by
    `events_to_stack(..., B=time_bins)` and transposed to [TB, 2, H, W] (:349) -- here by the DEVICE kernel
    (ebfi_amd.encodings, bit-exact with the reference function),
  * `RelativeLatentTs[k] = k / NumFramePerPeriod` for the k-th latent frame (GetTimestamp, :354-366, NumPeriodPerLoad = 1),
  * frames stored BGR uint8 [H, W, 3], returned RGB float / 255 (GetFrames, :296-311),
  * crop (random, seeded / centre, both snapped to `scale` like AugmentData :368-411 with scale = 1) and the two flips,
    applied identically to frames and event stacks.

One item = one period (NumPeriodPerLoad = NumPeriodPerSeq = 1, what config/train_ours.yml trains with); the
trainer then runs one optimiser pass per latent frame of the batch exactly like train_ours.py:237-251.

Storage.  A clip is either an `.npz` file with

    images      uint8 [N, H, W, 3]   BGR, like ori_images/image%09d
    event_idx   int64 [N]            index of the first event at / after frame i (the image attribute `<prex>_event_idx`)
    xs, ys      int16 / any [E]      pixel coordinates
    ts          float64 [E]          seconds, sorted
    ps          int8 / any [E]       polarity +-1

or, when `h5py` is importable (it is not part of the MI355X image: the import is optional and a missing module raises a
clear error only when an .h5 file is actually opened), an HDF5 file in the reference's own layout
(`ori_images/image%09d` + attrs `ori_event_idx`, `ori_events/{xs,ys,ts,ps}`), read at scale 1 ('ori').
"""
import os
import random

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------ host logic (pure numpy)
def period_items(num_imgs, frames_per_period, frames_per_blurry=None, exposure_method="Fixed", exposure_time=None, seed=0):
    """h5dataset.py:118-166 -> list of (latent_indices, blurry_indices, exposure_duty).  The last, possibly incomplete period
    is dropped exactly like `candidates_indices[:-1]` does there (also when num_imgs is a multiple of the period)."""
    P = int(frames_per_period)
    assert P >= 1, "Number of frames per period must >= 1!"
    assert exposure_method in ("Fixed", "Auto", "Custom"), "Error exposure setting!"
    rng = np.random.RandomState(seed)
    starts = np.arange(0, int(num_imgs), P)[:-1]
    items = []
    for j, idx in enumerate(starts):
        if exposure_method == "Fixed":
            e = int(frames_per_blurry)
            assert 1 <= e <= P, "Number of frames per blurry must be in [1, frames per period]!"
        elif exposure_method == "Auto":
            e = int(rng.randint(1, P)) if P > 1 else 1
        else:
            e = int(exposure_time[j % len(exposure_time)])
            assert e <= P, "Number of frames per blurry must <= Number of frames per period!"
        items.append(([int(idx) + i for i in range(P)], [int(idx) + i for i in range(e)], e / P))
    return items


def sequence_items(num_periods, periods_per_seq=1, sliding_window_seq=1, periods_per_load=1, sliding_window_load=1):
    """set_items (h5dataset.py:166-186): the dataset's items are SEQUENCES of loads, a load = [first period, last period].
    Sequence starts step by `sliding_window_seq`; a sequence that would run past the last period is dropped; inside a sequence
    loads start every `sliding_window_load` periods and a load that would cross the sequence's end is dropped.  Returns the
    list of sequences, each a list of (left, right) period indices -- infer_ours.py walks them in this order."""
    S, ws, Lp, wl = int(periods_per_seq), int(sliding_window_seq), int(periods_per_load), int(sliding_window_load)
    assert S >= 1, "Number of period per seq must >= 1!"
    assert 0 <= ws <= S, "Sliding window seq must be in [0, number of period per seq]"
    assert Lp >= 1, "Number of period per Load must >= 1!"
    assert 0 <= wl <= Lp, "Sliding window Load must be in [0, number of period per Load]"
    assert Lp <= S, "Number of period per load must <= Number of period per seq"
    if ws == 0 or wl == 0:
        raise ValueError("a sliding window of 0 never advances (numpy.arange raises in the reference too)")
    seqs = []
    for start in range(0, int(num_periods), ws):
        end = start + S - 1
        if end <= num_periods - 1:
            seqs.append([(i, i + Lp - 1) for i in range(start, end + 1, wl) if i + Lp - 1 <= end])
    return seqs


def add_noise(data, seed, noise_std=1.0, noise_fraction=0.1):
    """add_noise of the reference (h5dataset.py:455-463) on an event stack: `|N(0, std)|` truncated to an integer at a
    `noise_fraction` of the cells, drawn on the HOST from a generator seeded like the reference seeds torch's global one
    (`torch.manual_seed(seed)`; same engine, same draw order: one normal and one uniform per cell over the contiguous shape) --
    bit-identical counts for the same seed (tests/test_clipdata.py, fixture case 'noise').  The sum is formed on data's device."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    noise = (noise_std * torch.randn(tuple(data.shape), generator=g)).abs().int()
    if noise_fraction < 1.0:
        noise.masked_fill_(torch.rand(tuple(data.shape), generator=g) >= noise_fraction, 0)
    return data + noise.to(data.device)


def normalise_events(xs, ys, ts, ps):
    """GetEventsIndex (h5dataset.py:327-336): an empty slice becomes the single event (0, 0, 0, 0); timestamps become
    (t - t0) / (tN - t0 + 1e-6).  Returns float64 arrays (the reference concatenates into one float64 [4, N] tensor)."""
    xs, ys, ts, ps = (np.asarray(v) for v in (xs, ys, ts, ps))
    if len(xs) == 0 or len(ys) == 0 or len(ts) == 0 or len(ps) == 0:
        xs = ys = ts = ps = np.array([0.0])
    ts = ts.astype(np.float64)
    ts = (ts - ts[0]) / (ts[-1] - ts[0] + 1e-6)
    return xs.astype(np.float64), ys.astype(np.float64), ts, ps.astype(np.float64)


def crop_window(h, w, size, mode, scale=1, seed=None):
    """(i, j, th, tw) of AugmentData's random_crop / center_crop (h5dataset.py:369-411); None when the crop is larger than
    the frame (the reference then returns the data unchanged)."""
    th, tw = int(size[0]), int(size[1])
    if th >= h or tw >= w:
        return None
    if mode == "random":
        r = random.Random(seed)
        i, j = r.randint(0, h - th), r.randint(0, w - tw)
    else:
        i, j = int((h - th) / 2), int((w - tw) / 2)
    i, j = int(i // scale) * scale, int(j // scale) * scale
    return i // scale, j // scale, th // scale, tw // scale


# ------------------------------------------------------------------------------------------------ storage
class _NpzClip:
    def __init__(self, path):
        z = np.load(path)
        self.images = z["images"]
        self.event_idx = np.asarray(z["event_idx"]).astype(np.int64)
        self.xs, self.ys, self.ts, self.ps = z["xs"], z["ys"], z["ts"], z["ps"]
        if self.images.ndim != 4 or self.images.shape[-1] != 3 or len(self.event_idx) != len(self.images):
            raise ValueError("%s: images must be [N,H,W,3] with one event_idx per image" % path)
        self.num_imgs = int(self.images.shape[0])
        self.resolution = (int(self.images.shape[1]), int(self.images.shape[2]))

    def frame_bgr(self, i):
        return self.images[i]

    def events(self, i0, i1):
        a, b = int(self.event_idx[i0]), int(self.event_idx[i1])
        return self.xs[a:b], self.ys[a:b], self.ts[a:b], self.ps[a:b]


class _H5Clip:
    """The reference's file layout at scale 1 (h5dataset.py:31-40, :296-347)."""

    def __init__(self, path):
        try:
            import h5py
        except ImportError as e:          # (absent from the MI355X image)
            raise ImportError("reading %s needs h5py, which is not installed; convert the clip to .npz "
                              "(ebfi_amd.clipdata module docstring)" % path) from e
        self.f = h5py.File(path, "r")
        self.num_imgs = len(self.f["ori_images"].keys())
        self.resolution = tuple(int(v) for v in self.f.attrs["sensor_resolution"].tolist())

    def frame_bgr(self, i):
        return self.f["ori_images"]["image%09d" % i][:]

    def events(self, i0, i1):
        a = self.f["ori_images"]["image%09d" % i0].attrs["ori_event_idx"]
        b = self.f["ori_images"]["image%09d" % i1].attrs["ori_event_idx"]
        g = self.f["ori_events"]
        return g["xs"][a:b], g["ys"][a:b], g["ts"][a:b], g["ps"][a:b]


def open_clip(path):
    return _H5Clip(path) if path.endswith((".h5", ".hdf5")) else _NpzClip(path)


def list_clips(path):
    """A directory (every .npz / .h5 in it, sorted), a text file with one clip path per line (the reference's datalist.txt),
    or one clip file."""
    if os.path.isdir(path):
        return sorted(os.path.join(path, f) for f in os.listdir(path) if f.endswith((".npz", ".h5", ".hdf5")))
    if path.endswith(".txt"):
        with open(path) as fh:
            return [ln.strip() for ln in fh if ln.strip()]
    return [path]


# ------------------------------------------------------------------------------------------------ dataset
class ClipDataset:
    """Items in the reference's key names and shapes for L = NumPeriodPerLoad = 1 (h5dataset.py:283-295):
        SeqLatentF [1, 1, NumF, 3, H, W]   SeqBlurryF [1, 1, 3, H, W]   SeqHREv [1, TB, 2, H, W]
        RelativeLatentTs [1, 1, NumF]      SeqExposureDuty [1, 1, 1]
    Frames are produced on the host and moved to `device`; the event stack is binned on the device."""

    def __init__(self, paths, time_bins=16, frames_per_period=16, frames_per_blurry=16, exposure_method="Fixed",
                 exposure_time=None, crop=None, crop_mode="random", flips=False, device="cuda", seed=0,
                 flip_probs=(0.5, 0.5), center_crop=None, noise=None):
        """crop / crop_mode: the first crop of AugmentData's list (RandomCrop when enabled, else CenterCrop); center_crop: a
        CenterCrop applied AFTER a random crop when the config enables both (the reference walks its `augment` list in order,
        h5dataset.py:410-433); flip_probs: (horizontal_prob, vertical_prob) of data_augment.flip; noise: None or
        (noise_std, noise_fraction) of data_augment.noise -- applied to the event stack after crops and flips with the item's
        seed + 3, like AugmentData does ('Noise' sits behind the crops and flips in the shipped `augment` order)."""
        self.clips = [open_clip(p) for p in (list_clips(paths) if isinstance(paths, str) else list(paths))]
        if not self.clips:
            raise ValueError("no clips under %r" % (paths,))
        self.time_bins, self.P = int(time_bins), int(frames_per_period)
        self.crop, self.crop_mode, self.flips = crop, crop_mode, bool(flips)
        self.flip_probs, self.center_crop = (float(flip_probs[0]), float(flip_probs[1])), center_crop
        self.noise = None if noise is None else (float(noise[0]), float(noise[1]))
        self.device = torch.device(device)
        self.items = []
        for ci, clip in enumerate(self.clips):
            for it in period_items(clip.num_imgs, frames_per_period, frames_per_blurry, exposure_method, exposure_time, seed + ci):
                self.items.append((ci, it))

    def __len__(self):
        return len(self.items)

    def event_list(self, index):
        """The period's normalised event list (host arrays) -- what GetEventsIndex returns."""
        ci, (latent, _, _) = self.items[index]
        return normalise_events(*self.clips[ci].events(latent[0], latent[-1]))

    def host_item(self, index):
        """Everything of an item that is host work: (sharp [NumF,3,H,W], blurry [3,H,W], normalised event list, duty)."""
        ci, (latent, blurry, duty) = self.items[index]
        clip = self.clips[ci]
        rgb = lambda i: np.ascontiguousarray(clip.frame_bgr(i)[:, :, ::-1])                       # BGR -> RGB
        sharp = torch.from_numpy(np.stack([rgb(i) for i in latent])).permute(0, 3, 1, 2).float() / 255      # [NumF,3,H,W]
        blur = torch.from_numpy(np.stack([rgb(i) for i in blurry]).mean(0)).permute(2, 0, 1).float() / 255   # [3,H,W]
        return sharp, blur, self.event_list(index), duty

    def augment(self, tensors, resolution, seed):
        """Crop and flips of AugmentData, the same window / decision for every tensor of the item."""
        H, W = resolution
        if self.crop is not None:
            win = crop_window(H, W, self.crop, self.crop_mode, 1, seed + 2)
            if win is not None:
                i, j, th, tw = win
                tensors = [v[..., i:i + th, j:j + tw] for v in tensors]
                H, W = th, tw
        if self.center_crop is not None:
            win = crop_window(H, W, self.center_crop, "center", 1, seed + 2)
            if win is not None:
                i, j, th, tw = win
                tensors = [v[..., i:i + th, j:j + tw] for v in tensors]
        if self.flips:
            if random.Random(seed).random() < self.flip_probs[0]:
                tensors = [v.flip(-1) for v in tensors]
            if random.Random(seed + 1).random() < self.flip_probs[1]:
                tensors = [v.flip(-2) for v in tensors]
        return tensors

    def assemble(self, sharp, blur, stack, duty):
        dev = stack.device
        rel_ts = (torch.arange(self.P) / self.P).to(dev)      # integer tensor / int on the HOST, like GetTimestamp (a device
        #                                                       division may differ in the last bit)
        return {"SeqLatentF": sharp[None, None].contiguous(), "SeqBlurryF": blur[None, None].contiguous(),
                "SeqHREv": stack[None].contiguous(), "RelativeLatentTs": rel_ts[None, None],
                "SeqExposureDuty": torch.tensor([[[duty]]], dtype=torch.float32, device=dev)}

    def __getitem__(self, index, seed=None):
        from .encodings import events_to_stack
        if seed is None:
            seed = random.randint(0, 2 ** 32)
        sharp, blur, (xs, ys, ts, ps), duty = self.host_item(index)
        res = self.clips[self.items[index][0]].resolution
        dev = self.device
        to = lambda a, dt: torch.from_numpy(a).to(dev, dt)
        stack = events_to_stack(to(xs, torch.float64), to(ys, torch.float64), to(ts, torch.float64), to(ps, torch.float32),
                                self.time_bins, sensor_size=res).transpose(0, 1)                # [TB,2,H,W]
        sharp, blur, stack = self.augment([sharp.to(dev), blur.to(dev), stack], res, seed)
        if self.noise is not None:
            stack = add_noise(stack[None], seed + 3, *self.noise)[0]          # (drawn over [L=1, TB, 2, H, W] like the reference)
        return self.assemble(sharp, blur, stack, duty)


SUPPORTED_AUGMENT_ORDER = ["RandomCrop", "CenterCrop", "HorizontalFlip", "VertivcalFlip", "Noise", "HotPixel"]


def dataset_args_from_config(ds_cfg):
    """The reference's `train_dataloader.dataset` keys (config/train_ours.yml:115-150) -> ClipDataset keyword arguments.
    Everything the keys can ask for that this reader does NOT do is refused here instead of being ignored silently
    (round-4 advisory): another `augment` order than the shipped one (crops, then flips, then noise) and scale / ori_scale pairs whose ground truth is not the 'ori' groups of a clip.  hot_pixel needs no code: the
    reference's own test `type == [...]` (h5dataset.py:436) is never true, so it never adds hot pixels."""
    aug = ds_cfg.get("data_augment") or {}
    out = dict(crop=None, crop_mode="random", center_crop=None, flips=False, flip_probs=(0.5, 0.5), noise=None)
    # the tensors the model sees come from the reference's `gt_prex` groups (h5dataset.py:36-100): 'ori' exactly when `scale`
    # equals the factor `ori_scale` names -- the shipped pair (2, 'down2') and (1, 'ori') among them; any other pair reads
    # down-scaled groups this reader does not open
    scale, ori = int(ds_cfg.get("scale", 2)), str(ds_cfg.get("ori_scale", "down2"))
    factor = {"ori": 1, "down2": 2, "down4": 4, "down8": 8, "down16": 16}.get(ori)
    if factor is None or scale != factor:
        raise NotImplementedError("dataset.scale %r with ori_scale %r selects the reference's %s ground-truth groups; this reader "
                                  "opens the 'ori' groups only (scale == the factor ori_scale names)" % (scale, ori, "down-scaled"))
    if not aug.get("enabled", False):
        return out
    order = list(aug.get("augment") or SUPPORTED_AUGMENT_ORDER)
    known = [m for m in order if m in SUPPORTED_AUGMENT_ORDER]
    if known != [m for m in SUPPORTED_AUGMENT_ORDER if m in known] or len(known) != len(order):
        raise NotImplementedError("data_augment.augment %r: supported is the shipped order %r (crops before flips) or a "
                                  "sub-list of it" % (order, SUPPORTED_AUGMENT_ORDER))
    nz = aug.get("noise") or {}
    if nz.get("enabled") and "Noise" in order:          # (h5dataset.py:432-433: defaults of add_noise where a key is absent)
        out["noise"] = (float(nz.get("noise_std", 1.0)), float(nz.get("noise_fraction", 0.1)))
    rc, cc = aug.get("random_crop") or {}, aug.get("center_crop") or {}
    if rc.get("enabled") and "RandomCrop" in order:
        out["crop"], out["crop_mode"] = rc["size"], "random"
        if cc.get("enabled") and "CenterCrop" in order:
            out["center_crop"] = cc["size"]
    elif cc.get("enabled") and "CenterCrop" in order:
        out["crop"], out["crop_mode"] = cc["size"], "center"
    fl = aug.get("flip") or {}
    if fl.get("enabled"):
        ph = float(fl.get("horizontal_prob", 0.5)) if "HorizontalFlip" in order else 0.0
        pv = float(fl.get("vertical_prob", 0.5)) if "VertivcalFlip" in order else 0.0
        out["flips"], out["flip_probs"] = True, (ph, pv)
    return out


def collate(samples):
    return {k: torch.stack([s[k] for s in samples]) for k in samples[0]}


def batches(dataset, batch_size, rank=0, world=1, seed=0, epochs=None, shuffle=True, drop_last=True):
    """Per-rank batches: a seeded permutation per epoch split round-robin over ranks (what DistributedSampler does,
    h5dataloader.py:47-57); yields collated dicts [B, L=1, ...]."""
    epoch = 0
    while epochs is None or epoch < epochs:
        order = list(range(len(dataset)))
        if shuffle:
            random.Random(seed + epoch).shuffle(order)
        order = order[rank::world]
        for k in range(0, len(order), batch_size):
            idx = order[k:k + batch_size]
            if len(idx) < batch_size and drop_last:
                break
            yield collate([dataset.__getitem__(i, seed=seed + 7919 * epoch + i) for i in idx])
        epoch += 1


def model_inputs(batch):
    """The reference's loop over one collated batch (train_ours.py:226-251, L = NumP = 1): yields
    (Frame [B,3,H,W], Event [B,TB,2,H,W], T [B,1], GTEx [B,1], LatentF [B,3,H,W]) per latent frame."""
    latent = batch["SeqLatentF"][:, 0, 0]                 # [B, NumF, 3, H, W]
    frame = batch["SeqBlurryF"][:, 0, 0].contiguous()
    event = batch["SeqHREv"][:, 0].contiguous()
    ts = batch["RelativeLatentTs"][:, 0, 0]               # [B, NumF]
    duty = batch["SeqExposureDuty"][:, 0, 0].contiguous()  # [B, 1]
    for i in range(ts.shape[-1]):
        yield frame, event, ts[:, [i]].contiguous(), duty, latent[:, i].contiguous()


def write_synthetic_clip(path, num_imgs=33, H=64, W=64, events_per_frame=400, seed=0):
    """A small random clip in the .npz layout (tests, smoke runs of `train_ours.py --data`)."""
    g = np.random.RandomState(seed)
    images = g.randint(0, 256, size=(num_imgs, H, W, 3)).astype(np.uint8)
    counts = g.poisson(events_per_frame, size=num_imgs - 1)
    event_idx = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    E = int(event_idx[-1])
    ts = np.sort(g.uniform(0.0, (num_imgs - 1) / 240.0, size=E))
    np.savez(path, images=images, event_idx=event_idx, xs=g.randint(0, W, size=E).astype(np.int16),
             ys=g.randint(0, H, size=E).astype(np.int16), ts=ts, ps=(g.randint(0, 2, size=E) * 2 - 1).astype(np.int8))
    return path
