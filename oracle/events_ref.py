"""ORACLE (test infrastructure only): numpy restatement of the event-voxel binning.

Follows dataloader/encodings.py:307-350 (events_to_stack), :243-268 (events_to_image) and
:77-99 (binary_search_torch_tensor) of the reference, including three behaviours that a "clean"
histogram would not have:

  * the custom binary search returns the index of an exact hit at the probed position
    (left end, right end or midpoint, in that order), otherwise ``l`` ('left') or ``r`` ('right');
    the bin is the half-open slice [beg, end) with end = search(tend,'right') + 1, so two
    neighbouring bins can share events;
  * a count is ``p * p`` added in float32 where p > 0 (positive image) or p < 0 (negative image);
  * events whose x/y fall outside the sensor are zeroed IN PLACE in the caller's xs/ys
    (events_to_image writes through the slice view, encodings.py:253-256) with weight 0 -- so if
    such an event is shared with a LATER bin it re-appears there as a hit on pixel (0, 0).

Pinned bit-exactly against the reference function itself (tests/golden/make_golden.py imports
dataloader/encodings.py, which needs only numpy/torch) on the cases in tests/golden/events_*.npz.
"""
import numpy as np


def bsearch(t, l, r, x, side="left"):
    """encodings.py:77-99."""
    if r is None:
        r = len(t) - 1
    while l <= r:
        if t[l] == x:
            return l
        if t[r] == x:
            return r
        mid = l + (r - l) // 2
        if t[mid] == x:
            return mid
        elif t[mid] < x:
            l = mid + 1
        else:
            r = mid - 1
    return l if side == "left" else r


def bin_bounds(ts, B):
    """[(beg, end)] per bin, exactly the float64 arithmetic of encodings.py:326-332."""
    ts = np.asarray(ts, dtype=np.float64)
    n = len(ts)
    dt = ts[-1] - ts[0] + 1e-6
    delta_t = dt / B
    out = []
    for bi in range(B):
        tstart = ts[0] + delta_t * bi
        tend = tstart + delta_t
        beg = bsearch(ts, 0, n - 1, tstart)
        end = bsearch(ts, 0, n - 1, tend, side="right") + 1
        out.append((beg, end))
    return out


def events_to_stack(xs, ys, ts, ps, B, sensor_size):
    """-> float32 [2, B, H, W] (index 0 positive, 1 negative); inputs are not modified."""
    H, W = sensor_size
    xs = np.array(xs, dtype=np.float64, copy=True)
    ys = np.array(ys, dtype=np.float64, copy=True)
    ts = np.asarray(ts, dtype=np.float64)
    ps = np.asarray(ps, dtype=np.float32)
    stack = np.zeros((2, B, H, W), dtype=np.float32)
    if ts.sum() == 0 or len(ts) <= 3:
        return stack
    assert len(xs) == len(ys) == len(ts) == len(ps)
    for bi, (beg, end) in enumerate(bin_bounds(ts, B)):
        sl = slice(beg, end)          # python slice semantics: empty when beg >= end
        p = ps[sl]
        for pol, sel in ((0, p > 0), (1, p < 0)):
            x, y = xs[sl], ys[sl]     # views: the zeroing below persists, as in the reference
            bad = (x >= W) | (x < 0) | (y >= H) | (y < 0)
            val = np.where(sel, p * p, np.float32(0)).astype(np.float32)
            x[bad] = 0
            y[bad] = 0
            val[bad] = 0
            np.add.at(stack[pol, bi], (y.astype(np.int64), x.astype(np.int64)), val)
    return stack
