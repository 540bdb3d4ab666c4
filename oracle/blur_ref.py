"""ORACLE (test infrastructure only): numpy restatement of the blur-level maps.

Reference call sites: myutils/utils.py:34-49 (Frame2Lap) and :15-31 (Frame2DCP), used inside
EVFIAutoEx.forward (models/Ours/model_singleframe.py:311-323).  The arithmetic itself lives in
OpenCV (``cv2``), a third-party dependency the reference does not pin (no requirements file)
and that is absent from this image, so no golden vector can be generated:

    PARITY UNPINNED -- restated from OpenCV's published 8-bit algorithms (4.x constants).

Frame2Lap, per sample:
    u8   = (frame * 255) truncated to uint8                       (numpy astype)
    gray = cv2.cvtColor(u8, COLOR_BGR2GRAY) on H x W x 3 data whose channel 0 is really R:
           OpenCV 4.x 8-bit path: (c0*3735 + c1*19235 + c2*9798 + (1<<14)) >> 15
           (c0 is treated as "B"; so the weights land 0.114*R + 0.587*G + 0.299*B)
    lap  = cv2.Laplacian(gray, CV_64F)  -> ksize=1 aperture [[0,1,0],[1,-4,1],[0,1,0]],
           BORDER_REFLECT_101, no scaling; cast to float32 (integer valued, range +-1020)
Frame2DCP, per sample: min over the 3 channels, then a 35x35 erosion (minimum filter); OpenCV's
    default erosion border value is +inf, i.e. the window is clipped to the image.
"""
import numpy as np

BY15, GY15, RY15, GRAY_SHIFT = 3735, 19235, 9798, 15


def frame_to_gray_u8(frame):
    """frame: float [3,H,W] in [0,1] -> uint8 [H,W] (OpenCV BGR2GRAY applied to RGB-ordered data)."""
    u8 = (np.asarray(frame, dtype=np.float32) * np.float32(255)).astype(np.uint8).astype(np.int64)
    c0, c1, c2 = u8[0], u8[1], u8[2]
    return ((c0 * BY15 + c1 * GY15 + c2 * RY15 + (1 << (GRAY_SHIFT - 1))) >> GRAY_SHIFT).astype(np.uint8)


def laplacian_reflect101(gray):
    g = np.pad(gray.astype(np.int64), 1, mode="reflect")     # numpy 'reflect' == REFLECT_101
    return (g[:-2, 1:-1] + g[2:, 1:-1] + g[1:-1, :-2] + g[1:-1, 2:] - 4 * g[1:-1, 1:-1])


def frame2lap(frames):
    """frames: float [B,3,H,W] -> float32 [B,1,H,W]."""
    frames = np.asarray(frames)
    out = np.stack([laplacian_reflect101(frame_to_gray_u8(f)).astype(np.float32) for f in frames])
    return out[:, None]


def frame2dcp(frames, sz=35):
    """frames: float [B,3,H,W] -> float32 [B,1,H,W]; clipped-window minimum filter."""
    frames = np.asarray(frames, dtype=np.float32)
    B, _, H, W = frames.shape
    r = sz // 2
    dc = frames.min(axis=1)
    pad = np.pad(dc, ((0, 0), (r, r), (r, r)), mode="constant", constant_values=np.inf)
    # separable minimum
    tmp = np.full_like(dc, np.inf)
    for d in range(sz):
        tmp = np.minimum(tmp, pad[:, r:r + H, d:d + W])
    padv = np.pad(tmp, ((0, 0), (r, r), (0, 0)), mode="constant", constant_values=np.inf)
    out = np.full_like(dc, np.inf)
    for d in range(sz):
        out = np.minimum(out, padv[:, d:d + H, :])
    return out[:, None].astype(np.float32)
