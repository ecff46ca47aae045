"""Input-side helpers: the readers' preprocessing restated on device, and the synthetic DAVIS-shaped
frame pairs the benchmark uses (there is no dataset / network here).

preprocess_image follows data/davis2016_data_utils.py:86-91 (identical in fbms_data_utils.py:204-209 and
segtrackv2_data_utils.py:89-94): uint8 -> /255 - 0.5 -> legacy-bilinear resize to 384x640."""
from __future__ import annotations

import numpy as np
import torch

from . import ops

READER_H, READER_W = 384, 640


def preprocess_image(frames_u8: torch.Tensor, out_h: int = READER_H, out_w: int = READER_W) -> torch.Tensor:
    """frames_u8: [N,H,W,3] uint8 on the GPU -> float32 [N,out_h,out_w,3] in [-0.5, 0.5]."""
    x = frames_u8.to(torch.float32) / 255.0 - 0.5
    return ops.resize_bilinear_legacy(x.contiguous(), out_h, out_w)


def _smooth_noise(rng, shape, sigma):
    """low-pass filtered uniform noise in [0,1] (separable box filters ~ gaussian), numpy."""
    a = rng.random(shape, dtype=np.float32)
    k = max(1, int(sigma))
    for _ in range(3):
        for ax in (0, 1):
            c = np.cumsum(np.pad(a, [(k, k) if i == ax else (0, 0) for i in range(a.ndim)], mode="reflect"), axis=ax, dtype=np.float64)
            n = a.shape[ax]
            hi = np.take(c, np.arange(2 * k, 2 * k + n), axis=ax)
            lo = np.take(c, np.arange(0, n), axis=ax)
            a = ((hi - lo) / (2 * k)).astype(np.float32)
    a -= a.min()
    a /= max(a.max(), 1e-6)
    return a


def synthetic_davis_pairs(batch: int, seed: int, h: int = 480, w: int = 854, max_disp: float = 8.0):
    """DAVIS-480p-shaped uint8 frame pairs: frame 2 = frame 1 warped by a smooth random displacement field of
    <= max_disp px + 1 % noise.  Returns two uint8 arrays [B,h,w,3]."""
    rng = np.random.default_rng(seed)
    f1 = np.empty((batch, h, w, 3), np.uint8)
    f2 = np.empty((batch, h, w, 3), np.uint8)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    for b in range(batch):
        img = 0.6 * _smooth_noise(rng, (h, w, 3), 6) + 0.4 * _smooth_noise(rng, (h, w, 3), 2)
        dy = (_smooth_noise(rng, (h, w), 40) * 2 - 1) * max_disp
        dx = (_smooth_noise(rng, (h, w), 40) * 2 - 1) * max_disp
        sy = np.clip(yy + dy, 0, h - 1.001)
        sx = np.clip(xx + dx, 0, w - 1.001)
        y0, x0 = sy.astype(np.int32), sx.astype(np.int32)
        ty, tx = (sy - y0)[..., None], (sx - x0)[..., None]
        warped = ((1 - ty) * ((1 - tx) * img[y0, x0] + tx * img[y0, x0 + 1]) +
                  ty * ((1 - tx) * img[y0 + 1, x0] + tx * img[y0 + 1, x0 + 1]))
        warped = np.clip(warped + 0.01 * rng.standard_normal(warped.shape).astype(np.float32), 0, 1)
        f1[b] = np.round(img * 255).astype(np.uint8)
        f2[b] = np.round(warped * 255).astype(np.uint8)
    return f1, f2
