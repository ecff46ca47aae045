"""Deterministic inputs / weights shared by oracle/make_golden.py (which produced tests/golden/*.npz from the reference's
own Python) and the tests that replay them (TEST INFRASTRUCTURE ONLY).  Everything is drawn from numpy's frozen legacy
generator (np.random.RandomState: stream guaranteed stable across numpy versions), so the fixtures only have to hold outputs."""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

STEP_CFG = dict(batch_size=2, in_height=128, in_width=192, img_height=64, img_width=128, flow_normalizer=80.0, cbn=0.5,
                epsilon=75.0, beta1=0.9)


# BASELINE.json configs[1] -- the shape bench.py measures (tests/golden/step_cfg2.npz)
STEP_CFG2 = dict(batch_size=4, in_height=384, in_width=640, img_height=192, img_width=384, flow_normalizer=80.0, cbn=0.5,
                 epsilon=75.0, beta1=0.9)
CFG2_STRIDE = 4  # mask / predictions are kept at every 4th row and column (the flow, an INPUT of the replay, is kept whole)


def tensor(name: str, shape, scale=1.0, offset=0.0):
    """float32 array ~ offset + scale * N(0,1), seeded by the name."""
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    return (offset + scale * rs.standard_normal(size=tuple(shape))).astype(np.float32)


def params(specs):
    """Weights for a (name, shape, init) spec list: fan-in scaled normal kernels, small random biases / beta, gamma ~ 1."""
    out = OrderedDict()
    for name, shape, _ in specs:
        leaf = name.rsplit("/", 1)[1]
        if leaf in ("kernel", "weights"):
            fan_in = int(np.prod(shape[:-1]))
            out[name] = tensor(name, shape, scale=(2.0 / fan_in) ** 0.5)
        elif leaf == "gamma":
            out[name] = tensor(name, shape, scale=0.1, offset=1.0)
        else:  # bias / biases / beta
            out[name] = tensor(name, shape, scale=0.05)
    return out


def smooth(name: str, shape, cell=16, passes=8):
    """Low-frequency field in [0,1): random coarse grid, nearest-upsampled, box-filtered `passes` times (pure numpy)."""
    n, h, w, c = shape
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    hh, ww = h + 2 * passes, w + 2 * passes
    g = rs.rand(n, -(-hh // cell), -(-ww // cell), c)
    x = np.kron(g, np.ones((1, cell, cell, 1)))[:, :hh, :ww]
    for _ in range(passes):
        x = (x[:, :-2] + x[:, 1:-1] + x[:, 2:]) / 3.0
        x = (x[:, :, :-2] + x[:, :, 1:-1] + x[:, :, 2:]) / 3.0
    x = (x - x.min()) / (x.max() - x.min())
    return x.astype(np.float32)


def image_pair(batch=2, h=128, w=192):
    """Two frames in [-0.5, 0.5]: the second is the first shifted by (2, 3) pixels plus 1 % noise."""
    base = smooth("golden/frame", (batch, h + 8, w + 8, 3))
    img1 = base[:, 4:4 + h, 4:4 + w] - 0.5
    img2 = base[:, 2:2 + h, 1:1 + w] - 0.5 + 0.01 * tensor("golden/frame_noise", (batch, h, w, 3))
    return np.ascontiguousarray(img1, np.float32), np.ascontiguousarray(img2, np.float32)


def gt_mask(batch=2, h=128, w=192):
    m = smooth("golden/gt", (batch, h, w, 1), cell=32, passes=4)
    return (m > 0.55).astype(np.float32)


def grad_summary(g: np.ndarray):
    """What the fixtures keep of a gradient tensor: L2 norm, signed sum, the first 16 entries."""
    f = np.asarray(g, np.float64).ravel()
    return np.concatenate([[np.sqrt((f * f).sum()), f.sum()], np.pad(f[:16], (0, max(0, 16 - f.size)))]).astype(np.float64)


GRAD_SUBSET_STRIDE = 64


def grad_subset(g: np.ndarray, name: str):
    """Element-level part of the gradient fixtures (round 6): every 64th element of the flattened tensor, starting at a
    per-variable offset derived from its name (so that the picks do not line up with the channel strides), or the whole tensor when it
    has at most 4096 elements.  float32, as the reference computes them.  Tests take the same picks of their own gradient and compare
    element by element at 1e-3 of the tensor's largest element."""
    f = np.asarray(g, np.float32).ravel()
    if f.size <= 4096:
        return f.copy()
    off = sum(name.encode()) % GRAD_SUBSET_STRIDE
    return np.ascontiguousarray(f[off::GRAD_SUBSET_STRIDE])


# TensorFlow's own unit-test vectors for the resize kernels (tensorflow/python/ops/image_ops_test.py, r1.13,
# ResizeImagesTest.testResizeUpAlignCornersFalse / testResizeUpAlignCornersTrue): input [1,3,2,1], expected kernel outputs
TF_RESIZE_FALSE = dict(
    data=[64, 32, 32, 64, 50, 100], in_hw=(3, 2), out_hw=(6, 4),
    bilinear=[64.0, 48.0, 32.0, 32.0, 48.0, 48.0, 48.0, 48.0, 32.0, 48.0, 64.0, 64.0, 41.0, 61.5, 82.0, 82.0, 50.0, 75.0, 100.0, 100.0,
              50.0, 75.0, 100.0, 100.0],
    nearest=[64.0, 64.0, 32.0, 32.0, 64.0, 64.0, 32.0, 32.0, 32.0, 32.0, 64.0, 64.0, 32.0, 32.0, 64.0, 64.0, 50.0, 50.0, 100.0, 100.0,
             50.0, 50.0, 100.0, 100.0])
TF_RESIZE_TRUE = dict(
    data=[6, 3, 3, 6, 6, 9], in_hw=(3, 2), out_hw=(5, 4),
    bilinear=[6.0, 5.0, 4.0, 3.0, 4.5, 4.5, 4.5, 4.5, 3.0, 4.0, 5.0, 6.0, 4.5, 5.5, 6.5, 7.5, 6.0, 7.0, 8.0, 9.0],
    nearest=[6.0, 6.0, 3.0, 3.0, 3.0, 3.0, 6.0, 6.0, 3.0, 3.0, 6.0, 6.0, 6.0, 6.0, 9.0, 9.0, 6.0, 6.0, 9.0, 9.0])


# tensorflow/python/kernel_tests/conv_ops_test.py (r1.13), Conv2DTest: input and filter hold 1, 2, 3, ... in row-major
# order (NHWC / HWIO); (input shape, filter shape, stride, expected flattened output) of the 'SAME'-padded cases
TF_CONV_SAME = [
    ([1, 2, 3, 3], [2, 2, 3, 3], 2, [2271.0, 2367.0, 2463.0, 1230.0, 1305.0, 1380.0]),   # testConv2D2x2FilterStride2Same
    ([1, 3, 3, 1], [1, 1, 1, 1], 2, [1.0, 3.0, 7.0, 9.0]),                                # testConv2DKernelSmallerThanStrideSame
    ([1, 4, 4, 1], [1, 1, 1, 1], 2, [1.0, 3.0, 9.0, 11.0]),
    ([1, 4, 4, 1], [2, 2, 1, 1], 3, [44.0, 28.0, 41.0, 16.0]),
]
