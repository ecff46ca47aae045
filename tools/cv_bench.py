#!/usr/bin/env python3
"""Micro-benchmark of the cost-volume and warp entry points on the five PWC-Net pyramid levels of the benchmark
configuration (B=4, 384x640 input): algorithmic bytes (read c1 + warped features, write 81 channels) / time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsupervised_detection_amd import ops  # noqa: E402

LEVELS = [(6, 6, 10, 196), (5, 12, 20, 128), (4, 24, 40, 96), (3, 48, 80, 64), (2, 96, 160, 32)]


def timed(fn, reps=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    g = torch.Generator().manual_seed(0)
    for lvl, h, w, c in LEVELS:
        c1 = torch.randn(4, h, w, c, generator=g).cuda()
        c2 = torch.randn(4, h, w, c, generator=g).cuda()
        flow = (torch.randn(4, h, w, 2, generator=g) * 2).cuda()
        us = timed(lambda: ops.cost_volume(c1, c2))
        mb = 4 * h * w * (2 * c + 81) * 4 / 1e6
        uw = timed(lambda: ops.dense_image_warp(c2, flow))
        mbw = 4 * h * w * (2 * c + 2) * 4 / 1e6
        uf = timed(lambda: ops.warp_cost_volume(c1, c2, flow if lvl != 6 else None, 20.0 / 2 ** lvl))
        mbf = 4 * h * w * (2 * c + 2 + 81) * 4 / 1e6  # read c1 + c2 + flow, write 81 channels (the plan's launch also writes c1 into the slab)
        print(f"level {lvl} {h}x{w}x{c}: cost_volume {us:6.1f} us {mb / us * 1e3:7.1f} GB/s | warp {uw:6.1f} us {mbw / uw * 1e3:7.1f} GB/s | "
              f"fused warp+cost_volume {uf:6.1f} us {mbf / uf * 1e3:7.1f} GB/s (two kernels: {us + uw:6.1f} us)", flush=True)


if __name__ == "__main__":
    main()
