#!/usr/bin/env python3
"""Micro-benchmark of the filter-gradient entry point (udet_conv2d_backward_filter) on representative layer shapes of the hot
path (tuning aid; needs an MI355X).  The split count / staging variant are the autotuned ones (UDET_WGRAD_TUNE=0: the built-in
heuristic).

  python tools/wgrad_bench.py [--reps 20] [--only name]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsupervised_detection_amd import ops  # noqa: E402
from unsupervised_detection_amd._ffi import lib  # noqa: E402

# name, n, h, w, cin, cout, k, stride
SHAPES = [
    ("gen.conv5", 4, 48, 96, 128, 128, 3, 1),
    ("gen.conv3", 4, 96, 192, 64, 64, 3, 1),
    ("gen.conv16", 4, 192, 384, 32, 16, 3, 1),
    ("gen.conv1", 4, 192, 384, 8, 32, 5, 1),
    ("rec.deconv1", 12, 96, 192, 104, 16, 4, 1),
    ("rec.deconv2", 12, 48, 96, 200, 32, 4, 1),
    ("rec.deconv3", 12, 24, 48, 392, 64, 4, 1),
    ("rec.bconv2", 12, 96, 192, 16, 32, 5, 2),
    ("rec.bconv41", 12, 12, 24, 128, 128, 3, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--nsplit", default="", help="comma-separated pixel-slice counts to pin in turn (udet_debug_force_wgrad); default: the tuned count")
    ap.add_argument("--dma", type=int, default=-1, help="staging variant pinned with --nsplit: 0 register-staged, 1 / 2 LDS-DMA with a 2- / 3-stage ring")
    a = ap.parse_args()
    from unsupervised_detection_amd._devel import dbg as _dbg
    splits = [int(v) for v in a.nsplit.split(",") if v] or [0]
    g = torch.Generator().manual_seed(0)
    if os.environ.get("UDET_WGRAD_TUNE", "1") == "1":
        from unsupervised_detection_amd._devel import dbg
        dbg.udet_debug_set_tuning(1)
    for name, n, h, w, cin, cout, k, s in SHAPES:
        if a.only and a.only not in name:
            continue
        x = (torch.rand(n, h, w, cin, generator=g) - 0.5).cuda()
        oh, ow = -(-h // s), -(-w // s)
        dy = (torch.rand(n, oh, ow, cout, generator=g) - 0.5).cuda()
        gflop = 2.0 * n * oh * ow * cout * cin * k * k * 1e-9
        for ns in splits:
          _dbg.udet_debug_force_wgrad(ns, a.dma)
          name_ = name if ns == 0 else "%s ns=%d" % (name, ns)
          for _ in range(a.reps):
              ops.conv2d_backward_filter(x, dy, None, (k, k), s, 1, "none", 0.0, False)
          torch.cuda.synchronize()
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          for _ in range(a.reps):
              ops.conv2d_backward_filter(x, dy, None, (k, k), s, 1, "none", 0.0, False)
          e1.record()
          torch.cuda.synchronize()
          us = e0.elapsed_time(e1) * 1e3 / a.reps
          print(f"{name_:20s} {gflop:7.2f} GF  {us:7.1f} us  {gflop / us * 1e3:6.1f} TF (incl. reduction + python launch overhead)", flush=True)
    _dbg.udet_debug_force_wgrad(0, -1)


if __name__ == "__main__":
    main()
