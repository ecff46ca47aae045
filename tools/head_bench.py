#!/usr/bin/env python3
"""Micro-benchmark of the 2-channel heads through udet_conv2d / udet_conv2d_backward_data: the direct kernels (conv_thin.hip, the
untuned default) against the implicit-GEMM kernel (forced) on the head shapes of the path.  Needs an MI355X."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsupervised_detection_amd import ops  # noqa: E402
from unsupervised_detection_amd._devel import dbg  # noqa: E402

# name, n, h, w, cin, k
SHAPES = [("rec.flow1", 12, 96, 192, 50, 5), ("rec.flow2", 12, 48, 96, 98, 3), ("rec.flow3", 12, 24, 48, 194, 3),
          ("rec.flow4", 12, 12, 24, 386, 3), ("gen.conv17", 4, 192, 384, 16, 3), ("pwc.flow2", 4, 96, 160, 565, 3),
          ("pwc.flow3", 4, 48, 80, 597, 3), ("pwc.flow4", 4, 24, 40, 629, 3), ("pwc.dc_conv27", 4, 96, 160, 32, 3)]


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    for name, n, h, w, cin, k in SHAPES:
        x = (torch.rand(n, h, w, cin, generator=g) - 0.5).cuda()
        wt = (torch.rand(k, k, cin, 2, generator=g) - 0.5).cuda()
        b = torch.zeros(2).cuda()
        dy = (torch.rand(n, h, w, 2, generator=g) - 0.5).cuda()
        row = [name]
        for label, cfg in (("direct", (0, 0, -1)), ("igemm", (128, 32, 1))):
            dbg.udet_debug_force_conv(*cfg)
            tf = timed(lambda: ops.conv2d(x, wt, b, 1, 1, "none", 0.0, False))
            fam_f = dbg.udet_debug_last_conv() & 0xff
            td = timed(lambda: ops.conv2d_backward_data(dy, None, wt, (h, w), 1, 1, "none", 0.0))
            fam_d = dbg.udet_debug_last_conv() & 0xff
            row.append("%s: fwd %6.1f us (family %d)  bwd-data %6.1f us (family %d)" % (label, tf, fam_f, td, fam_d))
        dbg.udet_debug_force_conv(0, 0, -1)
        print(" | ".join(row), flush=True)


if __name__ == "__main__":
    main()
