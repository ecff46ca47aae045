#!/usr/bin/env python3
"""Fixed cost vs per-stage cost of the fused Winograd kernels on the generator's 48 x 96 x 128 grid (round 6):
the same launch with 8 ... 512 input channels (1 ... 64 K stages); a line fit gives the per-stage time and the intercept = prologue +
epilogue + launch.  python tools/wino_fixed_cost.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsupervised_detection_amd import ops  # noqa: E402
from unsupervised_detection_amd._devel import dbg  # noqa: E402


def time_it(fn, reps=50):
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
    for (n, h, w, cout) in ((4, 48, 96, 128), (4, 96, 192, 64), (4, 96, 160, 128)):
        for act in ("leaky", "elu"):
            for v in (0, 2, 4):
                row = []
                for cin in (8, 32, 64, 128, 256, 512):
                    x = (torch.rand(n, h, w, cin, generator=g) - 0.5).cuda()
                    wt = ((torch.rand(3, 3, cin, cout, generator=g) - 0.5) * (2.0 / (9 * cin)) ** 0.5).cuda()
                    b = torch.zeros(cout).cuda()
                    dbg.udet_debug_force_conv((1 << 25) + v, 0, 1)
                    us = time_it(lambda: ops.conv2d(x, wt, b, 1, 1, act, 0.1, False))
                    assert (dbg.udet_debug_last_conv() & 0xff) == 9
                    row.append((cin // 8, us))
                (s0, t0), (s1, t1) = row[2], row[-1]
                per = (t1 - t0) / (s1 - s0)
                print("N=%d %dx%d Cout=%d %-5s variant %d: %s | per stage %.2f us, intercept %.1f us" %
                      (n, h, w, cout, act, v, "  ".join("%d:%.1f" % r for r in row), per, t0 - per * s0), flush=True)
    dbg.udet_debug_force_conv(0, 0, -1)


if __name__ == "__main__":
    main()
