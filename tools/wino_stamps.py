#!/usr/bin/env python3
"""Where a fused-Winograd launch's fixed cost goes (libudet_exp.so only: make -C unsupervised_detection_amd/csrc exp):
per-workgroup cycle stamps of conv_wino8_kernel on the generator's 48 x 96 x 128 layer with 1 and 16 K stages.
    python tools/wino_stamps.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import knob_bench  # noqa: E402

lib = knob_bench.load_experiment_build()
import torch  # noqa: E402
from unsupervised_detection_amd import ops  # noqa: E402
dbg = lib  # (the experiment build carries the udet_debug_* hooks itself; libudet_debug.so would bind to libudet.so's state)
dbg.udet_debug_force_conv.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
dbg.udet_debug_force_conv.restype = None

NAMES = ["zero-fill + tables", "first stage lands", "K loop", "exchange + output transform + store loop (both roles)", "stores acknowledged"]
IDX = [0, 1, 2, 3, 5, 6]  # stamps of csrc/conv_wino.hip (4 is unused since both roles store)


def main():
    g = torch.Generator().manual_seed(0)
    for act in ("leaky", "elu"):
        for cin in (8, 128):
            n, h, w, cout = 4, 48, 96, 128
            x = (torch.rand(n, h, w, cin, generator=g) - 0.5).cuda()
            wt = ((torch.rand(3, 3, cin, cout, generator=g) - 0.5) * (2.0 / (9 * cin)) ** 0.5).cuda()
            b = torch.zeros(cout).cuda()
            dbg.udet_debug_force_conv((1 << 25) + 2, 0, 1)
            for _ in range(5):
                ops.conv2d(x, wt, b, 1, 1, act, 0.1, False)
            torch.cuda.synchronize()
            buf = (ctypes.c_longlong * (144 * 8))()
            assert lib.udet_exp_wino_stamps(buf, 144 * 8) == 0
            seg = [0.0] * 5
            for blk in range(144):
                for k in range(5):
                    seg[k] += (buf[blk * 8 + IDX[k + 1]] - buf[blk * 8 + IDX[k]]) / 144.0
            print("%s, %d K stage(s): " % (act, cin // 8) + "; ".join("%s %.0f" % (nm, v) for nm, v in zip(NAMES, seg)) + "  (cycles, mean of 144 workgroups)")
    dbg.udet_debug_force_conv(0, 0, -1)


if __name__ == "__main__":
    main()
