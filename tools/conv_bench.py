#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM convolution entry point (udet_conv2d) on representative layer shapes of
the hot path, optionally with a forced tile / split-K configuration (tuning aid; needs an MI355X).

  python tools/conv_bench.py [--cfg bm,bn,ks ...] [--reps 20] [--only name]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsupervised_detection_amd import ops  # noqa: E402
from unsupervised_detection_amd._devel import dbg as lib  # noqa: E402  (libudet_debug.so: the test-only hooks)

# name, n, h, w, cin, cout, k, stride, dil, up
SHAPES = [
    ("pwc.dc_conv21", 4, 96, 160, 568, 128, 3, 1, 1, False),
    ("pwc.conv2_2", 4, 96, 160, 376, 96, 3, 1, 1, False),
    ("pwc.conv2_3", 4, 96, 160, 472, 64, 3, 1, 1, False),
    ("pwc.conv2_4", 4, 96, 160, 536, 32, 3, 1, 1, False),
    ("pwc.dc_conv22", 4, 96, 160, 128, 128, 3, 1, 2, False),
    ("pwc.dc_conv31", 4, 48, 80, 600, 128, 3, 1, 1, False),
    ("pwc.dc_conv41", 4, 24, 40, 632, 128, 3, 1, 1, False),
    ("pwc.conv1a", 8, 384, 640, 3, 16, 3, 2, 1, False),
    ("pwc.conv1aa", 8, 192, 320, 16, 16, 3, 1, 1, False),
    ("pwc.conv2a", 8, 192, 320, 16, 32, 3, 2, 1, False),
    ("pwc.conv2aa", 8, 96, 160, 32, 32, 3, 1, 1, False),
    ("gen.conv1", 4, 192, 384, 5, 32, 5, 1, 1, False),
    ("gen.conv17", 4, 192, 384, 16, 2, 3, 1, 1, False),
    ("rec.flow1", 12, 96, 192, 56, 2, 5, 1, 1, False),
    ("rec.flow2", 12, 48, 96, 104, 2, 3, 1, 1, False),
    ("rec.flow1_dgrad", 12, 96, 192, 2, 50, 5, 1, 1, False),     # two input channels, wide output: conv_thin_k
    ("rec.flow2_dgrad", 12, 48, 96, 2, 98, 3, 1, 1, False),
    ("pwc.flow2", 4, 96, 160, 568, 2, 3, 1, 1, False),
    ("rec.aconv1", 12, 192, 384, 3, 16, 7, 2, 1, False),
    ("rec.aconv2", 12, 96, 192, 16, 32, 5, 2, 1, False),
    ("gen.conv5", 4, 48, 96, 128, 128, 3, 1, 1, False),
    ("gen.conv3", 4, 96, 192, 64, 64, 3, 1, 1, False),
    ("gen.conv15_up", 4, 96, 192, 64, 32, 3, 1, 1, True),
    ("gen.conv16", 4, 192, 384, 32, 16, 3, 1, 1, False),
    ("rec.deconv1", 12, 96, 192, 104, 16, 4, 1, 1, False),
    ("rec.deconv3", 12, 24, 48, 392, 64, 4, 1, 1, False),
    ("rec.deconv1_dgrad", 12, 96, 192, 16, 98, 4, 1, 1, False),   # short K (16 channels x 16 taps), wide N
    ("rec.deconv2_dgrad", 12, 48, 96, 32, 194, 4, 1, 1, False),
    ("gen.conv16_dgrad", 4, 192, 384, 16, 32, 3, 1, 1, False),
    ("rec.aconv41", 12, 12, 24, 128, 128, 3, 1, 1, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", action="append", default=[], help="bm,bn,ks forced configuration (repeatable; bm+65536 selects the non-specialised 256-thread kernel, bm+131072 the LDS-DMA kernel, 262144+{8,4} the tile-resident kernel with that tile height, bm+524288 the self-staging LDS-DMA kernel, bm+4194304 / bm+8388608 the LDS-DMA kernel with a 3 / 4 stage ring); "
                    "ks + 256 r: tail split for r workgroup slots per CU; the default heuristics always run")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    cfgs = [None] + [tuple(int(v) for v in c.split(",")) for c in args.cfg]
    g = torch.Generator().manual_seed(0)
    for name, n, h, w, cin, cout, k, s, d, up in SHAPES:
        if args.only and args.only not in name:
            continue
        x = (torch.rand(n, h, w, cin, generator=g) - 0.5).cuda()
        wt = ((torch.rand(k, k, cin, cout, generator=g) - 0.5) * (2.0 / (k * k * cin)) ** 0.5).cuda()
        b = torch.zeros(cout).cuda()
        us = 2 if up else 1
        gflop = 2.0 * n * (h * us // s) * (w * us // s) * cout * cin * k * k * 1e-9
        line = f"{name:16s} {gflop:7.2f} GF |"
        y_ref = None
        for cfg in cfgs:
            if cfg is None:
                lib.udet_debug_force_conv(0, 0, -1)
            else:
                lib.udet_debug_force_conv(*cfg)
            try:
                for _ in range(args.reps):  # warm-up: clocks ramp over the first launches
                    ops.conv2d(x, wt, b, s, d, "leaky", 0.1, up)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    ops.conv2d(x, wt, b, s, d, "leaky", 0.1, up)
                e1.record()
                torch.cuda.synchronize()
                us_ = e0.elapsed_time(e1) * 1e3 / args.reps
                y = ops.conv2d(x, wt, b, s, d, "leaky", 0.1, up)
                if y_ref is None:
                    y_ref = y
                err = float((y - y_ref).abs().max())
                tag = "auto" if cfg is None else "x".join(map(str, (cfg[0] & 0xffff, cfg[1], cfg[2] & 255))) + ("T%d" % (cfg[2] >> 8) if cfg[2] > 255 else "") + ("n" if (cfg[0] >> 16) & 1 else "") + ("d" if (cfg[0] >> 17) & 1 else "") + ("t" if (cfg[0] >> 18) & 1 else "") + ("s" if (cfg[0] >> 19) & 1 else "") + ("d3" if (cfg[0] >> 22) & 1 else "") + ("d4" if (cfg[0] >> 23) & 1 else "")
                line += f" {tag:>11s}: {us_:7.1f}us {gflop / us_ * 1e3:6.1f}TF" + (f" ERR={err:.1e}" if err > 1e-4 else "") + " |"
            except Exception as ex:  # unsupported forced tile
                line += f" {'x'.join(map(str, cfg))}: n/a |"
        print(line, flush=True)
    lib.udet_debug_force_conv(0, 0, -1)


if __name__ == "__main__":
    main()
