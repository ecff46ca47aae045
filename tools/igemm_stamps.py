#!/usr/bin/env python3
"""Where an implicit-GEMM (LDS-DMA) launch's time goes: per-workgroup cycle stamps (libudet_exp.so only: make -C .../csrc exp).
    python tools/igemm_stamps.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import knob_bench  # noqa: E402

lib = knob_bench.load_experiment_build()
import torch  # noqa: E402
from unsupervised_detection_amd import ops  # noqa: E402

lib.udet_debug_force_conv.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
lib.udet_debug_force_conv.restype = None
lib.udet_debug_last_conv.restype = ctypes.c_int
TS = 12  # stamps per workgroup (conv_igemm.hip: IGEMM_TS)
NAMES = ["tables", "first stage lands", "K loop", "tile store (issue)", "stores acknowledged"]
# name, n, h, w, cin, cout, k, stride, dil, (bm, bn, ks)
SHAPES = [
    ("rec.bconv41", 12, 12, 24, 128, 128, 3, 1, 1, (64, 64, 3)),
    ("rec.bconv31", 12, 24, 48, 64, 64, 3, 1, 1, (64, 64, 1)),
    ("rec.bconv2", 12, 96, 192, 16, 32, 5, 2, 1, (128, 32, 1)),
    ("pwc.conv4_2", 4, 24, 40, 440, 96, 3, 1, 1, (128, 32, 4)),
    ("pwc.conv5_1", 4, 12, 20, 344, 128, 3, 1, 1, (64, 64, 8)),
    ("gen.conv9 (d=8)", 4, 48, 96, 128, 128, 3, 1, 8, (64, 64, 1)),
    ("pwc.dc_conv24 (d=8)", 4, 96, 160, 128, 96, 3, 1, 8, (128, 96, 1)),
]


def main():
    g = torch.Generator().manual_seed(0)
    for name, n, h, w, cin, cout, k, s, d, (bm, bn, ks) in SHAPES:
        for act in ("leaky", "elu"):
            x = (torch.rand(n, h, w, cin, generator=g) - 0.5).cuda()
            wt = ((torch.rand(k, k, cin, cout, generator=g) - 0.5) * (2.0 / (k * k * cin)) ** 0.5).cuda()
            b = torch.zeros(cout).cuda()
            lib.udet_debug_force_conv(bm + (1 << 17) + (1 << 20), bn, ks)  # LDS-DMA kernel, split-K through the second launch
            for _ in range(5):
                ops.conv2d(x, wt, b, s, d, act, 0.1, False)
            torch.cuda.synchronize()
            fam = lib.udet_debug_last_conv()
            oh, ow = (h + s - 1) // s, (w + s - 1) // s
            nblk = min(1024, (n * oh * ow + bm - 1) // bm)
            buf = (ctypes.c_longlong * (nblk * TS))()
            assert lib.udet_exp_igemm_stamps(buf, nblk * TS) == 0
            seg = [0.0] * 5
            for blk in range(nblk):
                for q in range(5):
                    seg[q] += (buf[blk * TS + q + 1] - buf[blk * TS + q]) / nblk
            st = [0.0, 0.0, 0.0]  # last MFMA (3) -> store set-up done (8) -> first half block issued (9) -> tile stored (4)
            for blk in range(nblk):
                st[0] += (buf[blk * TS + 8] - buf[blk * TS + 3]) / nblk
                st[1] += (buf[blk * TS + 9] - buf[blk * TS + 8]) / nblk
                st[2] += (buf[blk * TS + 4] - buf[blk * TS + 9]) / nblk
            pre = [0.0, 0.0, 0.0]  # entry -> block decoded (6) -> tables written (7) -> barrier passed (1)
            for blk in range(nblk):
                pre[0] += (buf[blk * TS + 6] - buf[blk * TS]) / nblk
                pre[1] += (buf[blk * TS + 7] - buf[blk * TS + 6]) / nblk
                pre[2] += (buf[blk * TS + 1] - buf[blk * TS + 7]) / nblk
            print("%-20s %-5s %dx%d ks=%d (family %d, %d x-blocks): [tables = decode %.0f + fill %.0f + barrier %.0f] " % (
                  name, act, bm, bn, ks, fam & 0xff, nblk, pre[0], pre[1], pre[2]) +
                  "; ".join("%s %.0f" % (nm, v) for nm, v in zip(NAMES, seg)) +
                  " [tile store = set-up %.0f + first half block %.0f + rest %.0f]  (cycles)" % tuple(st), flush=True)
    lib.udet_debug_force_conv(0, 0, -1)


if __name__ == "__main__":
    main()
