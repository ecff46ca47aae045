#!/usr/bin/env python3
"""Per-launch report of one profiled step.

  UDET_PROF_DUMP=layers.csv python bench.py --steps 20 --warmup 5 --no-cpu-baseline --cycles 0
  python tools/layer_report.py layers.csv [--rate 100] [--top 40]

The CSV (one line per launch group of the serial profiling pass: category, layer, ms, algorithmic GFLOP, MB) is written by
libudet.so while bench.py's profiling pass runs.  The report ranks launches by the time they lose against a flat --rate
TFLOP/s (the practical ceiling of the fp32 MFMA kernels here is about 100-105), which is where to look first."""
import argparse
import collections
import csv

CAT = {0: "fwd", 1: "dgrad", 2: "wgrad", 3: "warp", 4: "costvol"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--rate", type=float, default=100.0, help="reference TFLOP/s for the 'lost time' column")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    rows = [(int(r[0]), r[1], float(r[2]), float(r[3]), float(r[4])) for r in csv.reader(open(a.csv))]
    tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for c, _, ms, gf, _ in rows:
        t = tot[c]
        t[0] += ms
        t[1] += gf
        t[2] += 1
    print("category   launches        ms     GFLOP   TFLOP/s")
    for c in sorted(tot):
        ms, gf, n = tot[c]
        print(f"{CAT.get(c, c):9s} {n:9d} {ms:9.3f} {gf:9.1f} {gf / ms if ms else 0:9.1f}")
    conv = [r for r in rows if r[0] < 3]
    lost = sorted(((ms - gf / a.rate, c, name, ms, gf) for c, name, ms, gf, _ in conv), reverse=True)
    print(f"\nconvolution launches by time lost against {a.rate:.0f} TFLOP/s (total {sum(l[0] for l in lost):.2f} ms "
          f"of {sum(r[2] for r in conv):.2f} ms):")
    for w, c, name, ms, gf in lost[:a.top]:
        print(f"  {CAT[c]:5s} {name:36s} {ms * 1e3:8.1f} us {gf:8.2f} GF {gf / ms if ms else 0:6.1f} TF   lost {w * 1e3:7.1f} us")


if __name__ == "__main__":
    main()
