#!/usr/bin/env python3
"""How full is the chip during a pipelined step?  Works on the raw rocprofv3 kernel trace (one row per dispatch):

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o steps -- \
      python $REPO/bench.py --tune-cache $REPO/profiles/rNN_tune.txt --trace-only --steps 10 --warmup 2
  python tools/timeline_report.py /tmp/tr --steps 12 --out gpurun_out/timeline.json

For every dispatch: start, end, workgroups (grid / workgroup size), threads per workgroup.  The last `--keep` steps' worth of
dispatches are cut into `--bin` microsecond bins; per bin the script sums, over the kernels running in it, the share of the chip
the launch can occupy at all: min(1, workgroups / (256 CUs x workgroups-per-CU)), where workgroups-per-CU = 2048 threads / threads
per workgroup capped at 8 (an upper bound of the launch's own parallelism: registers / LDS may admit fewer).  Reported:
  * busy: fraction of the window in which at least one kernel runs; mean number of concurrent kernels;
  * fill: mean over time of min(1, sum of the running launches' shares) -- the time-weighted share of the chip that has ANY work;
  * the kernels that run while the chip is less than half full, by the time they spend there (where a better packing would pay).
"""
import argparse
import collections
import csv
import glob
import json
import os


def short(name):
    return name.replace("void udet::", "").replace("udet::", "").split("(")[0][:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace_dir")
    ap.add_argument("--steps", type=int, required=True, help="warm-up + timed steps in the trace")
    ap.add_argument("--keep", type=int, default=6, help="steps (from the end of the trace) to analyse")
    ap.add_argument("--bin", type=float, default=5.0, help="bin width, microseconds")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    files = sorted(glob.glob(os.path.join(a.trace_dir, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        raise SystemExit("no *kernel_trace.csv under %s" % a.trace_dir)
    rows = list(csv.DictReader(open(files[-1])))
    ev = []
    for r in rows:
        wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
        grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        nwg = max(1, grid // max(1, wg))
        per_cu = max(1, min(8, 2048 // max(64, wg)))
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), min(1.0, nwg / (256.0 * per_cu)), nwg))
    ev.sort()
    per_step = len(ev) // a.steps  # (plan construction adds a handful of dispatches at the front: dropped with the warm-up)
    ev = ev[-per_step * a.keep:]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    binw = a.bin * 1e3
    nb = int((t1 - t0) / binw) + 1
    share = [0.0] * nb
    count = [0.0] * nb
    for s, e, _, sh, _ in ev:
        b0, b1 = int((s - t0) / binw), int((e - t0) / binw)
        for b in range(b0, b1 + 1):
            lo, hi = max(s, t0 + b * binw), min(e, t0 + (b + 1) * binw)
            if hi > lo:
                f = (hi - lo) / binw
                share[b] += sh * f
                count[b] += f
    busy = sum(1 for c in count if c > 0.02) / nb
    fill = sum(min(1.0, s) for s in share) / nb
    low = collections.Counter()
    for s, e, name, sh, nwg in ev:
        b0, b1 = int((s - t0) / binw), int((e - t0) / binw)
        for b in range(b0, b1 + 1):
            if share[b] < 0.5:
                lo, hi = max(s, t0 + b * binw), min(e, t0 + (b + 1) * binw)
                if hi > lo:
                    low[name] += (hi - lo) / 1e3
    hist = collections.Counter(min(10, int(min(1.0, s) * 10)) for s in share)
    rep = {"trace": os.path.basename(files[-1]), "steps_analysed": a.keep, "dispatches_per_step": per_step,
           "window_ms_per_step": round((t1 - t0) / 1e6 / a.keep, 3), "busy_fraction": round(busy, 4),
           "mean_concurrent_kernels": round(sum(count) / nb, 3), "fill": round(fill, 4),
           "ms_per_step_below_half_full": round(sum(1 for s in share if s < 0.5) * a.bin / 1e3 / a.keep, 3),
           "fill_histogram_tenths": {str(k): round(v / nb, 4) for k, v in sorted(hist.items())},
           "kernels_while_below_half_full_us_per_step": [(k, round(v / a.keep, 1)) for k, v in low.most_common(25)]}
    txt = json.dumps(rep, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
