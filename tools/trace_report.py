#!/usr/bin/env python3
"""Per-step kernel table from a rocprofv3 kernel trace of TIMED STEPS ONLY.

  # 1. a tuning run writes the configurations it picked (and the per-launch CSV of its measurement pass)
  UDET_PROF_DUMP=gpurun_out/layers.csv python bench.py --tune-cache gpurun_out/tune.txt > gpurun_out/bench.json
  # 2. a second process loads them (no tuning pass, no oracle leg, no measurement pass) and is traced
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/trace -o steps --output-format csv -- \
      python $REPO/bench.py --tune-cache $REPO/gpurun_out/tune.txt --trace-only --steps 20 --warmup 2
  # 3. this script
  python tools/trace_report.py gpurun_out/trace --steps 22 --layers gpurun_out/layers.csv --out profiles/rNN_trace_report.json \
      --copy-stats profiles/rNN_kernel_stats_steps.csv

The trace holds plan construction (a few pack / memset kernels), 2 warm-up and 20 timed steps -- every step launches the same
kernels, so totals / 22 are per-step figures.  The report sums the convolution kernels' time per step and divides the executed
GFLOP of one step (from the per-launch CSV) by it: the same quantity as `roofline.achieved` in the bench line, from an independent
clock (the profiler's dispatch timestamps).  Kernels run concurrently on the plan's side streams in this trace, so a kernel's
duration includes whatever it loses to its neighbours; the bench line's figure comes from a serial pass."""
import argparse
import csv
import glob
import json
import os
import shutil

CONV = ("conv_igemm", "conv_wino", "conv_tile", "conv_thin", "conv_wgrad", "wgrad_reduce", "conv_splitk_epilogue", "tap_gather", "bn_dot", "bn_finish")
PEAK = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace_dir")
    ap.add_argument("--steps", type=int, required=True, help="warm-up + timed steps in the trace")
    ap.add_argument("--layers", default="", help="per-launch CSV of the tuning run's measurement pass (executed GFLOP per step)")
    ap.add_argument("--out", default="")
    ap.add_argument("--copy-stats", default="")
    a = ap.parse_args()
    stats = sorted(glob.glob(os.path.join(a.trace_dir, "**", "*kernel_stats.csv"), recursive=True))
    if not stats:
        raise SystemExit("no *kernel_stats.csv under %s" % a.trace_dir)
    rows = list(csv.DictReader(open(stats[-1])))
    if a.copy_stats:
        shutil.copyfile(stats[-1], a.copy_stats)
    table, conv_ns, all_ns = [], 0.0, 0.0
    for r in rows:
        name, calls, tot = r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])
        if "lane_spin_kernel" in name or "lane_noop_kernel" in name:  # the one-off lane-placement probe (plan_exec.hip), not part of a step
            continue
        short = name.replace("void udet::", "").replace("(udet::ConvParams)", "").replace("udet::", "")
        is_conv = any(k in name for k in CONV)
        all_ns += tot
        if is_conv:
            conv_ns += tot
        table.append({"kernel": short[:110], "calls_per_step": round(calls / a.steps, 2), "us_per_step": round(tot / a.steps / 1e3, 2),
                      "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "conv": is_conv})
    table.sort(key=lambda t: -t["us_per_step"])
    exe = None
    if a.layers and os.path.exists(a.layers):
        exe = sum(float(r[3]) for r in csv.reader(open(a.layers)) if int(r[0]) < 3)
    rep = {"trace": os.path.basename(stats[-1]), "steps_in_trace": a.steps,
           "kernel_ms_per_step": round(all_ns / a.steps / 1e6, 3), "conv_kernel_ms_per_step": round(conv_ns / a.steps / 1e6, 3),
           "executed_gflop_per_step": exe,
           "conv_tflops": round(exe / (conv_ns / a.steps / 1e6), 2) if exe else None,
           "frac_of_fp32_mfma_peak": round(exe / (conv_ns / a.steps / 1e6) / PEAK, 4) if exe else None,
           "top_kernels": table[:25]}
    txt = json.dumps(rep, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
