#!/usr/bin/env python3
"""Collect the PMC counters of one convolution launch in separate rocprofv3 passes and assemble the JSON that bench.py reads
for `roofline.traffic` (profiles/r01_pmc_dc_conv21.json was produced this way).

  cd /tmp && export TMPDIR=/tmp
  python $REPO/tools/pmc_report.py --only dc_conv21 --cfg 131200,128,1 --kernel conv_igemm_dma_kernel --out $REPO/profiles/rNN_pmc_dc_conv21.json

One pass per counter group (FETCH_SIZE and WRITE_SIZE must not share a pass; `--pmc` is never combined with the system / HIP
trace domains).  FETCH_SIZE / WRITE_SIZE are reported in KB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts
the 128-byte requests of wide streaming reads at 64 bytes, so the read traffic is doubled."""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"],
          ["GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_LDS"], ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]]


def one_pass(counters, bench_args, kernel):
    d = tempfile.mkdtemp(prefix="pmc_")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "p", "--",
                                                                  sys.executable, os.path.join(ROOT, "tools", "conv_bench.py")] + bench_args
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out, dur = {}, []
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if kernel in r["Kernel_Name"]:
                out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    return {k: max(v) for k, v in out.items()}, dur  # warm-up launches of a forced config are identical: take any (max)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="dc_conv21")
    ap.add_argument("--cfg", default="131200,128,1")
    ap.add_argument("--kernel", default="conv_igemm_dma_kernel")
    ap.add_argument("--alg-bytes", type=float, default=173666304.0, help="algorithmic bytes of the launch (input + weights + output)")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    bench_args = ["--only", a.only, "--cfg", a.cfg, "--reps", "3"]
    res, durs = {"command": "rocprofv3 --kernel-trace --pmc <group> -- python tools/conv_bench.py " + " ".join(bench_args),
                 "kernel_filter": a.kernel}, []
    for g in GROUPS:
        vals, dur = one_pass(g, bench_args, a.kernel)
        res.update(vals)
        durs += dur
    if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
        rd, wr = res["FETCH_SIZE"] * 1024 * 2, res["WRITE_SIZE"] * 1024
        res.update({"FETCH_SIZE_KB": res.pop("FETCH_SIZE"), "WRITE_SIZE_KB": res.pop("WRITE_SIZE"), "hbm_read_bytes_corrected": rd,
                    "hbm_write_bytes": wr, "traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_total": a.alg_bytes,
                    "traffic_over_algorithmic": round((rd + wr) / a.alg_bytes, 3)})
    if durs:
        res["profiled_duration_us"] = sorted(durs)[len(durs) // 2]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in res and "GRBM_GUI_ACTIVE" in res and durs:
        res["effective_clock_GHz"] = round(res["GRBM_GUI_ACTIVE"] / (res["profiled_duration_us"] * 1e3) / 8, 3)  # summed over 8 XCDs
        res["mfma_pipe_busy_frac"] = round(res["SQ_VALU_MFMA_BUSY_CYCLES"] / (res["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)  # 256 CUs x 4 SIMDs
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
