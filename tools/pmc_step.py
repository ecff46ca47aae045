#!/usr/bin/env python3
"""PMC counters of WHOLE STEPS (every kernel of the adversarial step, not one hand-picked launch): HBM traffic and MFMA-pipe
occupancy per kernel family and for the ten longest dispatches, in separate rocprofv3 passes.

  cd /tmp && export TMPDIR=/tmp
  python $REPO/tools/pmc_step.py --tune-cache $REPO/gpurun_out/tune.txt --out $REPO/profiles/rNN_pmc_step.json

Each pass runs `python bench.py --tune-cache <file> --trace-only --no-pipeline --steps S --warmup 1` with UDET_SERIAL=1 (one
stream: dispatch order is the program order and identical in every pass) under `rocprofv3 --kernel-trace --pmc <group>`:
  pass 1  FETCH_SIZE                      (KB; on gfx950 the 128-byte requests of wide streaming reads count 64 bytes: x2,
  pass 2  WRITE_SIZE                       MI355X_MICROARCH.md, HBM / rocprofv3 section; FETCH and WRITE never share a pass)
  pass 3  SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE
`--pmc` is never combined with the system / HIP trace domains.  Per-step figures = totals / (S + 1)."""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]]


def short(name):
    return name.replace("void udet::", "").replace("(udet::ConvParams)", "").replace("udet::", "")[:100]


def one_pass(counters, tune_cache, steps):
    d = tempfile.mkdtemp(prefix="pmcstep_")
    env = dict(os.environ, UDET_SERIAL="1")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                                                                  os.path.join(ROOT, "bench.py"), "--tune-cache", tune_cache, "--trace-only",
                                                                  "--no-pipeline", "--steps", str(steps), "--warmup", "1"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
    disp = collections.OrderedDict()  # dispatch id -> {name, dur_us, counters}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = int(r["Dispatch_Id"])
            e = disp.setdefault(k, {"kernel": short(r["Kernel_Name"]), "dur_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [disp[k] for k in sorted(disp)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tune-cache", required=True)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    nsteps = a.steps + 1
    passes = [one_pass(g, a.tune_cache, a.steps) for g in GROUPS]
    n = min(len(p) for p in passes)
    merged = []
    for i in range(n):  # serial execution: the i-th dispatch is the same launch in every pass
        e = dict(passes[0][i])
        for p in passes[1:]:
            if p[i]["kernel"] != e["kernel"]:
                raise SystemExit("dispatch order differs between passes at %d: %s vs %s" % (i, p[i]["kernel"], e["kernel"]))
            e.update({k: v for k, v in p[i].items() if k not in ("kernel", "dur_us")})
            if "GRBM_GUI_ACTIVE" in p[i]:
                e["dur_us_clk"] = p[i]["dur_us"]  # the duration of the launch in the pass that counted its clock cycles
        e["dur_us"] = min(p[i]["dur_us"] for p in passes)
        merged.append(e)

    # Per-step figures come from whole step PERIODS: the dispatches between two occurrences of a once-per-step kernel (loss_finish_kernel;
    # a period holds every launch of a step exactly once).  What precedes the first marker is set-up -- the zero-fill of the 2.5 GB arena,
    # the re-layout of PWC-Net's weights -- and used to be averaged into the steps (2.5 GB per step of "traffic" that no step moves).
    marks = [i for i, e in enumerate(merged) if "loss_finish_kernel" in e["kernel"]]
    all_dispatches = len(merged)
    if len(marks) >= 2:
        merged = merged[marks[0] + 1:marks[-1] + 1]
        nsteps = len(marks) - 1

    def traffic(e):
        return e.get("FETCH_SIZE", 0.0) * 1024 * 2 + e.get("WRITE_SIZE", 0.0) * 1024

    def mfma_busy(e):  # SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs in quad-cycles; GRBM_GUI_ACTIVE over the 8 XCDs
        g = e.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        return round(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g * 1024), 3) if g > 0 else None
    def clock_ghz(e):  # effective shader clock while the launch ran (the chip clocks to its power budget: MI355X_MICROARCH.md, DVFS);
                       # meaningful for long launches only -- GRBM_GUI_ACTIVE also counts the cycles around a dispatch
        g, d = e.get("GRBM_GUI_ACTIVE", 0.0) / 8.0, e.get("dur_us_clk", 0.0)
        return round(g / (d * 1e3), 3) if g > 0 and d > 0 else None
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    for e in merged:
        f = fam[e["kernel"]]
        f["dispatches"] += 1
        f["us"] += e["dur_us"]
        f["hbm_bytes"] += traffic(e)
        f["mfma_busy_cycles"] += e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        f["gui_active"] += e.get("GRBM_GUI_ACTIVE", 0.0)
    families = sorted(({"kernel": k, "dispatches_per_step": round(v["dispatches"] / nsteps, 1), "us_per_step": round(v["us"] / nsteps, 1),
                        "hbm_MB_per_step": round(v["hbm_bytes"] / nsteps / 1e6, 2),
                        "mfma_pipe_busy_frac": round(v["mfma_busy_cycles"] / (v["gui_active"] / 8 * 1024), 3) if v["gui_active"] else None}
                       for k, v in fam.items()), key=lambda t: -t["us_per_step"])
    last = merged[len(merged) - len(merged) // nsteps:]  # the dispatches of the last step
    top = sorted(last, key=lambda e: -e["dur_us"])[:10]
    rep = {"command": "UDET_SERIAL=1 rocprofv3 --kernel-trace --pmc <group> -- python bench.py --tune-cache <file> --trace-only --no-pipeline "
                      "--steps %d --warmup 1 (one pass per counter group)" % a.steps,
           "step_periods_counted": nsteps, "dispatches_in_each_pass": all_dispatches, "dispatches_per_step": round(len(merged) / nsteps, 1),
           "hbm_traffic_MB_per_step": round(sum(traffic(e) for e in merged) / nsteps / 1e6, 1),
           "hbm_read_MB_per_step_corrected_x2": round(sum(e.get("FETCH_SIZE", 0.0) for e in merged) * 1024 * 2 / nsteps / 1e6, 1),
           "hbm_write_MB_per_step": round(sum(e.get("WRITE_SIZE", 0.0) for e in merged) * 1024 / nsteps / 1e6, 1),
           "kernel_us_per_step_under_pmc": round(sum(e["dur_us"] for e in merged) / nsteps, 1),
           "top10_dispatches_of_one_step": [{"kernel": e["kernel"], "us": round(e["dur_us"], 1), "hbm_MB": round(traffic(e) / 1e6, 2),
                                             "fetch_MB_x2": round(e.get("FETCH_SIZE", 0.0) * 2048 / 1e6, 2),
                                             "write_MB": round(e.get("WRITE_SIZE", 0.0) * 1024 / 1e6, 2), "mfma_pipe_busy_frac": mfma_busy(e),
                                             "effective_clock_GHz": clock_ghz(e)}
                                            for e in top],
           "kernel_families": families[:30],
           # the HBM-bound kernels of the north star are reported whatever their rank: every family of the fused warp + cost-volume
           # kernel, and each of its dispatches of one step (one per pyramid level, level 6 first) with its own bytes and duration
           "warp_cost_volume_families": [f for f in families if "warp_cost_volume" in f["kernel"] or f["kernel"].startswith(("warp_kernel", "cost_volume_kernel"))],
           "warp_cost_volume_dispatches_of_one_step": [{"kernel": e["kernel"], "us": round(e["dur_us"], 1),
                                                        "fetch_MB_x2": round(e.get("FETCH_SIZE", 0.0) * 2048 / 1e6, 3),
                                                        "write_MB": round(e.get("WRITE_SIZE", 0.0) * 1024 / 1e6, 3),
                                                        "hbm_MB": round(traffic(e) / 1e6, 3),
                                                        "GBs": round(traffic(e) / (e["dur_us"] * 1e-6) / 1e9, 1) if e["dur_us"] > 0 else None}
                                                       for e in last if "warp_cost_volume" in e["kernel"]]}
    wcv = rep["warp_cost_volume_dispatches_of_one_step"]
    rep["warp_cost_volume_MB_per_step"] = round(sum(d["hbm_MB"] for d in wcv), 3)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep, indent=1)[:6000])


if __name__ == "__main__":
    main()
