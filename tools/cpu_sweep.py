#!/usr/bin/env python3
"""Thread sweep of bench.py's `cpu_baseline` leg: the PyTorch-CPU oracle (oracle/oracle_torch.py, kind "port") timed on full adversarial
steps at 8 / 16 / 32 / 64 / 128 threads of the GPU box's host, same seeded weights and frame pairs as bench.py.  Backs the "16 threads:
more are slower" choice of the bench line with numbers (profiles/rNN_cpu_threads.txt).  No GPU is touched.

  python tools/cpu_sweep.py [--threads 8,16,32,64,128] [--reps 2] [--batch 4]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="8,16,32,64,128")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    from oracle import oracle_torch as O
    from unsupervised_detection_amd import weights as W
    w = tuple(W.as_dict(W.init_flat(n, 8964), n) for n in (W.NET_PWC, W.NET_GEN, W.NET_REC))
    g = torch.Generator().manual_seed(8964)  # reader-shaped pairs (384x640, [-0.5, 0.5]); the content does not matter for the timing
    i1 = torch.rand(a.batch, 384, 640, 3, generator=g) - 0.5
    i2 = (i1 + 0.02 * torch.randn(a.batch, 384, 640, 3, generator=g)).clamp(-0.5, 0.5)

    class C(O.Flags):
        batch_size = a.batch
    rows = []
    for th in [int(t) for t in a.threads.split(",")]:
        torch.set_num_threads(th)
        pp, pg, pr = ({k: v.clone() for k, v in d.items()} for d in w)
        opt = O.TFAdam()
        times = []
        for r in range(a.reps + 1):
            for d in (pg, pr):
                for k in d:
                    d[k] = d[k].detach().requires_grad_(True)
            t0 = time.time()
            with torch.no_grad():
                image, flow, _ = O.prepare_inputs(pp, i1, i2, C)
            out = O.forward_from_flow(pg, pr, image, flow, C)
            gg = O.grads_of(out["generator"], pg)
            gr = O.grads_of(out["recover"], pr)
            with torch.no_grad():
                cg, _ = O.clip_or_noise(gg, 0.2, True, lambda k, s: torch.rand(s) * 0.4 - 0.2)
                cr, _ = O.clip_or_noise(gr, 0.2, False)
                pgd = {k: v.detach() for k, v in pg.items()}
                prd = {k: v.detach() for k, v in pr.items()}
                opt.apply(pgd, cg)
                opt.apply(prd, cr)
                pg, pr = pgd, prd
            if r > 0:
                times.append(time.time() - t0)
        times.sort()
        med = times[len(times) // 2]
        rows.append({"threads": th, "s_per_step": round(med, 3), "frame_pairs_per_s": round(a.batch / med, 3)})
        print(json.dumps(rows[-1]), flush=True)
    best = max(rows, key=lambda r: r["frame_pairs_per_s"])
    print(json.dumps({"host_cpu_count": os.cpu_count(), "batch": a.batch, "reps": a.reps, "best": best,
                      "workload": "full adversarial step (PWC fwd + gen fwd + 3x recover fwd + both backward + clipped Adam), 384x640 -> 192x384"}))


if __name__ == "__main__":
    main()
