#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one step (launch-bound check)?  Prints enqueue ms vs GPU ms per step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsupervised_detection_amd import data
from unsupervised_detection_amd.engine import BOTH, Engine, EngineConfig
from unsupervised_detection_amd.trainer import TrainState, train_step

eng = Engine(EngineConfig(batch_size=4))
st = TrainState(eng, seed=8964, autotune=True)
f1, f2 = data.synthetic_davis_pairs(4, 8964)
img1 = data.preprocess_image(torch.from_numpy(f1).cuda())
img2 = data.preprocess_image(torch.from_numpy(f2).cuda())
for _ in range(5):
    train_step(st, img1, img2, BOTH)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    train_step(st, img1, img2, BOTH)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, total {1e3 * (t2 - t0) / n:.2f} ms/step")
# per phase enqueue cost
for name, fn in (("pack", lambda: eng.pack_trainable(st.w_gen, st.w_rec)), ("forward", lambda: eng.forward(img1, img2, 3)),
                 ("backward", lambda: eng.backward(BOTH, st.w_gen, st.w_rec, st.g_gen, st.g_rec))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: enqueue {1e3 * (t1 - t0):.2f} ms, gpu-complete {1e3 * (t2 - t0):.2f} ms")
