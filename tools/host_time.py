"""How long the HOST needs to enqueue one pipelined training step (no synchronisation inside the loop) against what the step takes
on the GPU: python tools/host_time.py (needs an MI355X and profiles/r04_tune.txt)."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsupervised_detection_amd import data
from unsupervised_detection_amd._ffi import lib
from unsupervised_detection_amd.engine import BOTH, Engine, EngineConfig
from unsupervised_detection_amd.trainer import TrainState, train_step
eng = Engine(EngineConfig(batch_size=4))
prio = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else None  # priorities of the three side streams (lanes 1 | 3 | 4, 5)
lib.udet_tune_load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r04_tune.txt').encode())
st = TrainState(eng, seed=8964, autotune=False)
batches = []
for i in range(4):
    f1, f2 = data.synthetic_davis_pairs(4, 8964 + 1000 * i)
    batches.append((data.preprocess_image(torch.from_numpy(f1).cuda()), data.preprocess_image(torch.from_numpy(f2).cuda())))
torch.cuda.synchronize()
torch.cuda.set_stream(torch.cuda.Stream())
if prio:
    print("stream priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
    side = [torch.cuda.Stream(priority=q) for q in prio]
    eng.pin_lanes(side)
    print("lanes", eng.lane_queues())
def step(i):
    a, b = batches[i % 4]; nx = batches[(i + 1) % 4]
    train_step(st, a, b, BOTH, next_pair=nx)
for i in range(8): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter(); host = []
for i in range(8, 48):
    h0 = time.perf_counter(); step(i); host.append(time.perf_counter() - h0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue per step ms: mean %.3f min %.3f max %.3f' % (1e3 * sum(host) / len(host), 1e3 * min(host), 1e3 * max(host)))
print('enqueue loop total %.3f ms/step; incl. final sync %.3f ms/step' % (1e3 * (t1 - t0) / 40, 1e3 * (t2 - t0) / 40))
