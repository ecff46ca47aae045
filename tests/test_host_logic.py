"""CPU tests of the host logic: flags mirror, data helpers, and the data-parallel collective path with
world_size=2 over gloo (the RCCL path is the same code with backend "nccl")."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flags_mirror_defaults():
    from unsupervised_detection_amd.config import FLAGS, parse_flags
    assert (FLAGS.img_height, FLAGS.img_width, FLAGS.batch_size) == (192, 384, 16)
    assert (FLAGS.flow_normalizer, FLAGS.cbn, FLAGS.epsilon, FLAGS.beta1) == (80.0, 0.5, 75.0, 0.9)
    assert (FLAGS.iters_rec, FLAGS.iters_gen, FLAGS.summary_freq, FLAGS.max_epochs) == (1, 3, 30, 40)
    f = parse_flags(["--batch_size", "4", "--cbn", "1.0"])
    assert f.batch_size == 4 and f.cbn == 1.0 and f.dataset == "DAVIS2016"


def test_synthetic_pairs_shape_and_motion():
    from unsupervised_detection_amd import data
    a, b = data.synthetic_davis_pairs(1, 3, 96, 128)
    assert a.shape == (1, 96, 128, 3) and a.dtype == np.uint8 and b.shape == a.shape
    a2, _ = data.synthetic_davis_pairs(1, 3, 96, 128)
    assert np.array_equal(a, a2)  # seeded
    assert 1.0 < np.abs(a.astype(np.float32) - b.astype(np.float32)).mean() < 60.0  # moved, not unrelated


def test_schedule_matches_reference():
    """adversarial_learner.py:382-389 with iters_rec=1, iters_gen=3: steps 4,8,.. train the recover."""
    sched = ["rec" if (s % 4) < 1 else "gen" for s in range(1, 9)]
    assert sched == ["gen", "gen", "gen", "rec", "gen", "gen", "gen", "rec"]


def _dp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unsupervised_detection_amd.trainer import allreduce_mean_
    from oracle import oracle_torch as O
    # each rank holds the gradient of its local batch; the mean must equal the global-batch gradient
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(1000, generator=g)
    buf = local.clone()
    allreduce_mean_(buf)
    # identical clip + Adam on every rank keeps replicas bit-identical
    w = torch.ones(1000)
    opt = O.TFAdam()
    clipped, _ = O.clip_or_noise({"x": buf}, 0.2, False)
    p = {"x": w}
    opt.apply(p, clipped)
    q.put((rank, buf.numpy().copy(), p["x"].numpy().copy(), local.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_allreduce_mean_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    mean = (res[0][3] + res[1][3]) / 2
    assert np.allclose(res[0][1], mean, atol=1e-7) and np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(res[0][2], res[1][2])  # replicas stay bit-identical after the update
