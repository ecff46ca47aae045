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


def _world1_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=0, world_size=1)
    from unsupervised_detection_amd.trainer import _dp_active, allreduce_mean_
    x = torch.arange(8, dtype=torch.float32)
    os.environ.pop("UDET_DP_WORLD1", None)
    off = _dp_active(None)
    os.environ["UDET_DP_WORLD1"] = "1"
    on = _dp_active(None)
    y = x.clone()
    allreduce_mean_(y)  # SUM over one rank, divided by one: the collective really runs
    q.put((off, on, _dp_active(False), bool(torch.equal(x, y))))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_world_size_one_exchanges_only_when_asked_to():
    """UDET_DP_WORLD1=1 (trainer._dp_active): a one-rank group issues its collectives -- how a one-GPU box executes the RCCL branch
    (tests/test_bench_gpu.py); without the flag a one-rank group exchanges nothing, and group=False never does."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_world1_worker, args=(29810 + os.getpid() % 150, q))
    p.start()
    off, on, never, same = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0
    assert off is False and on is True and never is False and same


def _tune_worker(rank, world, port, q, tune_file):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unsupervised_detection_amd._ffi import lib
    from unsupervised_detection_amd.trainer import TrainState

    class FakeEngine:  # rank 0's "autotune" fills the library's configuration cache from a file; any other rank tuning is a failure
        def autotune(self, *a):
            assert rank == 0, "only rank 0 may tune"
            return int(lib.udet_tune_load(tune_file.encode()))
    st = TrainState.__new__(TrainState)
    st.engine, st.w_gen, st.w_rec, st.g_gen, st.g_rec = FakeEngine(), None, None, None, None
    before = int(lib.udet_tuned_shapes())
    n = st._autotune_shared()
    q.put((rank, before, n, int(lib.udet_tuned_shapes())))
    dist.barrier()
    dist.destroy_process_group()


def test_autotune_is_shared_through_the_process_group(tmp_path):
    """trainer.TrainState._autotune_shared: rank 0 tunes, the text of udet_tune_save travels through the process group, every other
    rank loads it -- all ranks end up with the same configurations (host-only part of the C ABI: runs without a GPU)."""
    from unsupervised_detection_amd._ffi import lib
    g = tmp_path / "hdr.txt"
    assert lib.udet_tune_save(str(g).encode()) == 0
    hdr = g.read_text().splitlines()[0] + "\n"  # this build's header line (format version + tuning ABI)
    f = tmp_path / "tune.txt"
    f.write_text(hdr + "c 918273 128 64 2 2 0 0\nc 918274 64 64 1 4 0 0\nw 918275 7\n")  # (the process-global cache is only touched in the workers)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 200
    ps = [ctx.Process(target=_tune_worker, args=(r, 2, port, q, str(f))) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == 0 and res[1][1] == 0            # fresh processes: nothing tuned
    assert res[0][2] == res[1][2] and res[0][2] >= 2    # both report the same number of configurations ...
    assert res[0][3] == res[1][3] == res[0][2]          # ... and hold them


def _fake_davis(root, seqs=(("bear", 6), ("camel", 5), ("cows", 7))):
    os.makedirs(os.path.join(root, "ImageSets", "480p"), exist_ok=True)
    rows = []
    for name, n in seqs:
        for i in range(n):
            rows.append("/JPEGImages/480p/%s/%05d.jpg /Annotations/480p/%s/%05d.png" % (name, i, name, i))
    for part in ("train", "val", "trainval"):
        with open(os.path.join(root, "ImageSets", "480p", part + ".txt"), "w") as f:
            f.write("\n".join(rows) + "\n")


def test_reader_shards_pairs_across_ranks(tmp_path, monkeypatch):
    """Data-parallel input (cli.py): the ranks share ONE shuffle of the pair table and each takes its own rows of every global
    batch -- disjoint first frames within a global step, the table covered once per epoch (every rank used to
    draw from the full table on its own)."""
    from unsupervised_detection_amd import data
    _fake_davis(str(tmp_path))
    monkeypatch.setattr(data, "preprocess_image", lambda x, *a, **k: x)
    monkeypatch.setattr(data, "augment_pair", lambda a, b, crop, rng: (a, b))
    loader = lambda path, ch: np.zeros((2, 2, ch), np.uint8)
    world, bs = 2, 2
    its = [data.Davis2016Reader(str(tmp_path), max_temporal_len=2, min_temporal_len=1, num_threads=1, device="cpu", seed=5,
                                loader=loader, shard=(r, world)).image_inputs(batch_size=bs, partition="train") for r in range(world)]
    n_table = 2 * sum(n - 2 for n in (6, 5, 7))  # (first, +1) and (last, -1) rows of every sequence at t_len = 2
    steps = n_table // (bs * world)
    seen = []
    for _ in range(steps):
        names = [tuple(next(it)["fname"]) for it in its]
        assert not set(names[0]) & set(names[1])  # the ranks of one global step hold different pairs
        seen += [n for t in names for n in t]
    # one epoch = the shuffled table once (each frame appears at most twice: as a forward and as a backward pair)
    assert len(seen) == steps * bs * world and max(seen.count(n) for n in set(seen)) <= 2
    # a single rank keeps the old behaviour: one generator drives shuffle and draws
    one = data.Davis2016Reader(str(tmp_path), 2, 1, 1, "cpu", seed=5, loader=loader)
    assert one.rng is one.order_rng


def test_latest_checkpoint_accepts_reference_and_tf_files(tmp_path):
    """adversarial_learner.py:345-350: resume takes whatever the Saver left in checkpoint_dir."""
    from unsupervised_detection_amd.learner import _latest_checkpoint
    d = tmp_path / "ck"
    d.mkdir()
    assert _latest_checkpoint(str(d)) == ""
    (d / "model-3.index").write_bytes(b"")       # a reference-written Saver checkpoint
    (d / "model-3.data-00000-of-00001").write_bytes(b"")
    assert _latest_checkpoint(str(d)) == str(d / "model-3")
    (d / "model-7.tf.index").write_bytes(b"")    # --save_tf_checkpoint
    assert _latest_checkpoint(str(d)) == str(d / "model-7.tf")
    (d / "model-7").write_bytes(b"")             # the native torch file of the same epoch wins
    assert _latest_checkpoint(str(d)) == str(d / "model-7")
    (d / "model-12").write_bytes(b"")
    assert _latest_checkpoint(str(d)) == str(d / "model-12")
    e = tmp_path / "only_best"
    e.mkdir()
    (e / "model.best").write_bytes(b"")
    assert _latest_checkpoint(str(e)) == str(e / "model.best")
