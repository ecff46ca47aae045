"""bench.py's multi-process path on a 1-GPU box: two ranks share cuda:0 and reduce over gloo (UDET_BENCH_ONE_GPU=1).
The numbers mean nothing; what is checked is that every rank issues the same sequence of collectives (a rank-0-only
step would deadlock the real RCCL run) and that rank 0 prints exactly one JSON line with the contract's fields."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_complete_and_report_once():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, UDET_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cycles", "1",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["config"]["global_batch"] == 8
    assert out["value"] > 0 and out["roofline"]["frac"] > 0 and out["reference_schedule"]["cycles"] == 1
    assert out["allreduce_ms"] is not None and out["allreduce_ms"] > 0 and out["parity_check"] is None
    # the exchange runs the same stream / event choreography as an RCCL run (trainer._exchange_gradients) and reports what the
    # overlap hides; per-step distribution and the timed-region roofline fraction are part of the line
    ex = out["gradient_exchange"]
    assert ex is not None and ex["ms_per_step_without_exchange"] > 0 and ex["overlap_hidden_ms"] >= 0
    assert out["ms_per_step_median"] > 0 and out["ms_per_step_p95"] >= out["ms_per_step_median"] and out["resident_batches"] >= 4
    assert out["roofline"]["frac_step"] > 0


def test_gpus_flag_without_a_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how a driver calls the N = 1 form) re-executes itself under
    torch.distributed.run: exactly one JSON line, from rank 0 of the children."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(UDET_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cycles", "0", "--no-cpu-baseline",
           "--ensemble-frames", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8 and out["allreduce_ms"] > 0


def test_rccl_branch_at_world_size_one():
    """The RCCL branch of bench.py / trainer._exchange_gradients executed on the ONE GPU of the box (UDET_DP_WORLD1=1):
    init_process_group("nccl", device_id=...), the recover gradients' all-reduce on the communication stream behind
    udet_stream_wait_grads, the generator gradients' on the compute stream, the event hand-over, the stand-alone exchange
    (allreduce_ms), barrier + destroy.  A group of one rank proves nothing about scaling; it proves that the calls, the stream
    semantics and the environment (HSA_ENABLE_IPC_MODE_LEGACY=0) are valid on this ROCm / RCCL stack before an 8-GPU node runs them
    (models/adversarial_learner.py:167-172 is the algebra the exchange implements: the mean over ranks, here over one)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "UDET_BENCH_ONE_GPU")}
    env.update(UDET_DP_WORLD1="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--cycles", "1", "--no-cpu-baseline",
           "--ensemble-frames", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["process_group"] == {"backend": "nccl", "world_size": 1, "forced_at_world_size_one": True}
    assert out["n_gpus"] == 1 and out["allreduce_ms"] is not None and out["allreduce_ms"] > 0
    ex = out["gradient_exchange"]
    assert ex is not None and ex["ms_per_step_without_exchange"] > 0
    # a one-rank mean changes nothing: the losses of the run are finite and the step time is that of the plain N = 1 run (loosely)
    assert all(v == v for v in out["losses"].values()) and out["ms_per_step"] < 40
    # round 6: the line names the library that produced it (the release build has no experiment knobs) and carries the algorithmic bytes
    assert out["release_library"] == {"file": "libudet.so", "experiment_knobs": "compiled out", "debug_hooks_loaded": False}
    assert out["roofline"]["alg_bytes_per_step"] > 4e9 and out["roofline"]["alg_bytes_forward"] > 2e9


def test_experiment_build_is_refused_and_labelled():
    """bench.py exits on a library that exports udet_exp_knob unless tools/knob_bench.py asked for it, and then says so in the line
    (no work-skipping switch behind a headline number).  Skipped where libudet_exp.so has not been built (`make exp`)."""
    import pytest
    if not os.path.exists(os.path.join(ROOT, "unsupervised_detection_amd", "libudet_exp.so")):
        pytest.skip("libudet_exp.so not built")
    env = dict(os.environ)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "knob_bench.py"), "--", "--steps", "2", "--warmup", "1", "--cycles", "0", "--no-cpu-baseline",
           "--ensemble-frames", "0", "--no-autotune"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["release_library"]["experiment_knobs"].startswith("PRESENT")
    # the same library through bench.py's own argument list (no --allow-experiment-build): refused
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import knob_bench; knob_bench.load_experiment_build(); import bench; "
            "sys.argv = ['bench.py', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--cycles', '0', '--ensemble-frames', '0']; bench.main()"
            % (ROOT, os.path.join(ROOT, "tools")))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "not the release library" in (r.stderr + r.stdout)
