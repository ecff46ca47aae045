"""Device-side training state + the data-parallel step.

One process per GPU.  Every op of the path is per-sample except the two batch means of the losses
(models/adversarial_learner.py:167-172,184,191), so grad(global batch B*R) = mean_r grad(local batch B):
the only exchange is the all-reduce(avg) of the flat gradient buffer(s) the step computed (RCCL over xGMI when
the process group backend is "nccl"; gloo in the CPU tests of the host logic), followed by the identical
clip / escape-noise / Adam on every rank (noise from a counter-based stream keyed by (seed, step, index))."""
from __future__ import annotations

import torch

from . import weights as W
from .engine import BOTH, GEN, REC, Engine


class TrainState:
    """Flat fp32 parameter / gradient / Adam-slot buffers of the two trainable networks + frozen PWC-Net."""

    def __init__(self, engine: Engine, seed: int = 8964, w_pwc=None, w_gen=None, w_rec=None, autotune: bool = False):
        dev = engine.device
        self.engine = engine
        self.w_pwc = (w_pwc if w_pwc is not None else W.init_flat(W.NET_PWC, seed)).to(dev)
        self.w_gen = (w_gen if w_gen is not None else W.init_flat(W.NET_GEN, seed)).to(dev)
        self.w_rec = (w_rec if w_rec is not None else W.init_flat(W.NET_REC, seed)).to(dev)
        # both gradient buffers are views of ONE allocation (the recover part starts on a 256-byte boundary).  A step that computes
        # both gradients still exchanges them in TWO collectives (_exchange_gradients): the recover gradients are final long before
        # the generator-loss pass ends, so their all-reduce runs on a communication stream under the rest of that pass
        n_gen, n_rec = self.w_gen.numel(), self.w_rec.numel()
        off_rec = (n_gen + 63) // 64 * 64
        self.g_all = torch.zeros(off_rec + n_rec, dtype=torch.float32, device=dev)
        self.g_gen, self.g_rec = self.g_all[:n_gen], self.g_all[off_rec:off_rec + n_rec]
        self.m_gen, self.v_gen = torch.zeros_like(self.w_gen), torch.zeros_like(self.w_gen)
        self.m_rec, self.v_rec = torch.zeros_like(self.w_rec), torch.zeros_like(self.w_rec)
        engine.pack_pwc(self.w_pwc)
        engine.pack_trainable(self.w_gen, self.w_rec)
        self._dirty = 0  # networks whose flat weights changed since their last re-layout (GEN | REC bits)
        if autotune:
            self.tuned_shapes = self._autotune_shared()

    def _autotune_shared(self) -> int:
        """The one-off kernel autotune.  In a data-parallel job rank 0 tunes and every other rank loads ITS configurations (the text of
        udet_tune_save, broadcast through the process group -- a collective: every rank must construct its TrainState with autotune=True):
        all GPUs then run the same kernels.  Independent tuning passes pick different tiles for a few shapes; the slowest rank is what a
        step costs, and the ranks' local gradients would differ in their rounding."""
        import os
        import tempfile

        import torch.distributed as dist

        from ._ffi import lib
        e = self.engine
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return e.autotune(self.w_gen, self.w_rec, self.g_gen, self.g_rec)
        text, n = [None], 0
        fd, path = tempfile.mkstemp(prefix="udet_tune_", suffix=".txt")
        os.close(fd)
        try:
            if dist.get_rank() == 0:
                n = e.autotune(self.w_gen, self.w_rec, self.g_gen, self.g_rec)
                lib.udet_tune_save(path.encode())
                with open(path) as f:
                    text[0] = f.read()
            dist.broadcast_object_list(text, src=0)
            if dist.get_rank() != 0:
                with open(path, "w") as f:
                    f.write(text[0])
                n = int(lib.udet_tune_load(path.encode()))
                if n < 0:  # (cannot happen between ranks of one build; tune locally rather than run untuned)
                    n = e.autotune(self.w_gen, self.w_rec, self.g_gen, self.g_rec)
        finally:
            os.remove(path)
        return n


def flush_weights(st: TrainState):
    """train_step defers the re-layout of the weights an optimizer apply changed to the NEXT step.  Anything else that reads
    the packed layout in between (a stand-alone forward: validation, inference on the training state) must see the weights
    that were just trained -- and that save() writes: flush the pending re-layout first."""
    dirty = getattr(st, "_dirty", 0)
    if dirty:
        st.engine.pack_trainable(st.w_gen if dirty & GEN else None, st.w_rec if dirty & REC else None)
    st._dirty = 0


def _dp_active(group) -> bool:
    """Is there a gradient exchange to do?  No process group, group=False (a process that trains alone although a group exists) or a
    group of ONE rank: no.  UDET_DP_WORLD1=1 lifts the last shortcut: a one-rank group then issues its (trivial) collectives through
    exactly the calls, streams and events an N-rank job uses -- how a one-GPU box executes the RCCL branch (`backend="nccl"`,
    tests/test_bench_gpu.py::test_rccl_branch_at_world_size_one) before an 8-GPU node ever does."""
    import os

    import torch.distributed as dist
    if group is False or not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("UDET_DP_WORLD1") == "1"


def allreduce_mean_(t: torch.Tensor, group=None, async_op=False):
    """In-place mean over the data-parallel group (no-op without an initialised process group)."""
    import torch.distributed as dist
    if not _dp_active(group):
        return None  # group=False: this process trains alone even though a process group exists (reference runs of the tests)
    # SUM + one in-place division on every backend (gloo has no AVG; on RCCL the extra elementwise pass over the 19 MB payload costs
    # ~10 us and keeps ONE code path that the one-GPU gloo tests execute end to end).  Every rank divides the same sum by the same
    # count: replicas stay bit-identical.
    work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=False)
    t.div_(dist.get_world_size(group))
    return work if async_op else None


def _exchange_gradients(st: TrainState, which: int, group):
    """grad(global batch) = mean over ranks of grad(local batch): the step's only collective(s).

    The stream / event choreography is the SAME for every backend -- only the collective inside `allreduce_mean_` differs (RCCL
    "nccl" on the GPU box, gloo in the one-GPU tests), so tests/test_dp_gpu.py executes exactly the code an 8-GPU run does:
      which = BOTH: a communication stream waits for the completion event of the RECOVER gradients alone
        (udet_stream_wait_grads: udet_backward records it where g_rec is final, before the caller's stream joins the longer
        generator-loss pass) and reduces g_rec there -- under the rest of the generator-loss pass;  the generator gradients are
        reduced on the compute stream once the whole backward is done;  the compute stream then waits for the communication
        stream.  Both collectives are issued in the same order on every rank.
      which = REC / GEN: one collective on the compute stream over what the step computed."""
    if not _dp_active(group):
        return
    e = st.engine
    if which != BOTH:
        allreduce_mean_(st.g_rec if which & REC else st.g_gen, group)
        return
    from ._ffi import check, lib
    comm = getattr(st, "_comm_stream", None)
    if comm is None:
        comm = st._comm_stream = torch.cuda.Stream(device=e.device)
    main = torch.cuda.current_stream(e.device)
    check(lib.udet_stream_wait_grads(e._h, W.NET_REC, comm.cuda_stream))
    with torch.cuda.stream(comm):
        allreduce_mean_(st.g_rec, group)  # (gloo: the host blocks here until g_rec is final and reduced; the generator-loss
        done = comm.record_event()        #  pass is already enqueued and keeps running on the device)
    allreduce_mean_(st.g_gen, group)
    main.wait_event(done)  # the optimizer applies follow on the compute stream


def exchange_alone(st: TrainState, group=None):
    """The collectives of a which=BOTH step with nothing to overlap with (bench.py: `allreduce_ms`): g_rec on the communication
    stream, g_gen on the compute stream, the compute stream then waits -- the streams and the order of _exchange_gradients; the
    communication stream starts behind the compute stream instead of behind the recover-gradient event."""
    if not _dp_active(group):
        return
    e = st.engine
    comm = getattr(st, "_comm_stream", None)
    if comm is None:
        comm = st._comm_stream = torch.cuda.Stream(device=e.device)
    main = torch.cuda.current_stream(e.device)
    comm.wait_stream(main)
    with torch.cuda.stream(comm):
        allreduce_mean_(st.g_rec, group)
        done = comm.record_event()
    allreduce_mean_(st.g_gen, group)
    main.wait_event(done)


def train_step(st: TrainState, img1, img2, which: int = BOTH, group=None, next_pair=None):
    """One adversarial step on this rank's frame pairs: PWC flow + generator fwd + 3x recover fwd + the
    backward pass(es) of `which` + gradient all-reduce + clipped Adam.
    which=REC / GEN reproduce train_recover_op / train_generator_op (adversarial_learner.py:224-234);
    which=BOTH computes both gradients from one forward (the benchmark's "both backward" step).
    next_pair=(img1', img2'): cross-step pipelining -- the frozen PWC-Net's flow of the NEXT pair is enqueued on side
    streams right after this step's forward and overlaps its backward; the next call (which must be given that same
    pair) then starts from the prefetched flow.  Every step still does all of its work, one step earlier for PWC."""
    e = st.engine
    # re-layout of whichever network the PREVIOUS steps updated (every forward reads both networks, so a recover step right
    # after a generator step must see the new generator too)
    flush_weights(st)

    def call(fn, *a):
        # conv_fp16 plans: a forward / backward call that finds an overflow report of an EARLIER optimizer update (dropped on the device,
        # weights untouched) raises before it enqueues anything (include/udet.h, udet_config.conv_fp16).  Like a skipped step of dynamic
        # loss scaling: count it, say so, and issue the call again -- training continues on the unchanged weights.
        if not getattr(e.cfg, "conv_fp16", False):  # (fp32 plans cannot raise it: no retry wrapper at all)
            return fn(*a)
        from ._ffi import UdetOverflow
        for _ in range(4):
            try:
                return fn(*a)
            except UdetOverflow as ex:
                st.overflow_skipped = getattr(st, "overflow_skipped", 0) + 1
                import sys
                print("[udet] fp16 overflow: an optimizer update was dropped (%d so far): %s" % (st.overflow_skipped, ex), file=sys.stderr)
        return fn(*a)

    if getattr(st, "_prefetched", None) is not None:
        p1, p2 = st._prefetched
        if p1.data_ptr() != img1.data_ptr() or p2.data_ptr() != img2.data_ptr():
            raise ValueError("train_step: this step's pair is not the one prefetched by the previous call")
        e.prefetch_consume()
        st._prefetched = None
        if next_pair is not None:  # fork the next pair's PWC flow here: it then overlaps this step's forward as well
            e.prefetch_flow(next_pair[0], next_pair[1])
            st._prefetched = (next_pair[0], next_pair[1])
        call(e.forward_in_place, 3)
    else:
        call(e.forward, img1, img2, 3)
        if next_pair is not None:
            e.prefetch_flow(next_pair[0], next_pair[1])
            st._prefetched = (next_pair[0], next_pair[1])
    # one call: with BOTH the two backward passes run concurrently on the plan's side streams (udet_backward)
    call(e.backward, which, st.w_gen, st.w_rec, st.g_gen, st.g_rec)
    _exchange_gradients(st, which, group)
    if which & GEN:
        e.apply(W.NET_GEN, st.w_gen, st.g_gen, st.m_gen, st.v_gen)
    if which & REC:
        e.apply(W.NET_REC, st.w_rec, st.g_rec, st.m_rec, st.v_rec)
    st._dirty = which & (GEN | REC)
