"""GPU: the one-off autotune pass (udet_autotune) must leave the plan's results unchanged up to rounding --
after it, a full adversarial step still matches the CPU oracle within the 1e-3 north_star tolerance and the
workspace's activation regions are clean (zero padding channels)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle_torch as O  # noqa: E402


def test_autotuned_step_matches_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from unsupervised_detection_amd import weights as W
    from unsupervised_detection_amd.engine import BOTH, Engine, EngineConfig
    eng = Engine(EngineConfig(batch_size=2, in_height=128, in_width=192, img_height=64, img_width=128))
    pp, pg, pr = (O.init_params(O.pwc_param_specs(), 21), O.init_params(O.generator_param_specs(), 22),
                  O.init_params(O.recover_param_specs(), 23))
    w = {n: W.from_dict(p, n).cuda() for n, p in ((W.NET_PWC, pp), (W.NET_GEN, pg), (W.NET_REC, pr))}
    g = {n: torch.zeros_like(w[n]) for n in (W.NET_GEN, W.NET_REC)}
    eng.pack_pwc(w[W.NET_PWC])
    eng.pack_trainable(w[W.NET_GEN], w[W.NET_REC])
    n = eng.autotune(w[W.NET_GEN], w[W.NET_REC], g[W.NET_GEN], g[W.NET_REC])
    assert n > 20  # distinct convolution problems timed and cached
    # padding channels of the slabs are zero again (they are never written by the step)
    assert float(eng.buffer("pwc.slab3")[..., 529:532].abs().max()) == 0.0
    assert float(eng.buffer("gen.in")[..., 5:].abs().max()) == 0.0
    gen = torch.Generator().manual_seed(7)
    i1 = torch.rand(2, 128, 192, 3, generator=gen) - 0.5
    i2 = torch.rand(2, 128, 192, 3, generator=gen) - 0.5
    eng.forward(i1.cuda(), i2.cuda(), 3)
    eng.backward(BOTH, w[W.NET_GEN], w[W.NET_REC], g[W.NET_GEN], g[W.NET_REC])
    torch.cuda.synchronize()

    class C(O.Flags):
        img_height, img_width, batch_size = 64, 128, 2
    image, flow, _ = O.prepare_inputs(pp, i1, i2, C)
    gflow = eng.buffer("flow").cpu().clone()
    assert float((gflow - flow).abs().max()) < 1e-3 * max(1.0, float(flow.abs().max()))
    for d in (pg, pr):
        for k in d:
            d[k] = d[k].double().requires_grad_(True)
    out = O.forward_from_flow(pg, pr, image.double(), gflow.double(), C)
    L = eng.losses()
    for k in ("generator", "recover"):
        assert abs(L[k] - float(out[k])) < 1e-3 * max(1.0, abs(float(out[k]))), (k, L[k], float(out[k]))
    assert float((eng.buffer("mask").cpu() - out["mask"].float()).abs().max()) < 1e-3
    gr = O.grads_of(out["recover"], pr)
    ref = W.from_dict({k: v.float() for k, v in gr.items()}, W.NET_REC)
    err = float((g[W.NET_REC].cpu() - ref).abs().max())
    assert err < 1e-3 * max(1.0, float(ref.abs().max())), err


def test_pipelined_and_concurrent_steps_are_bit_identical_to_serial(monkeypatch):
    """Cross-step prefetch of the PWC flow + the multi-stream step only re-order independent work: three training steps
    must leave exactly the same weights as the same steps run strictly serially (UDET_SERIAL=1, no prefetch)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from unsupervised_detection_amd.engine import BOTH, Engine, EngineConfig
    from unsupervised_detection_amd.trainer import TrainState, train_step
    cfg = EngineConfig(batch_size=2, in_height=128, in_width=192, img_height=64, img_width=128)
    gen = torch.Generator().manual_seed(11)
    pairs = [((torch.rand(2, 128, 192, 3, generator=gen) - 0.5).cuda(), (torch.rand(2, 128, 192, 3, generator=gen) - 0.5).cuda())
             for _ in range(3)]

    def run(pipelined, concurrent=None):
        eng = Engine(cfg)
        if concurrent is not None:
            eng.set_concurrent(concurrent)  # udet_plan_set_concurrent: the documented switch (UDET_SERIAL only sets its initial value)
        st = TrainState(eng, seed=5)
        for i, (a, b) in enumerate(pairs):
            nxt = pairs[i + 1] if (pipelined and i + 1 < len(pairs)) else None
            train_step(st, a, b, BOTH, next_pair=nxt)
        torch.cuda.synchronize()
        return st.w_gen.clone(), st.w_rec.clone(), st.engine.losses()

    monkeypatch.setenv("UDET_SERIAL", "1")
    g0, r0, l0 = run(False)
    monkeypatch.setenv("UDET_SERIAL", "0")
    g1, r1, l1 = run(True)
    assert torch.equal(g0, g1) and torch.equal(r0, r1)
    assert l0 == l1
    g2, r2, l2 = run(False, concurrent=False)   # the same through the API, on a plan created concurrent
    g3, r3, l3 = run(True, concurrent=True)
    assert torch.equal(g0, g2) and torch.equal(r0, r2) and torch.equal(g0, g3) and torch.equal(r0, r3) and l0 == l2 == l3
