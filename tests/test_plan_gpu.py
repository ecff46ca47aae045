"""GPU parity of the step plan (C-ABI udet_plan_*) against the CPU oracle on the same seeded inputs.
Floating-point tolerance: 1e-3 (BASELINE.json north_star), relative to the tensor's scale."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle_torch as O  # noqa: E402


def _perturbed(specs, seed):
    p = O.init_params(specs, seed)
    g = torch.Generator().manual_seed(seed)
    for k in p:
        if k.endswith(("bias", "biases", "beta")):
            p[k] = torch.randn(p[k].shape, generator=g) * 0.05
        if k.endswith("gamma"):
            p[k] = 1 + torch.randn(p[k].shape, generator=g) * 0.1
    return p


class Cfg(O.Flags):
    img_height, img_width, batch_size = 64, 128, 2


# "decoder-lowres-backward": the recover decoder's low-resolution ("up-conv algebra") form for the backward-data pass of levels 1-3 as
# well -- a plan of this size would take it for the forward pass only (plan_build.hip: too few source pixels to be worth it)
@pytest.fixture(scope="module", params=["default", "decoder-lowres-backward"])
def env(request):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from unsupervised_detection_amd import weights as W
    from unsupervised_detection_amd._devel import dbg
    from unsupervised_detection_amd.engine import Engine, EngineConfig
    dbg.udet_debug_upb_min_pixels(0 if request.param == "decoder-lowres-backward" else -1)
    try:
        eng = Engine(EngineConfig(batch_size=2, in_height=128, in_width=192, img_height=64, img_width=128))
    finally:
        dbg.udet_debug_upb_min_pixels(-1)
    pp, pg, pr = _perturbed(O.pwc_param_specs(), 11), _perturbed(O.generator_param_specs(), 12), _perturbed(O.recover_param_specs(), 13)
    flat = {"pwc": W.from_dict(pp, W.NET_PWC).cuda(), "gen": W.from_dict(pg, W.NET_GEN).cuda(), "rec": W.from_dict(pr, W.NET_REC).cuda()}
    eng.pack_pwc(flat["pwc"])
    eng.pack_trainable(flat["gen"], flat["rec"])
    g = torch.Generator().manual_seed(5)
    base = torch.rand(2, 128 + 8, 192 + 8, 3, generator=g)
    # smooth-ish pair: second frame is the first shifted by (2,3) px plus noise
    img1 = torch.nn.functional.avg_pool2d(base.permute(0, 3, 1, 2), 5, 1, 2).permute(0, 2, 3, 1).contiguous()
    img2 = (img1[:, 2:130, 3:195] + 0.01 * torch.randn(2, 128, 192, 3, generator=g)).contiguous() - 0.5
    img1 = img1[:, 4:132, 4:196].contiguous() - 0.5
    return dict(eng=eng, W=W, pp=pp, pg=pg, pr=pr, flat=flat, img1=img1, img2=img2)


def rel_err(a, ref):
    return float((a - ref).abs().max()) / max(1e-6, float(ref.abs().max()))


def test_pwc_forward(env):
    eng = env["eng"]
    flow = eng.pwc_forward(env["img1"].cuda(), env["img2"].cuda()).cpu()
    ref, pyr = O.pwc_forward(env["pp"], env["img1"], env["img2"])
    assert flow.shape == ref.shape
    # intermediate check: the level-2 refined flow
    fr2 = eng.buffer("pwc.rflow2")[..., :2].cpu()
    assert rel_err(fr2, pyr[-1]) < 1e-3
    assert rel_err(flow, ref) < 1e-3


def test_forward_and_losses(env):
    eng = env["eng"]
    eng.forward(env["img1"].cuda(), env["img2"].cuda(), 3)
    image, flow, _ = O.prepare_inputs(env["pp"], env["img1"], env["img2"], Cfg)
    assert rel_err(eng.buffer("image").cpu(), image) < 1e-5
    assert rel_err(eng.buffer("flow").cpu(), flow) < 1e-3
    # feed the oracle the GPU's own flow so that PWC rounding does not leak into the comparison below
    gflow = eng.buffer("flow").cpu().clone()
    out = O.forward_from_flow(env["pg"], env["pr"], image, gflow, Cfg)
    assert float((eng.buffer("mask").cpu() - out["mask"]).abs().max()) < 1e-3
    pred = eng.buffer("pred").cpu()
    ref = torch.cat([out["pred"], out["pred_c"], out["pred_img"]], 0)
    assert rel_err(pred, ref) < 1e-3
    L = eng.losses()
    for k in L:
        assert abs(L[k] - float(out[k])) < 1e-3 * max(1.0, abs(float(out[k]))), (k, L[k], float(out[k]))


def _rand_inputs(seed=21):
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(2, 64, 128, 3, generator=g) - 0.5
    flow = torch.randn(2, 64, 128, 2, generator=g) * 0.1
    flow = torch.nn.functional.avg_pool2d(flow.permute(0, 3, 1, 2), 7, 1, 3).permute(0, 2, 3, 1).contiguous() * 3
    return image, flow


def test_test_graph_single_call(env):
    """build_test_graph (adversarial_learner.py:450-523): generator + ONE recover call."""
    eng = env["eng"]
    image, flow = _rand_inputs()
    eng.forward_from_flow(image.cuda(), flow.cuda(), 1)
    m = O.generator_net(env["pg"], image, O.preprocess_flow_batch(flow))
    pred = O.recover_net(env["pr"], image, flow * (1 - m), m)
    assert float((eng.buffer("mask").cpu() - m).abs().max()) < 1e-3
    assert rel_err(eng.buffer("pred").cpu()[:2], pred) < 1e-3


def test_backward_matches_oracle_autograd(env):
    eng, W = env["eng"], env["W"]
    image, flow = _rand_inputs()
    eng.forward_from_flow(image.cuda(), flow.cuda(), 3)
    g_gen = torch.zeros(W.param_total(W.NET_GEN), device="cuda")
    g_rec = torch.zeros(W.param_total(W.NET_REC), device="cuda")
    eng.backward(3, env["flat"]["gen"], env["flat"]["rec"], g_gen, g_rec)
    # float64 oracle
    pg = {k: v.double().requires_grad_(True) for k, v in env["pg"].items()}
    pr = {k: v.double().requires_grad_(True) for k, v in env["pr"].items()}
    out = O.forward_from_flow(pg, pr, image.double(), flow.double(), Cfg)
    gg = O.grads_of(out["generator"], pg)
    gr = O.grads_of(out["recover"], pr)
    for net, got, ref in ((W.NET_GEN, g_gen.cpu(), gg), (W.NET_REC, g_rec.cpu(), gr)):
        d = W.as_dict(got, net)
        scale = max(float(v.abs().max()) for v in ref.values())
        worst = 0.0
        for k, v in ref.items():
            err = float((d[k].double() - v).abs().max())
            tol = 1e-3 * max(float(v.abs().max()), 1e-3 * scale)
            worst = max(worst, err / tol)
            assert err < tol, (k, err, float(v.abs().max()))
        assert worst < 1.0


def test_apply_matches_tf_adam_with_shared_powers(env):
    eng, W = env["eng"], env["W"]
    eng.adam_step = 0
    opt = O.TFAdam(beta1=0.9)
    res = {}
    for net, key, seed in ((W.NET_GEN, "gen", 31), (W.NET_REC, "rec", 32), (W.NET_GEN, "gen", 33)):
        n = W.param_total(net)
        g = torch.Generator().manual_seed(seed)
        grad = torch.randn(n, generator=g) * 0.3  # many entries beyond the +-0.2 clip
        if key not in res:
            res[key] = dict(w=env["flat"][key].clone(), m=torch.zeros(n, device="cuda"), v=torch.zeros(n, device="cuda"),
                            p={"x": env["flat"][key].cpu().clone()})
        st = res[key]
        gd = grad.cuda()
        eng.apply(net, st["w"], gd, st["m"], st["v"])
        clipped, changed = O.clip_or_noise({"x": grad}, 0.2, net == W.NET_GEN)
        assert not changed and torch.equal(gd.cpu(), clipped["x"])
        sub = O.TFAdam.__new__(O.TFAdam)  # per-net slot view sharing the global powers
        params = st["p"]
        if "opt" not in st:
            st["opt"] = {"m": {}, "v": {}}
        opt.m, opt.v = st["opt"]["m"], st["opt"]["v"]
        opt.apply(params, clipped)
        assert float((st["w"].cpu() - params["x"]).abs().max()) < 1e-6
    assert eng.adam_step == 3


def test_escape_noise_branch(env):
    """loss_utils.py:19-26: mean_v(mean|g_v|) < 1e-5 -> every gradient <- abs(U(-0.2,0.2)); same stream for equal seeds."""
    eng, W = env["eng"], env["W"]
    n = W.param_total(W.NET_GEN)
    outs = []
    for _ in range(2):
        eng.adam_step = 7
        w = env["flat"]["gen"].clone()
        g = torch.full((n,), 1e-7, device="cuda")
        m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        eng.apply(W.NET_GEN, w, g, m, v)
        assert float(eng.buffer("noise_flag").view(-1)[1]) == 1.0
        outs.append(g.cpu())
    assert torch.equal(outs[0], outs[1])
    assert float(outs[0].min()) >= 0.0 and float(outs[0].max()) <= 0.2
    assert abs(float(outs[0].mean()) - 0.1) < 2e-3  # |U(-.2,.2)| is U(0,.2)
    # host replica of the counter-based stream (splitmix64 of seed, step, index)
    M = (1 << 64) - 1
    def u01(seed, step, idx):
        z = (seed * 0x9E3779B97F4A7C15 + step * 0xBF58476D1CE4E5B9 + idx + 0x94D049BB133111EB) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
        return np.float32(z >> 40) * np.float32(1.0 / 16777216.0)
    for idx in (0, 1, 12345, n - 1):
        exp = abs((u01(8964, 8, idx) * np.float32(2) - np.float32(1)) * np.float32(0.2))
        assert abs(float(outs[0][idx]) - float(exp)) < 1e-7
    # recover never takes the branch (can_change=False)
    nr = W.param_total(W.NET_REC)
    g = torch.full((nr,), 1e-7, device="cuda")
    eng.apply(W.NET_REC, env["flat"]["rec"].clone(), g, torch.zeros(nr, device="cuda"), torch.zeros(nr, device="cuda"))
    assert float(g.max()) <= 1.0000001e-7
