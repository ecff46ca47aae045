"""GPU parity at BASELINE.json configs[1] -- the shape bench.py measures: B = 4 frame pairs, 384x640 (PWC-Net) -> 192x384
(generator / recover), AFTER udet_autotune has picked the kernels, tiles and split-K counts for exactly these problem shapes.

Everything the adversarial step produces is compared with the CPU oracle on the same seeded weights / inputs at the
north_star tolerance (1e-3 of the tensor's scale): PWC flow, image / flow resizes, mask, the three recover predictions, the 8
losses{} entries, every parameter gradient of both networks for the joint backward (which = 3) AND for the two train ops the
reference actually runs -- the generator-loss backward alone (which = GEN) and the recover-loss backward alone (which = REC)
(models/adversarial_learner.py:224-234,376-397; loss_utils.py:12-32) --, one optimizer apply on those gradients, and the
4-crop augmented test graph at B = 1 (adversarial_learner.py:525-592)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle_torch as O  # noqa: E402

B, IN_H, IN_W, H, W_ = 4, 384, 640, 192, 384


def _perturbed(specs, seed):
    """reference initializers + non-trivial biases / BN parameters (zero-initialised ones would hide whole terms)"""
    p = O.init_params(specs, seed)
    g = torch.Generator().manual_seed(seed)
    for k in p:
        if k.endswith(("bias", "biases", "beta")):
            p[k] = torch.randn(p[k].shape, generator=g) * 0.05
        if k.endswith("gamma"):
            p[k] = 1 + torch.randn(p[k].shape, generator=g) * 0.1
    return p


def _smooth_pair(n, h, w, seed, shift=(3, 5)):
    """a textured frame and a copy displaced by `shift` pixels plus noise, in [-0.5, 0.5]"""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(n, h + 16, w + 16, 3, generator=g)
    base = torch.nn.functional.avg_pool2d(base.permute(0, 3, 1, 2), 5, 1, 2).permute(0, 2, 3, 1)
    i1 = base[:, 8:8 + h, 8:8 + w].contiguous() - 0.5
    i2 = (base[:, 8 - shift[0]:8 - shift[0] + h, 8 - shift[1]:8 - shift[1] + w] + 0.01 * torch.randn(n, h, w, 3, generator=g)).contiguous() - 0.5
    return i1, i2


def _noise_stream(seed, step, n):
    """host replica of the escape-noise stream (elementwise.hip udet_uniform01: splitmix64 of (seed, step, index)) -> U(-0.2, 0.2)"""
    import numpy as np
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(step) * np.uint64(0xBF58476D1CE4E5B9)
             + np.arange(n, dtype=np.uint64) + np.uint64(0x94D049BB133111EB)) & M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M
        z ^= z >> np.uint64(31)
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return torch.from_numpy((u * np.float32(2) - np.float32(1)) * np.float32(0.2))


def rel_err(a, ref):
    return float((a - ref).abs().max()) / max(1e-6, float(ref.abs().max()))


class Cfg(O.Flags):
    img_height, img_width, batch_size = H, W_, B


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctypes
    from unsupervised_detection_amd import weights as W
    from unsupervised_detection_amd._ffi import lib
    from unsupervised_detection_amd.engine import Engine, EngineConfig
    torch.set_num_threads(min(32, torch.get_num_threads()))
    eng = Engine(EngineConfig(batch_size=B, in_height=IN_H, in_width=IN_W, img_height=H, img_width=W_))
    pp, pg, pr = _perturbed(O.pwc_param_specs(), 41), _perturbed(O.generator_param_specs(), 42), _perturbed(O.recover_param_specs(), 43)
    flat = {"pwc": W.from_dict(pp, W.NET_PWC).cuda(), "gen": W.from_dict(pg, W.NET_GEN).cuda(), "rec": W.from_dict(pr, W.NET_REC).cuda()}
    eng.pack_pwc(flat["pwc"])
    eng.pack_trainable(flat["gen"], flat["rec"])
    g_gen, g_rec = torch.zeros_like(flat["gen"]), torch.zeros_like(flat["rec"])
    tuned = eng.autotune(flat["gen"], flat["rec"], g_gen, g_rec)
    lib.udet_tune_rejected.restype = ctypes.c_int
    img1, img2 = _smooth_pair(B, IN_H, IN_W, 44)
    # ---- HIP forward -------------------------------------------------------------------------------------------------
    eng.forward(img1.cuda(), img2.cuda(), 3)
    torch.cuda.synchronize()
    got = {k: eng.buffer(k).cpu().clone() for k in ("flow_full", "image", "flow", "mask", "pred")}
    got["losses"] = eng.losses()
    # ---- oracle: PWC in fp32, the trainable part in fp64 on the GPU's own flow (so that PWC rounding does not leak into
    # the comparison of what follows; the flow itself is compared separately) --------------------------------------------
    image, flow, flow_full = O.prepare_inputs(pp, img1, img2, Cfg)
    pg64 = {k: v.double().requires_grad_(True) for k, v in pg.items()}
    pr64 = {k: v.double().requires_grad_(True) for k, v in pr.items()}
    out = O.forward_from_flow(pg64, pr64, got["image"].double(), got["flow"].double(), Cfg)
    grads = {"gen": O.grads_of(out["generator"], pg64), "rec": O.grads_of(out["recover"], pr64)}
    return dict(eng=eng, W=W, lib=lib, tuned=tuned, flat=flat, pp=pp, pg=pg, pr=pr, img1=img1, img2=img2, got=got,
                ref=dict(image=image, flow=flow, flow_full=flow_full, out=out, grads=grads))


def test_autotuner_ran_and_rejected_nothing(env):
    assert env["tuned"] > 150  # distinct convolution / filter-gradient problems of the config-2 plan
    assert env["lib"].udet_tune_rejected() == 0  # every cached winner reproduced the built-in configuration's output


def test_pwc_flow_and_resizes(env):
    got, ref = env["got"], env["ref"]
    assert rel_err(got["flow_full"], ref["flow_full"]) < 1e-3
    assert torch.equal(got["image"], ref["image"])  # legacy bilinear resize: bit-exact
    assert rel_err(got["flow"], ref["flow"]) < 1e-3


def test_mask_predictions_losses(env):
    got, out = env["got"], env["ref"]["out"]
    assert float((got["mask"] - out["mask"].float()).abs().max()) < 1e-3
    ref = torch.cat([out["pred"], out["pred_c"], out["pred_img"]], 0).float()
    assert rel_err(got["pred"], ref) < 1e-3
    for k, v in got["losses"].items():
        assert abs(v - float(out[k])) < 1e-3 * max(1.0, abs(float(out[k]))), (k, v, float(out[k]))


def _check_grads(W, net, got_flat, ref):
    d = W.as_dict(got_flat.cpu(), net)
    scale = max(float(v.abs().max()) for v in ref.values())
    for k, v in ref.items():
        err = float((d[k].double() - v).abs().max())
        tol = 1e-3 * max(float(v.abs().max()), 1e-3 * scale)
        assert err < tol, (k, err, float(v.abs().max()))


@pytest.mark.parametrize("which", [3, 1, 2], ids=["both", "generator_loss_only", "recover_loss_only"])
def test_parameter_gradients(env, which):
    """which = 1 / 2 are train_generator_op / train_recover_op's compute_gradients -- the only passes the reference runs"""
    eng, W = env["eng"], env["W"]
    sentinel = 12345.0
    g_gen = torch.full_like(env["flat"]["gen"], sentinel)
    g_rec = torch.full_like(env["flat"]["rec"], sentinel)
    eng.forward_from_flow(env["got"]["image"].cuda(), env["got"]["flow"].cuda(), 3)
    eng.backward(which, env["flat"]["gen"], env["flat"]["rec"], g_gen, g_rec)
    torch.cuda.synchronize()
    if which & 1:
        _check_grads(W, W.NET_GEN, g_gen, env["ref"]["grads"]["gen"])
    else:
        assert float(g_gen.min()) == sentinel and float(g_gen.max()) == sentinel  # untouched
    if which & 2:
        _check_grads(W, W.NET_REC, g_rec, env["ref"]["grads"]["rec"])
    else:
        assert float(g_rec.min()) == sentinel and float(g_rec.max()) == sentinel


def test_one_optimizer_apply_on_the_step_gradients(env):
    """clip +-0.2 + Adam with the shared beta powers (loss_utils.py:22-32, adversarial_learner.py:216) on this step's
    gradients: the updated weights equal the oracle's TF Adam applied to the same (HIP) gradients"""
    eng, W = env["eng"], env["W"]
    g_gen, g_rec = torch.zeros_like(env["flat"]["gen"]), torch.zeros_like(env["flat"]["rec"])
    eng.forward_from_flow(env["got"]["image"].cuda(), env["got"]["flow"].cuda(), 3)
    eng.backward(3, env["flat"]["gen"], env["flat"]["rec"], g_gen, g_rec)
    eng.adam_step = 0
    opt = O.TFAdam(beta1=0.9)
    for net, key, g in ((W.NET_REC, "rec", g_rec), (W.NET_GEN, "gen", g_gen)):
        w = env["flat"][key].clone()
        raw = g.cpu().clone()
        m, v = torch.zeros_like(w), torch.zeros_like(w)
        eng.apply(net, w, g, m, v)
        # at this size the untrained generator's gradient can be below train_op's 1e-5 threshold (mean over the variables of
        # mean|g_v|), so the escape-noise branch may be the one taken: the oracle is handed the library's counter-based
        # stream (host replica above), sliced per variable
        offs = {name: (off, int(torch.tensor(shape).prod())) for name, shape, off in W.param_table(net)}
        stream = _noise_stream(8964, eng.adam_step, raw.numel())
        cl, changed = O.clip_or_noise(W.as_dict(raw, net), 0.2, net == W.NET_GEN, lambda k, s: stream[offs[k][0]:offs[k][0] + offs[k][1]].view(*s))
        clipped = {"x": W.from_dict(cl, net)}
        if net == W.NET_GEN:
            assert changed == (float(eng.buffer("noise_flag").view(-1)[1]) != 0.0)
        assert torch.equal(g.cpu(), clipped["x"])
        params = {"x": env["flat"][key].cpu().clone()}
        opt.m, opt.v = {}, {}
        opt.apply(params, clipped)
        assert float((w.cpu() - params["x"]).abs().max()) < 1e-6
    assert eng.adam_step == 2


def test_augmented_test_graph_at_full_resolution(env):
    """build_aug_test_graph (adversarial_learner.py:525-592): batch 1, the four central crops resized back to 384x640,
    PWC flow + generator only; masks against the oracle."""
    from unsupervised_detection_amd import data as D
    from unsupervised_detection_amd.engine import Engine, EngineConfig

    class C1(O.Flags):
        img_height, img_width, batch_size = H, W_, 1
    eng = Engine(EngineConfig(batch_size=1, in_height=IN_H, in_width=IN_W, img_height=H, img_width=W_))
    eng.pack_pwc(env["flat"]["pwc"])
    eng.pack_trainable(env["flat"]["gen"], env["flat"]["rec"])
    i1, i2 = env["img1"][:1].cuda().contiguous(), env["img2"][:1].cuda().contiguous()
    crops, masks_b1 = (0.85, 0.9, 0.95, 1.0), []
    for crop in crops:
        c1, c2 = D.central_cropping(i1, crop), D.central_cropping(i2, crop)
        eng.forward(c1, c2, 0)
        torch.cuda.synchronize()
        image, flow, _ = O.prepare_inputs(env["pp"], c1.cpu(), c2.cpu(), C1)
        gflow = eng.buffer("flow").cpu().clone()
        assert rel_err(gflow, flow) < 1e-3
        m = O.generator_net(env["pg"], image, O.preprocess_flow_batch(gflow))
        assert float((eng.buffer("mask").cpu() - m).abs().max()) < 1e-3, crop
        masks_b1.append(eng.buffer("mask").cpu().clone())
    # the learner's form of the same graph: the four crops as the batch of ONE plan (every op of the path is per sample)
    e4 = env["eng"]
    c1 = torch.cat([D.central_cropping(i1, c) for c in crops], 0)
    c2 = torch.cat([D.central_cropping(i2, c) for c in crops], 0)
    e4.forward(c1, c2, 0)
    torch.cuda.synchronize()
    assert float((e4.buffer("mask").cpu() - torch.cat(masks_b1, 0)).abs().max()) < 1e-4


def test_stream_wait_grads_hands_over_the_recover_gradients_early(env):
    """udet_stream_wait_grads (trainer._exchange_gradients): with which = 3 a communication stream that waits only for
    the RECOVER gradients' completion event may read g_rec while the longer generator-loss pass is still running.  One process,
    no collective: a side stream waits on the event and clones g_rec;
      (a) the clone equals g_rec after a full synchronisation (the event really marks the final buffer),
      (b) the side stream's work is done no later than the whole backward is (timestamps on the two streams): the overlap window the
          RCCL exchange uses is whatever the longer pass leaves,
      (c) waiting for the generator event instead yields the final g_gen."""
    eng, W, lib, flat = env["eng"], env["W"], env["lib"], env["flat"]
    from unsupervised_detection_amd._ffi import check
    eng.forward(env["img1"].cuda(), env["img2"].cuda(), 3)
    g_gen, g_rec = torch.zeros_like(flat["gen"]), torch.zeros_like(flat["rec"])
    side_r, side_g = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(2):  # (the first round also warms the allocator of the side streams)
        g_gen.zero_()
        g_rec.zero_()
        torch.cuda.synchronize()
        t0, t_main, t_side = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        t0.record()
        eng.backward(3, flat["gen"], flat["rec"], g_gen, g_rec)
        t_main.record()  # the caller's stream has joined BOTH passes here
        check(lib.udet_stream_wait_grads(eng._h, W.NET_REC, side_r.cuda_stream))
        with torch.cuda.stream(side_r):
            early = g_rec.clone()
            t_side.record(side_r)
        check(lib.udet_stream_wait_grads(eng._h, W.NET_GEN, side_g.cuda_stream))
        with torch.cuda.stream(side_g):
            gen_copy = g_gen.clone()
        torch.cuda.synchronize()
    assert float(g_rec.abs().max()) > 0 and torch.equal(early, g_rec)
    assert torch.equal(gen_copy, g_gen)
    ms_side, ms_main = t0.elapsed_time(t_side), t0.elapsed_time(t_main)
    print("recover gradients final (and cloned) after %.2f ms, whole backward after %.2f ms" % (ms_side, ms_main))
    # (round 4: with the generator's backward-data on the Winograd kernels the two passes end within ~0.1 ms of each other at this shape
    # -- the window is what is left of the longer pass, possibly nothing; what must hold is that the hand-over never comes LATER than
    # the joined backward)
    assert ms_side <= ms_main + 0.05, (ms_side, ms_main)
    _check_grads(W, W.NET_REC, early, env["ref"]["grads"]["rec"])
    # an unknown net / no backward yet is an argument error, not a hang
    with pytest.raises(ValueError):
        check(lib.udet_stream_wait_grads(eng._h, 0, side_r.cuda_stream))
