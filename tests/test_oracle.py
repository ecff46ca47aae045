"""Pins the CPU oracle (oracle/) with the known-answer properties that follow
from the reference code and cross-checks its two
independent restatements (torch ops vs explicit numpy loops).  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import oracle_np as ONP
from oracle import oracle_torch as O

torch.manual_seed(0)
RNG = np.random.default_rng(0)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ---- (5) SAME-pad table A and resize tables D/E --------------------------------
@pytest.mark.parametrize("n,k,s,d,exp", [
    (192, 3, 2, 1, (0, 1)), (192, 5, 2, 1, (1, 2)), (192, 7, 2, 1, (2, 3)), (12, 4, 1, 1, (1, 2)),
    (48, 3, 1, 16, (16, 16)), (48, 5, 1, 1, (2, 2)), (6, 3, 2, 1, (0, 1)), (3, 3, 1, 1, (1, 1)),
])
def test_same_pad_table(n, k, s, d, exp):
    assert O.same_pad(n, k, s, d)[:2] == exp
    assert ONP.same_pad(n, k, s, d)[:2] == exp


def test_legacy_bilinear_tables():
    lo, hi, l = O._legacy_interp_table(384, 192)  # pure even-row sampling
    assert (lo == 2 * np.arange(192)).all() and (l == 0).all()
    lo, hi, l = O._legacy_interp_table(640, 384)  # 5:3 -> lerp in {0,1/3,2/3}
    frac = np.round(l * 3) / 3
    assert np.abs(l - frac).max() < 1e-4 and set(np.round(l * 3).astype(int)) == {0, 1, 2}
    lo, hi, l = O._legacy_interp_table(96, 192)  # x2 up: lerp {0,.5}, last sample clamped
    assert set(l.tolist()) == {0.0, 0.5} and hi[-1] == 95 and lo[-1] == 95


@pytest.mark.parametrize("n", [3, 6, 12, 24, 48, 96, 128, 240])
def test_nn_align_corners_x2_is_replicate(n):
    x = torch.arange(n, dtype=torch.float32).view(1, n, 1, 1).expand(1, n, 2, 1)
    y = O.resize_nearest_align_corners(x, 2 * n, 4)
    assert (y[0, :, 0, 0].numpy() == np.arange(2 * n) // 2).all()


def test_resizes_torch_vs_numpy():
    x = RNG.standard_normal((2, 12, 20, 3)).astype(np.float32)
    for oh, ow in ((6, 12), (24, 40), (7, 11), (12, 20)):
        a = O.resize_bilinear_legacy(t(x), oh, ow).numpy()
        b = ONP.resize_bilinear_legacy(x, oh, ow)
        assert np.array_equal(a, b), (oh, ow)
    a = O.resize_nearest_align_corners(t(x), 24, 40).numpy()
    assert np.array_equal(a, ONP.resize_nearest_align_corners(x, 24, 40))


# ---- conv semantics (A, B, C) ---------------------------------------------------
@pytest.mark.parametrize("k,s,d,h,w", [(3, 1, 1, 6, 10), (3, 2, 1, 8, 12), (5, 2, 1, 8, 12), (7, 2, 1, 12, 12),
                                        (4, 1, 1, 6, 12), (3, 1, 2, 9, 7), (5, 1, 1, 6, 6), (3, 2, 1, 7, 9)])
def test_conv_same_torch_vs_loops(k, s, d, h, w):
    x = RNG.standard_normal((2, h, w, 3)).astype(np.float32)
    wt = RNG.standard_normal((k, k, 3, 4)).astype(np.float32)
    b = RNG.standard_normal((4,)).astype(np.float32)
    a = O.conv2d_same(t(x), t(wt), t(b), s, d).numpy()
    ref = ONP.conv2d_same(x, wt, b, s, d)
    assert a.shape == ref.shape
    assert np.abs(a - ref).max() < 1e-4


def test_conv_transpose_torch_vs_loops_and_adjoint():
    x = RNG.standard_normal((2, 5, 6, 3)).astype(np.float32)
    wt = RNG.standard_normal((4, 4, 2, 3)).astype(np.float32)  # [kh,kw,out,in]
    b = RNG.standard_normal((2,)).astype(np.float32)
    a = O.conv2d_transpose_k4s2_same(t(x), t(wt), t(b)).numpy()
    ref = ONP.conv2d_transpose_k4s2_same(x, wt, b)
    assert a.shape == (2, 10, 12, 2) and np.abs(a - ref).max() < 1e-4
    # definition: adjoint of the SAME stride-2 conv 2h->h with the same kernel read as HWIO [kh,kw,out(in-ch),in(out-ch)]
    y = torch.randn(2, 10, 12, 2, dtype=torch.float64)
    xx = torch.randn(2, 5, 6, 3, dtype=torch.float64)
    wd = t(wt).double()
    lhs = (O.conv2d_same(y, wd, None, 2, 1) * xx).sum()
    rhs = (y * O.conv2d_transpose_k4s2_same(xx, wd, None)).sum()
    assert abs(float(lhs - rhs)) < 1e-9 * max(1.0, abs(float(lhs)))


# ---- (1) warp known answers -----------------------------------------------------
def test_warp_zero_flow_identity_and_integer_shift():
    img = RNG.standard_normal((2, 7, 9, 4)).astype(np.float32)
    z = np.zeros((2, 7, 9, 2), np.float32)
    out0 = O.dense_image_warp(t(img), t(z)).numpy()
    # exact in the interior; on the last row/col floor is clamped to size-2 and alpha=1, so
    # out = 1*(b-a)+a which equals b only to rounding (reference semantics, core_warp.py:101-115,146-148)
    assert np.array_equal(out0[:, :-1, :-1], img[:, :-1, :-1]) and np.allclose(out0, img, atol=1e-6)
    a, b = 2, -3
    fl = z.copy(); fl[..., 0] = a; fl[..., 1] = b
    out = O.dense_image_warp(t(img), t(fl)).numpy()
    yy = np.clip(np.arange(7) - a, 0, 6); xx = np.clip(np.arange(9) - b, 0, 8)
    assert np.allclose(out, img[:, yy][:, :, xx], atol=1e-6)  # channel 0 moves rows; borders saturate


def test_warp_torch_vs_loops_bit_exact():
    img = RNG.standard_normal((2, 6, 8, 5)).astype(np.float32)
    fl = (RNG.standard_normal((2, 6, 8, 2)) * 3).astype(np.float32)
    fy, fx, ay, ax = O.warp_indices(t(fl))
    gy, gx, by, bx = ONP.warp_indices(fl)
    assert np.array_equal(fy.numpy(), gy) and np.array_equal(fx.numpy(), gx)
    assert np.array_equal(ay.numpy(), by) and np.array_equal(ax.numpy(), bx)
    assert np.array_equal(O.dense_image_warp(t(img), t(fl)).numpy(), ONP.dense_image_warp(img, fl))


# ---- (2) cost volume ------------------------------------------------------------
def test_cost_volume_known_answers():
    c1 = RNG.standard_normal((1, 10, 12, 6)).astype(np.float32)
    c2 = RNG.standard_normal((1, 10, 12, 6)).astype(np.float32)
    cv = O.cost_volume(t(c1), t(c2)).numpy()
    centre = (c1 * c2).mean(-1)
    assert np.allclose(cv[..., 40], np.where(centre > 0, centre, 0.1 * centre), atol=1e-6)
    assert np.abs(cv - ONP.cost_volume(c1, c2)).max() < 1e-5
    # planted displacement: one-hot features; c2 is c1 moved by (dy,dx)=(2,-3)
    a = np.zeros((1, 10, 12, 1), np.float32); a[0, 4, 6, 0] = 1
    b = np.zeros_like(a); b[0, 6, 3, 0] = 1
    cv = O.cost_volume(t(a), t(b)).numpy()
    assert cv[0, 4, 6].argmax() == (2 + 4) * 9 + (-3 + 4)
    # border channels see zero padding
    ones = np.ones((1, 10, 12, 2), np.float32)
    cv = O.cost_volume(t(ones), t(ones)).numpy()
    assert cv[0, 0, 0, 0] == 0 and cv[0, 0, 0, 40] == 1


# ---- (3),(4),(6) generator / recover --------------------------------------------
def _small_nets(seed=3, dtype=torch.float32):
    pg = O.init_params(O.generator_param_specs(), seed, dtype)
    pr = O.init_params(O.recover_param_specs(), seed + 1, dtype)
    g = torch.Generator().manual_seed(seed)
    for p in (pg, pr):
        for k in p:
            if k.endswith(("bias", "biases", "beta")):
                p[k] = (torch.randn(p[k].shape, generator=g, dtype=torch.float64) * 0.05).to(dtype)
            if k.endswith("gamma"):
                p[k] = (1 + torch.randn(p[k].shape, generator=g, dtype=torch.float64) * 0.05).to(dtype)
    return pg, pr


def test_generator_mask_is_sigmoid_and_complement():
    pg, _ = _small_nets()
    img = torch.rand(1, 64, 64, 3) - 0.5
    fl = torch.randn(1, 64, 64, 2)
    m = O.generator_net(pg, img, fl)
    assert m.shape == (1, 64, 64, 1) and float(m.min()) >= 0 and float(m.max()) <= 1
    # BN with gamma=1, beta=0 is a pure x0.99950037 scale
    assert abs(O.BN_SCALE - 0.99950037) < 1e-8


def test_recover_third_call_depends_only_on_image():
    _, pr = _small_nets()
    img = torch.rand(1, 64, 128, 3) - 0.5
    z = torch.zeros(1, 64, 128, 2); one = torch.ones(1, 64, 128, 1)
    a = O.recover_net(pr, img, z, one)
    b = O.recover_net(pr, img, z.clone(), one.clone())
    assert a.shape == (1, 64, 128, 2) and torch.equal(a, b)


# ---- (7),(8) losses -------------------------------------------------------------
def test_charbonnier_known_answer_and_swap_invariance():
    x = torch.randn(2, 8, 8, 2)
    l = O.charbonnier_loss(x, x, torch.ones(2, 8, 8, 1))
    assert torch.allclose(l, torch.full((2,), 8 * 8 * 2 * math.sqrt(1e-6)), rtol=1e-5)


def test_losses_bounds_and_swap():
    pg, pr = _small_nets()
    class C(O.Flags):
        img_height, img_width, batch_size = 64, 64, 2
    img = torch.rand(2, 64, 64, 3) - 0.5
    fl = torch.randn(2, 64, 64, 2) * 0.1
    out = O.forward_from_flow(pg, pr, img, fl, C)
    assert float(out["generator"]) <= 2.0
    assert torch.allclose(out["mask"] + (1 - out["mask"]), torch.ones_like(out["mask"]))


# ---- (9) parameter counts -------------------------------------------------------
def test_param_counts():
    cnt = lambda s: sum(int(np.prod(sh)) for _, sh, _ in s)
    assert cnt(O.pwc_param_specs()) == 14079050
    assert cnt(O.generator_param_specs()) == 1451062
    assert cnt(O.recover_param_specs()) == 3388610
    assert 14079050 + 1451062 + 3388610 == 18918722  # printed by adversarial_learner.py:323-325,338


# ---- semantics J/K: Adam with shared beta powers, clip / noise -------------------
def test_tf_adam_shared_powers_and_clip_noise():
    opt = O.TFAdam(beta1=0.9)
    p = {"a": torch.ones(3)}
    opt.apply(p, {"a": torch.full((3,), 0.1)})
    # first step of Adam moves by ~lr regardless of gradient scale
    assert torch.allclose(p["a"], torch.full((3,), 1 - 1e-4), atol=1e-7)
    q = {"b": torch.ones(2)}
    opt.apply(q, {"b": torch.full((2,), 0.1)})  # second apply on another net: t=2 powers
    lr_t = 1e-4 * math.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    m, v = 0.1 * 0.1, 0.01 * 0.001
    assert torch.allclose(q["b"], torch.full((2,), 1 - lr_t * m / (math.sqrt(v) + 1e-8)), atol=1e-8)
    g = {"w": torch.tensor([0.5, -0.5, 0.01])}
    c, ch = O.clip_or_noise(g, 0.2, False)
    assert not ch and torch.equal(c["w"], torch.tensor([0.2, -0.2, 0.01]))
    tiny = {"w": torch.full((4,), 1e-7)}
    c, ch = O.clip_or_noise(tiny, 0.2, True, lambda k, s: torch.full(s, -0.1))
    assert ch and torch.equal(c["w"], torch.full((4,), 0.1))  # abs(U) is non-negative


# ---- PWC plumbing at a tiny size -------------------------------------------------
def test_pwc_forward_shapes():
    pp = O.init_params(O.pwc_param_specs(), 5)
    i1 = torch.rand(1, 64, 128, 3) - 0.5
    i2 = torch.rand(1, 64, 128, 3) - 0.5
    flow, pyr = O.pwc_forward(pp, i1, i2)
    assert flow.shape == (1, 64, 128, 2) and [tuple(f.shape[1:3]) for f in pyr] == [(1, 2), (2, 4), (4, 8), (8, 16), (16, 32)]
    assert torch.isfinite(flow).all()


# ---------------------------------------------------------------------------------------------------------------
# TensorFlow's own unit-test vectors for the resize kernels (tensorflow/python/ops/image_ops_test.py, r1.13,
# ResizeImagesTest.testResizeUpAlignCornersFalse / testResizeUpAlignCornersTrue): expected outputs of the C++ kernels the
# oracle and the TF stand-in restate.  They pin the legacy (non-half-pixel) sampling of tf.image.resize_images and the
# align_corners=True nearest-neighbour rounding used by the generator's x2 upsampling (convolution_utils.py:4-24,70).
# ---------------------------------------------------------------------------------------------------------------
from oracle.golden_inputs import TF_RESIZE_FALSE, TF_RESIZE_TRUE  # noqa: E402


def test_resize_kernels_reproduce_tensorflows_unit_test_vectors():
    import numpy as np
    import torch
    from oracle import oracle_np as ONP
    from oracle import oracle_torch as OT
    from oracle import tf1_shim as S
    v = TF_RESIZE_FALSE
    x = np.asarray(v["data"], np.float32).reshape(1, *v["in_hw"], 1)
    oh, ow = v["out_hw"]
    want_b = np.asarray(v["bilinear"], np.float32).reshape(1, oh, ow, 1)
    want_n = np.asarray(v["nearest"], np.float32).reshape(1, oh, ow, 1)
    assert np.array_equal(OT.resize_bilinear_legacy(torch.from_numpy(x), oh, ow).numpy(), want_b)
    assert np.array_equal(ONP.resize_bilinear_legacy(x, oh, ow), want_b)
    assert np.array_equal(S.resize_bilinear(torch.from_numpy(x), [oh, ow], align_corners=False).numpy(), want_b)
    assert np.array_equal(OT.resize_nearest_legacy(torch.from_numpy(x), oh, ow).numpy(), want_n)
    assert np.array_equal(ONP.resize_nearest_legacy(x, oh, ow), want_n)
    assert np.array_equal(S.resize_nearest_neighbor(torch.from_numpy(x), [oh, ow], align_corners=False).numpy(), want_n)
    v = TF_RESIZE_TRUE
    x = np.asarray(v["data"], np.float32).reshape(1, *v["in_hw"], 1)
    oh, ow = v["out_hw"]
    want_b = np.asarray(v["bilinear"], np.float32).reshape(1, oh, ow, 1)
    want_n = np.asarray(v["nearest"], np.float32).reshape(1, oh, ow, 1)
    assert np.allclose(S.resize_bilinear(torch.from_numpy(x), [oh, ow], align_corners=True).numpy(), want_b, atol=1e-6)
    assert np.array_equal(OT.resize_nearest_align_corners(torch.from_numpy(x), oh, ow).numpy(), want_n)
    assert np.array_equal(ONP.resize_nearest_align_corners(x, oh, ow), want_n)
    assert np.array_equal(S.resize_nearest_neighbor(torch.from_numpy(x), [oh, ow], align_corners=True).numpy(), want_n)


def test_same_padded_convolution_reproduces_tensorflows_unit_test_vectors():
    """conv_ops_test.py (TF r1.13): the SAME-padding cases -- stride 2 on a 2x3 grid (pad (0,1)), kernel smaller than the
    stride, stride 3 -- against both oracle restatements and the TF stand-in."""
    import numpy as np
    import torch
    from oracle import oracle_np as ONP
    from oracle import oracle_torch as OT
    from oracle import tf1_shim as S
    from oracle.golden_inputs import TF_CONV_SAME
    for tin, fin, stride, expected in TF_CONV_SAME:
        x = np.arange(1, int(np.prod(tin)) + 1, dtype=np.float32).reshape(tin)
        w = np.arange(1, int(np.prod(fin)) + 1, dtype=np.float32).reshape(fin)
        assert OT.conv2d_same(torch.from_numpy(x), torch.from_numpy(w), None, stride).reshape(-1).tolist() == expected
        assert ONP.conv2d_same(x, w, None, stride).reshape(-1).tolist() == expected
        assert S.nn_conv2d(torch.from_numpy(x), torch.from_numpy(w), [1, stride, stride, 1], "SAME").reshape(-1).tolist() == expected


def test_transposed_convolution_reproduces_tensorflows_unit_test():
    """conv2d_transpose_test.py (TF r1.13) testConv2DTransposeSame: ones [2,6,4,3] through a 3x3 stride-2 'SAME' transposed
    convolution with a ones filter [3,3,2,3] -> [2,12,8,2]; every output is 3, +3 where one coordinate is an interior multiple
    of the stride, +9 where both are.  Pins which border the SAME alignment favours in the TF stand-in's general kernel; the
    oracle's 4x4 stride-2 special case is held to the stand-in by the PWC-Net fixtures (up_flow / up_feat layers)."""
    import torch
    from oracle import tf1_shim as S
    S.reset(lambda name, shape: torch.ones(shape) if name.endswith("kernel") else torch.zeros(shape))
    y = S.layers_conv2d_transpose(torch.ones(2, 6, 4, 3), 2, 3, 2, "same", name="t").numpy()
    assert y.shape == (2, 12, 8, 2)
    for h in range(12):
        for w in range(8):
            h_in = h % 2 == 0 and 0 < h < 11
            w_in = w % 2 == 0 and 0 < w < 7
            target = 3.0 + (9.0 if h_in and w_in else (3.0 if h_in or w_in else 0.0))
            assert (y[:, h, w, :] == target).all(), (h, w)
