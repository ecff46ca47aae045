"""Post-processing stage.  CPU: the oracle's restatement of Pillow's 8-bit bilinear resampler against the REAL
Pillow (bit for bit), scipy's bytescale / gaussian_filter semantics, cv2.remap's fixed-point sampling on hand-computed cases, the
sanity / rectify / selection logic of the reference scripts.  GPU (-m gpu): every libudet.so kernel of the stage against that
oracle -- bit-exact for the byte / integer / float64 work, 1e-5 for the float32 CRF marginals."""
import numpy as np
import pytest

from oracle import oracle_post as P


def _mask(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    cy, cx = rng.uniform(0.3, 0.7) * h, rng.uniform(0.3, 0.7) * w
    m = np.exp(-(((yy - cy) / (0.2 * h)) ** 2 + ((xx - cx) / (0.15 * w)) ** 2)) + 0.05 * rng.random((h, w))
    return (m / m.max()).astype(np.float32)


# ------------------------------------------------------------------------------------------------------------- CPU ----
@pytest.mark.parametrize("h,w,oh,ow", [(192, 384, 172, 345), (172, 345, 192, 384), (192, 384, 203, 406), (181, 363, 192, 384),
                                       (10, 7, 13, 5), (5, 9, 3, 20)])
def test_resampler_restatement_equals_pillow(h, w, oh, ow):
    from PIL import Image
    a = np.random.default_rng(h * w + oh).integers(0, 256, (h, w), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(a).resize((ow, oh), resample=Image.BILINEAR))
    assert np.array_equal(P.pil_bilinear_u8(a, oh, ow), ref)


def test_bytescale_and_imresize():
    x = np.array([[0.0, 0.5], [1.0, 0.25]])
    assert P.bytescale(x).tolist() == [[0, 128], [255, 64]]  # (x - min) * 255 / (max - min) + 0.5, truncated
    assert P.bytescale(np.full((2, 2), 3.0)).tolist() == [[0, 0], [0, 0]]  # constant image: cscale = 1
    m = _mask(192, 384, 0)
    assert np.array_equal(P.imresize(m, (172, 345)), P.pil_bilinear_u8(P.bytescale(m), 172, 345))


def test_sanity_and_rectify():
    s = np.zeros((8, 10), np.float32)
    s[0:2] = 1.0
    assert abs(P.sanity_check(s) - (2 * 10 + 2 * 2 * 2) / (4 * 10 + 4 * 8)) < 1e-12  # top strip + its share of the side strips
    m = _mask(192, 384, 1)
    up = P.rectify_pred_mask(m, 95 / 90.0, 192, 384)    # crop > 1: central window enlarged to the full frame
    dn = P.rectify_pred_mask(m, 85 / 90.0, 192, 384)    # crop < 1: shrunk, placed on a zero canvas
    assert up.shape == dn.shape == (192, 384) and abs(up.max() - 1.0) < 1e-5 and abs(dn.max() - 1.0) < 1e-5
    hh, ww = int(192 * 85 / 90.0), int(384 * 85 / 90.0)
    h0, w0 = int((192 - hh) / 2), int((384 - ww) / 2)
    assert dn[:h0].max() == 0 and dn[:, :w0].max() == 0 and dn[h0 + hh:].max() == 0


def test_remap_known_answers():
    src = np.arange(20, dtype=np.float32).reshape(4, 5)
    zero = np.zeros((4, 5, 2), np.float32)
    assert np.array_equal(P.remap_bilinear(src, zero + np.dstack(np.meshgrid(np.arange(5), np.arange(4))).astype(np.float32)), src)
    grid = np.dstack(np.meshgrid(np.arange(5), np.arange(4))).astype(np.float32)
    half = grid.copy()
    half[..., 0] += 0.5  # half a pixel to the right: mean of horizontal neighbours, 0 beyond the last column
    out = P.remap_bilinear(src, half)
    assert np.allclose(out[:, :4], (src[:, :4] + src[:, 1:]) / 2) and np.allclose(out[:, 4], src[:, 4] / 2)
    q = grid.copy()
    q[..., 1] += 1.0 / 64 + 1e-4  # rounds to 1/32 of a pixel
    assert np.allclose(P.remap_bilinear(src, q)[:3], src[:3] * (31 / 32) + src[1:] * (1 / 32))
    far = grid + 100.0
    assert P.remap_bilinear(src, far).max() == 0.0  # BORDER_CONSTANT


def test_gaussian_filter_matches_scipy_and_selection():
    from scipy import ndimage
    m = _mask(40, 60, 2).astype(np.float64)
    for sigma in (0.1, 1.0, 2.5):
        assert np.allclose(P.gaussian_filter(m, sigma), ndimage.gaussian_filter(m, sigma), atol=1e-12)
    gt = (m > 0.5).astype(np.float32)
    a, b, c = m.astype(np.float32), np.roll(m, 5, 1).astype(np.float32), np.roll(m, -9, 0).astype(np.float32)
    assert P.select_candidate(a, b, c, gt)[1] == 0 and P.select_candidate(b, a, c, gt)[1] == 1 and P.select_candidate(c, b, a, gt)[1] == 2


def test_dense_crf_sharpens_towards_the_image_edge():
    """mean field with a bilateral Potts term: a blurred mask over a two-colour image snaps to the colour boundary"""
    H, W = 16, 24
    img = np.zeros((H, W, 3), np.uint8)
    img[:, 12:] = 200
    soft = np.clip(np.linspace(0.2, 0.8, W)[None].repeat(H, 0) + 0.0, 0, 1)
    new_mask, iou = P.refine(soft, img, 0.1, 3.0, 5.0, 5.0, (np.arange(W)[None].repeat(H, 0) >= 12).astype(np.float32), iters=10, radius=6)
    assert iou == 1.0 and new_mask[:, :12].max() == 0 and new_mask[:, 12:].min() == 1


# ------------------------------------------------------------------------------------------------------------- GPU ----
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def PP():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from unsupervised_detection_amd import post_processing as pp
    return pp


@gpu
def test_gpu_sanity_rectify_soft_score(PP):
    ms = [_mask(192, 384, s) for s in range(16)]
    got = PP.sanity_check(np.stack(ms))
    ref = np.array([P.sanity_check(m.astype(np.float64)) for m in ms])
    assert np.allclose(got, ref, rtol=1e-12)
    for ratio in (85 / 90.0, 95 / 90.0, 100 / 90.0):
        g = PP.rectify_pred_mask(ms[0], ratio, 192, 384).cpu().numpy()
        assert np.array_equal(g, P.rectify_pred_mask(ms[0].astype(np.float64), ratio, 192, 384))  # bytes, integer resampling, one division
    edge = np.ones((192, 384), np.float32)  # touches the border everywhere: sanity >= 0.6 -> replaced by its partner / zeroed
    pb = [[ms[0], ms[1], edge, ms[3]], [ms[4], ms[5], ms[6], ms[7]]]
    pf = [[ms[8], ms[9], ms[10], edge], [ms[12], edge, ms[14], ms[15]]]
    pb[1][1] = edge  # both directions bad for (shift 2, crop 90): contributes zeros
    got = PP.soft_score(pb, pf).cpu().numpy()
    ref = P.soft_score(pb, pf)
    assert np.array_equal(got, ref) and got.min() == 0.0 and abs(got.max() - 1.0) < 1e-5


@gpu
def test_gpu_remap_and_propagation(PP):
    import torch
    rng = np.random.default_rng(5)
    H, W = 192, 384
    src = _mask(H, W, 20)
    flow = (rng.normal(0, 3.0, (H, W, 2))).astype(np.float32)
    flow[:10] += 40.0  # samples beyond the border
    m = flow.astype(np.float64).copy()
    m[..., 0] += np.arange(W)
    m[..., 1] += np.arange(H)[:, None]
    assert np.array_equal(PP.remap(src, flow).cpu().numpy(), P.remap_bilinear(src, m.astype(np.float32)))
    ra, prev = _mask(H, W, 21), _mask(H, W, 22)
    assert np.array_equal(PP.propagate_step(ra, prev, flow).cpu().numpy(), P.propagate_step(ra, prev, flow))
    # the sequence driver with a fixed flow field per call
    masks = [_mask(H, W, 30 + k) for k in range(4)]
    imgs = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(4)]
    flows = {}

    def flow_fn(a, b):
        key = (id(a), id(b))
        if key not in flows:
            flows[key] = rng.normal(0, 2.0, (H, W, 2)).astype(np.float32)
        return flows[key]
    fwd, bwd = PP.propagate(masks, imgs, flow_fn)
    ra = masks[0]
    for k in range(1, 4):
        ra = P.propagate_step(ra, masks[k - 1], flows[(id(imgs[k]), id(imgs[k - 1]))])
        assert np.array_equal(fwd[k].cpu().numpy(), ra)
    ra = masks[3]
    for k in (2, 1, 0):
        ra = P.propagate_step(ra, masks[k + 1], flows[(id(imgs[k]), id(imgs[k + 1]))])
        assert np.array_equal(bwd[k].cpu().numpy(), ra)
    # pyflow's replacement: the path's own PWC-Net on the buffer's uint8 frames
    f = PP.PWCFlow()(imgs[0], imgs[1])
    assert tuple(f.shape) == (H, W, 2) and bool(torch.isfinite(f).all())


@gpu
def test_gpu_gaussian_and_dense_crf(PP):
    m = _mask(40, 60, 7).astype(np.float64)
    for sigma in (0.1, 1.3):
        assert np.allclose(PP.gaussian_filter(m, sigma).cpu().numpy(), P.gaussian_filter(m, sigma), atol=1e-13)
    rng = np.random.default_rng(8)
    H, W = 24, 32
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    img[:, 16:] //= 4
    soft = _mask(H, W, 9)
    un = P.unary_from_mask(soft, 0.1)
    Q = PP.dense_crf(un, img, 3.0, 13.0, 5.0, iters=5, radius=6).cpu().numpy()
    ref = P.dense_crf(un, img, 3.0, 13.0, 5.0, iters=5, radius=6)
    assert np.abs(Q - ref).max() < 1e-5 and np.allclose(Q.sum(0), 1.0, atol=1e-6)
    gt = (soft > 0.5).astype(np.float32)
    nm, iou = PP.refine(soft, img, 0.1, 3.0, 13.0, 5.0, gt, iters=5, radius=6)
    nm_ref, iou_ref = P.refine(soft, img, 0.1, 3.0, 13.0, 5.0, gt, iters=5, radius=6)
    assert np.array_equal(nm, nm_ref) and abs(float(iou) - float(iou_ref)) < 1e-7
    # full frame, the reference's parameters (sxy 25, srgb 5, compat 5, 50 iterations): runs and returns a proper labelling
    big, bimg = _mask(192, 384, 10), rng.integers(0, 256, (192, 384, 3), dtype=np.uint8)
    nm, iou = PP.refine(big, bimg, 0.1, 25.0, 5.0, 5.0, (big > 0.5).astype(np.float32))
    assert nm.shape == (192, 384) and set(np.unique(nm)) <= {0.0, 1.0} and 0.0 <= iou <= 1.0
