"""GPU parity of the single-op C-ABI entry points against the CPU oracle (same seeded inputs).
Tolerance for floating-point results: 1e-3 (BASELINE.json north_star), tightened where the
arithmetic allows; the warp grid-index math is checked bit-exactly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle_torch as O  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from unsupervised_detection_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("n,h,w,c,scale", [(4, 12, 20, 128, 0.625), (4, 24, 40, 96, 1.25), (2, 48, 80, 64, 2.5),
                                           (1, 96, 160, 32, 5.0), (1, 7, 9, 4, 1.0)])
def test_warp_bit_exact(ops, n, h, w, c, scale):
    img = rnd(n, h, w, c, seed=1)
    flow = rnd(n, h, w, 2, seed=2, scale=3.0)
    out, idx, alpha = ops.dense_image_warp(img.cuda(), flow.cuda(), scale, debug=True)
    sf = flow * np.float32(scale)
    fy, fx, ay, ax = O.warp_indices(sf)
    assert torch.equal(idx[..., 0].cpu(), fy) and torch.equal(idx[..., 1].cpu(), fx)   # bit-exact indices
    assert torch.equal(alpha[..., 0].cpu(), ay) and torch.equal(alpha[..., 1].cpu(), ax)  # bit-exact alphas
    ref = O.dense_image_warp(img, sf)
    assert torch.equal(out.cpu(), ref)  # same float32 op order, no FMA contraction -> bit-exact output


def test_warp_border_saturation_and_identity(ops):
    img = rnd(1, 8, 8, 4, seed=3)
    z = torch.zeros(1, 8, 8, 2)
    assert torch.allclose(ops.dense_image_warp(img.cuda(), z.cuda()).cpu(), img, atol=1e-6)
    far = torch.full((1, 8, 8, 2), 100.0)
    out = ops.dense_image_warp(img.cuda(), far.cuda()).cpu()
    assert torch.allclose(out, img[:, :1, :1].expand_as(img), atol=1e-6)


@pytest.mark.parametrize("n,h,w,c", [(4, 6, 10, 196), (4, 12, 20, 128), (2, 24, 40, 96), (2, 48, 80, 64),
                                     (1, 96, 160, 32), (1, 5, 7, 4)])
def test_cost_volume(ops, n, h, w, c):
    c1, c2 = rnd(n, h, w, c, seed=4), rnd(n, h, w, c, seed=5)
    out = ops.cost_volume(c1.cuda(), c2.cuda()).cpu()
    ref = O.cost_volume(c1, c2)
    assert out.shape == ref.shape
    assert (out - ref).abs().max() < 1e-5



@pytest.mark.parametrize("n,h,w,c,scale", [(4, 12, 20, 128, 0.625), (4, 24, 40, 96, 1.25), (2, 48, 80, 64, 2.5),
                                           (1, 96, 160, 32, 5.0), (2, 13, 21, 36, 1.0), (1, 5, 7, 4, 1.0)])
def test_fused_warp_cost_volume_is_bit_identical_to_the_two_kernels(ops, n, h, w, c, scale):
    """model_pwcnet.py:616-623 in one launch: the warped tensor (kept in LDS) and the correlation must equal udet_warp ->
    udet_cost_volume bit for bit (same rounded grid-index math, same FMA order), and the oracle within 1e-5."""
    c1, c2 = rnd(n, h, w, c, seed=31), rnd(n, h, w, c, seed=32)
    flow = rnd(n, h, w, 2, seed=33, scale=2.0)
    corr, warped = ops.warp_cost_volume(c1.cuda(), c2.cuda(), flow.cuda(), scale, return_warped=True)
    w_ref = ops.dense_image_warp(c2.cuda(), flow.cuda(), scale)
    assert torch.equal(warped, w_ref)
    assert torch.equal(warped.cpu(), O.dense_image_warp(c2, flow * np.float32(scale)))
    assert torch.equal(corr, ops.cost_volume(c1.cuda(), w_ref))
    ref = O.cost_volume(c1, O.dense_image_warp(c2, flow * np.float32(scale)))
    assert (corr.cpu() - ref).abs().max() < 1e-5


@pytest.mark.parametrize("n,h,w,c", [(4, 6, 10, 196), (1, 9, 11, 8)])
def test_fused_cost_volume_without_warp(ops, n, h, w, c):
    """level 6: cost_volume(c1, c2) (model_pwcnet.py:619-620) through the fused kernel with flow = None"""
    c1, c2 = rnd(n, h, w, c, seed=34), rnd(n, h, w, c, seed=35)
    corr = ops.warp_cost_volume(c1.cuda(), c2.cuda(), None)
    assert torch.equal(corr, ops.cost_volume(c1.cuda(), c2.cuda()))
    assert (corr.cpu() - O.cost_volume(c1, c2)).abs().max() < 1e-5


CONV_CASES = [
    # n,h,w,cin,cout,k,s,d,act,alpha,up
    (2, 24, 40, 64, 128, 3, 1, 1, "leaky", 0.1, False),
    (2, 24, 40, 120, 96, 3, 1, 1, "leaky", 0.1, False),     # Kc % 16 != 0 -> BK=8 path, BN=96
    (1, 32, 48, 3, 16, 7, 2, 1, "leaky", 0.2, False),        # recover aconv1 (asymmetric SAME 2/3)
    (1, 32, 48, 16, 32, 5, 2, 1, "leaky", 0.2, False),       # 5x5 s2 (1/2)
    (1, 31, 47, 32, 64, 3, 2, 1, "leaky", 0.1, False),       # odd sizes, 3x3 s2
    (2, 12, 24, 128, 128, 3, 1, 16, "elu", 0.0, False),      # dilation 16 (most taps culled)
    (2, 24, 48, 128, 128, 3, 1, 4, "elu", 0.0, False),
    (1, 12, 24, 256, 128, 4, 1, 1, "leaky", 0.2, False),     # 4x4 s1 (pad 1/2)
    (1, 24, 48, 5, 32, 5, 1, 1, "elu", 0.0, False),          # generator conv1
    (1, 24, 48, 128, 64, 3, 1, 1, "elu", 0.0, True),         # gen_deconv: fused NN x2
    (1, 24, 40, 565, 2, 3, 1, 1, "none", 0.0, False),        # 2-channel flow head, ragged Cin
    (4, 6, 10, 529, 128, 3, 1, 1, "leaky", 0.1, False),      # tiny level-6 grid -> split-K
    (1, 48, 96, 50, 2, 5, 1, 1, "none", 0.0, False),         # recover flow1 5x5
    (4, 6, 10, 196, 196, 3, 1, 1, "leaky", 0.1, False),      # Cout=196 (two N tiles)
    (1, 24, 48, 104, 16, 4, 1, 1, "none", 0.0, False),       # recover deconv1: operand-swapped filter gradient (16 channels)
    (1, 24, 48, 32, 16, 3, 1, 1, "none", 0.0, False),        # generator conv16 shape, swapped view
    (1, 24, 48, 64, 8, 3, 1, 1, "none", 0.0, False),         # 8 output channels
]


def _oracle_conv(x, w, b, s, d, act, alpha, up):
    if up:
        x = O.resize_nearest_align_corners(x, 2 * x.shape[1], 2 * x.shape[2])
    y = O.conv2d_same(x, w, b, s, d)
    if act == "leaky":
        y = O.leaky_relu(y, alpha)
    elif act == "elu":
        y = torch.nn.functional.elu(y)
    return y


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_forward(ops, case):
    n, h, w, cin, cout, k, s, d, act, alpha, up = case
    x = rnd(n, h, w, cin, seed=6)
    wt = rnd(k, k, cin, cout, seed=7, scale=(2.0 / (k * k * cin)) ** 0.5)
    b = rnd(cout, seed=8, scale=0.1)
    y = ops.conv2d(x.cuda(), wt.cuda(), b.cuda(), s, d, act, alpha, up).cpu()
    ref = _oracle_conv(x.double(), wt.double(), b.double(), s, d, act, alpha, up).float()
    assert y.shape == ref.shape
    assert (y - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_backward(ops, case):
    n, h, w, cin, cout, k, s, d, act, alpha, up = case
    x = rnd(n, h, w, cin, seed=9).double().requires_grad_(True)
    wt = rnd(k, k, cin, cout, seed=10, scale=(2.0 / (k * k * cin)) ** 0.5).double().requires_grad_(True)
    b = rnd(cout, seed=11, scale=0.1).double().requires_grad_(True)
    y = _oracle_conv(x, wt, b, s, d, act, alpha, up)
    dy = rnd(*y.shape, seed=12).double()
    gx, gw, gb = torch.autograd.grad((y * dy).sum(), [x, wt, b])
    ys = y.detach().float().cuda()
    dw, db = ops.conv2d_backward_filter(x.detach().float().cuda(), dy.float().cuda(), ys, (k, k), s, d, act, alpha, up)
    tol = lambda ref: 2e-4 * max(1.0, float(ref.abs().max()))
    assert (dw.cpu() - gw.float()).abs().max() < tol(gw)
    assert (db.cpu() - gb.float()).abs().max() < tol(gb)
    if not up:
        dx = ops.conv2d_backward_data(dy.float().cuda(), ys, wt.detach().float().cuda(), (h, w), s, d, act, alpha)
        assert (dx.cpu() - gx.float()).abs().max() < tol(gx)


# Winograd-domain filter gradient (conv_wgrad_wino.hip; udet_debug_force_wgrad(slices, 3)): n, h, w, cin, cout, dilation, K slices
WGRAD_WINO_CASES = [
    (2, 24, 48, 64, 64, 1, 16),       # whole strips (W / 2 = 24 tiles = three strips of eight)
    (1, 32, 64, 128, 128, 2, 9),      # dilation 2: four sub-lattices of 16 x 32, four channel-block pairs
    (2, 13, 21, 64, 128, 1, 5),       # odd grid: half tiles on both axes, a partly filled strip
    (1, 41, 50, 64, 64, 3, 1000),     # dilation 3: nine sub-lattices of unequal size; more slices asked for than strips exist
    (1, 9, 7, 64, 64, 1, 1),          # smaller than one strip, one slice
    (3, 12, 24, 128, 64, 4, 7),       # dilation 4 on a small grid: 3 x 6 sub-lattices, mostly padding tiles
]


@pytest.mark.parametrize("case", WGRAD_WINO_CASES)
def test_filter_gradient_winograd_family(ops, case):
    """dW and db of a 3x3 stride-1 convolution through the Winograd-domain family against float64 autograd, and against the direct
    family on the same data; the family under test really ran."""
    from unsupervised_detection_amd._devel import dbg
    n, h, w, cin, cout, d, ns = case
    x = rnd(n, h, w, cin, seed=91).double()
    wt = rnd(3, 3, cin, cout, seed=92, scale=(2.0 / (9 * cin)) ** 0.5).double().requires_grad_(True)
    b = rnd(cout, seed=93, scale=0.1).double().requires_grad_(True)
    y = O.conv2d_same(x, wt, b, 1, d)
    dy = rnd(*y.shape, seed=94).double()
    gw, gb = torch.autograd.grad((y * dy).sum(), [wt, b])
    xg, dyg = x.float().cuda(), dy.float().cuda()
    try:
        dbg.udet_debug_force_wgrad(ns, 3)
        dw, db = ops.conv2d_backward_filter(xg, dyg, None, (3, 3), 1, d, "none", 0.0, False)
        last = dbg.udet_debug_last_wgrad()
        dbg.udet_debug_force_wgrad(max(1, min(ns, 8)), 1)
        dw1, db1 = ops.conv2d_backward_filter(xg, dyg, None, (3, 3), 1, d, "none", 0.0, False)
        last1 = dbg.udet_debug_last_wgrad()
    finally:
        dbg.udet_debug_force_wgrad(0, -1)
    eligible = not (d == 4 and h == 12)  # (padding tiles beyond three times the pixels: the family declines, the direct form runs)
    assert ((last >> 20) == 3) == eligible and (last1 >> 20) != 3
    tol = lambda ref: 2e-4 * max(1.0, float(ref.abs().max()))
    assert (dw.cpu() - gw.float()).abs().max() < tol(gw)
    assert (db.cpu() - gb.float()).abs().max() < tol(gb)
    assert (dw.cpu() - dw1.cpu()).abs().max() < 1e-4 * max(1.0, float(gw.abs().max()))


def test_filter_gradient_winograd_family_at_the_generator_shape(ops):
    """The family on the generator's 128 -> 128 layers' own problem (4 x 48 x 96: models/nets.py:23-31), 64 K slices x 4 channel-block pairs =
    one workgroup per CU; reference: the float32 PyTorch-CPU weight gradient (5.4 GFLOP; float64 would take minutes)."""
    from unsupervised_detection_amd._devel import dbg
    n, h, w, c = 4, 48, 96, 128
    x = rnd(n, h, w, c, seed=95)
    dy = rnd(n, h, w, c, seed=96)
    gw = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).contiguous(), (c, c, 3, 3), dy.permute(0, 3, 1, 2).contiguous(), padding=1).permute(2, 3, 1, 0)
    gb = dy.sum((0, 1, 2))
    try:
        dbg.udet_debug_force_wgrad(64, 3)
        dw, db = ops.conv2d_backward_filter(x.cuda(), dy.cuda(), None, (3, 3), 1, 1, "none", 0.0, False)
        assert (dbg.udet_debug_last_wgrad() >> 20) == 3 and (dbg.udet_debug_last_wgrad() & 0xfffff) == 64
    finally:
        dbg.udet_debug_force_wgrad(0, -1)
    assert (dw.cpu() - gw).abs().max() < 2e-4 * max(1.0, float(gw.abs().max()))
    assert (db.cpu() - gb).abs().max() < 2e-4 * max(1.0, float(gb.abs().max()))


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 6, 10, 529, 2), (1, 12, 20, 2, 2), (1, 24, 40, 64, 32)])
def test_conv2d_transpose(ops, n, h, w, cin, cout):
    x = rnd(n, h, w, cin, seed=13)
    wt = rnd(4, 4, cout, cin, seed=14, scale=(1.0 / (16 * cin)) ** 0.5)
    b = rnd(cout, seed=15, scale=0.1)
    y = ops.conv2d_transpose4x4s2(x.cuda(), wt.cuda(), b.cuda()).cpu()
    ref = O.conv2d_transpose_k4s2_same(x.double(), wt.double(), b.double()).float()
    assert y.shape == ref.shape and (y - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max()))


# every convolution kernel family on the same problems (the autotuner picks among them per shape at run time):
# bm bit 16 = non-specialised 256-thread kernel, bit 17 = LDS-DMA staging, bit 18 = tile-resident kernel (low bits = tile height),
# bit 19 = self-staging LDS-DMA kernel (4 waves, 16-wide stages)
FAMILIES = {"plain": (128 + (1 << 16), 32, 1), "wave_spec": (128, 32, 1), "lds_dma": (128 + (1 << 17), 32, 1),
            "lds_dma_split3": (128 + (1 << 17), 32, 3), "tile8": ((1 << 18) + 8, 0, 1), "tile4": ((1 << 18) + 4, 0, 1),
            "self_staging": (128 + (1 << 19), 64, 1), "self_staging_128": (128 + (1 << 19), 128, 1),
            "self_staging_split2": (64 + (1 << 19), 64, 2), "self_staging_n32": (128 + (1 << 19), 32, 1),
            "self_staging_256x32": (256 + (1 << 19), 32, 1),
            # split-K both ways: bit 21 = slabs summed by the last-arriving workgroup (ticket counter), bit 20 = second launch
            "wave_spec_split4_fold": (128 + (1 << 21), 32, 4), "wave_spec_split4_2pass": (128 + (1 << 20), 32, 4),
            "plain_split2_fold": (128 + (1 << 16) + (1 << 21), 32, 2), "lds_dma_split3_2pass": (128 + (1 << 17) + (1 << 20), 32, 3),
            "lds_dma_split5_fold_64": (64 + (1 << 17) + (1 << 21), 64, 5), "self_staging_split2_2pass": (64 + (1 << 19) + (1 << 20), 64, 2),
            # bit 22 / 23 = LDS-DMA staging with a 3 / 4 stage ring (two / three stages in flight)
            "lds_dma_ring3": (128 + (1 << 22), 32, 1), "lds_dma_ring4_64_split2": (64 + (1 << 23), 64, 2)}
THIN_CASES = [
    # n,h,w,cin,cout,k,s
    (1, 32, 64, 16, 16, 3, 1),    # 16-wide MFMA tile kernel
    (2, 16, 64, 32, 32, 3, 1),
    (1, 32, 64, 104, 16, 4, 1),   # recover deconv1: four channel-block passes
    (1, 32, 64, 64, 64, 3, 1),    # two N tiles
    (1, 32, 48, 4, 16, 7, 2),     # stride-2 forward, 7x7 over the ld=4 input
    (1, 32, 64, 16, 32, 5, 2),    # stride 2: backward-data runs four parity classes, 16 output channels
    (1, 32, 64, 32, 64, 3, 2),
    (1, 30, 50, 16, 16, 3, 1),    # ragged: 50 = 32 + 18 columns, 30 rows; partial tiles on both axes
    (2, 13, 37, 32, 32, 5, 1),    # odd sizes, 5x5
    (1, 31, 47, 32, 64, 3, 2),    # odd grid, stride 2: backward-data falls back to one launch per parity class
    (1, 24, 64, 32, 194, 4, 1),   # the backward-data view of recover deconv2: 194 output columns = seven 32-column blocks
]


def _tile_fits(th, cin, cout, k, s):
    """LDS need of the tile-resident kernel (conv_tile_lds_bytes): halo tile of <= 32 channels + all taps' weights of one
    32- (16-) column block must fit 96 KB; otherwise a forced tile family falls back to the built-in choice."""
    cb = min((cin + 7) // 8 * 8, 32)
    nw = 16 if cout <= 16 else 32
    pix = ((th - 1) * s + k) * (31 * s + k)
    return (cb // 4 * (pix | 1) + (k * k * cb // 4 + 64 // nw) * nw) * 16 + (k * k + 2 * pix) * 4 <= 96 * 1024


@pytest.fixture
def force_conv():
    from unsupervised_detection_amd._devel import dbg as lib  # libudet_debug.so: the test-only hooks
    yield lib
    lib.udet_debug_force_conv(0, 0, -1)


WS_OF = {"plain": 0, "wave_spec": 1, "lds_dma": 2, "lds_dma_split3": 2, "tile8": 3, "tile4": 3, "self_staging": 6,
         "self_staging_128": 6, "self_staging_split2": 6, "self_staging_n32": 6, "self_staging_256x32": 6,
         "wave_spec_split4_fold": 1, "wave_spec_split4_2pass": 1, "plain_split2_fold": 0, "lds_dma_split3_2pass": 2,
         "lds_dma_split5_fold_64": 2, "self_staging_split2_2pass": 6, "lds_dma_ring3": 4, "lds_dma_ring4_64_split2": 5}


@pytest.mark.parametrize("family", list(FAMILIES))
@pytest.mark.parametrize("case", THIN_CASES)
def test_conv_kernel_families(ops, force_conv, family, case):
    n, h, w, cin, cout, k, s = case
    x = rnd(n, h, w, cin, seed=21).double().requires_grad_(True)
    wt = rnd(k, k, cin, cout, seed=22, scale=(2.0 / (k * k * cin)) ** 0.5).double()
    b = rnd(cout, seed=23, scale=0.1).double()
    y = _oracle_conv(x, wt, b, s, 1, "leaky", 0.1, False)
    lin = O.conv2d_same(x, wt, None, s, 1)
    dy = rnd(*y.shape, seed=24).double()
    gx, = torch.autograd.grad((lin * dy).sum(), [x])
    force_conv.udet_debug_force_conv(*FAMILIES[family])
    got = ops.conv2d(x.detach().float().cuda(), wt.float().cuda(), b.float().cuda(), s, 1, "leaky", 0.1, False).cpu()
    if WS_OF[family] != 3 or _tile_fits(8 if family == "tile8" else 4, cin, cout, k, s):
        assert (force_conv.udet_debug_last_conv() & 0xff) == WS_OF[family]  # the family under test really ran
    if "_fold" in family or "_2pass" in family:  # ... with the split count and the summation mode asked for
        last = force_conv.udet_debug_last_conv()
        assert (last >> 20) & 0xff > 1 and (last >> 28) & 1 == (1 if "_fold" in family else 0)
    assert (got - y.detach().float()).abs().max() < 1e-4 * max(1.0, float(y.abs().max()))
    # backward-data of the linear layer (no act' on load: the form the step uses, dU being materialised by its producer)
    dx = ops.conv2d_backward_data(dy.float().cuda(), lin.detach().float().cuda(), wt.float().cuda(), (h, w), s, 1, "none", 0.0).cpu()
    # backward-data: K = cout (at most 256 channels in the tile-resident kernel), N = cin (at most 256 columns), stride-1 walk over the dY grid
    if WS_OF[family] != 3 or (cin <= 256 and cout <= 256 and _tile_fits(8 if family == "tile8" else 4, cout, cin, k, 1)):
        assert (force_conv.udet_debug_last_conv() & 0xff) == WS_OF[family]
    assert (dx - gx.float()).abs().max() < 2e-4 * max(1.0, float(gx.abs().max()))


# Winograd F(2x2,3x3) family (conv_wino.hip; bit 25 of the forced tile, low bits = variant: bit 0 64 tiles x 64 channels / 128 x 32,
# bit 1 the four-wave / the eight-wave kernel; 4: the half-size form -- 32 tiles x 64 channels, two-buffer ring, two workgroups per CU):
# n, h, w, cin, cout, dilation
WINO_CASES = [
    (1, 32, 64, 64, 64, 1),     # whole blocks
    (2, 37, 53, 24, 40, 1),     # odd grid (partial tiles on both axes), ragged channel counts
    (1, 41, 50, 40, 70, 3),     # dilation 3: nine sub-lattices of unequal size
    (1, 24, 40, 128, 128, 2),   # generator atrous shape, two N blocks
    (2, 16, 16, 568, 32, 1),    # deep ragged slab window (PWC estimator), 32 output channels
    (1, 9, 7, 8, 96, 1),        # smaller than one block, one K stage
]


@pytest.mark.parametrize("variant,ks", [(0, 1), (1, 1), (0, 3), (1, 2), (2, 1), (3, 1), (2, 2), (3, 3), (4, 1), (4, 2)])
@pytest.mark.parametrize("case", WINO_CASES)
def test_conv_winograd_family(ops, force_conv, variant, ks, case):
    """3x3 stride-1 convolutions and their backward-data pass through the fused Winograd kernel (forward: pack mode 7 layout built
    from the packed weights; backward-data: the mirrored / transposed tap set), incl. K slices through the split-K slabs."""
    n, h, w, cin, cout, d = case
    if variant == 4 and min(cin, cout) <= 32:  # (the backward-data launch's N axis is cin)
        pytest.skip("the half-size form (32 tiles x 64 channels, two workgroups per CU) takes layers wider than 32 channels")
    x = rnd(n, h, w, cin, seed=61).double().requires_grad_(True)
    wt = rnd(3, 3, cin, cout, seed=62, scale=(2.0 / (9 * cin)) ** 0.5).double()
    b = rnd(cout, seed=63, scale=0.1).double()
    y = _oracle_conv(x, wt, b, 1, d, "leaky", 0.1, False)
    lin = O.conv2d_same(x, wt, None, 1, d)
    dy = rnd(*y.shape, seed=64).double()
    gx, = torch.autograd.grad((lin * dy).sum(), [x])
    force_conv.udet_debug_force_conv((1 << 25) + variant, 0, ks)
    got = ops.conv2d(x.detach().float().cuda(), wt.float().cuda(), b.float().cuda(), 1, d, "leaky", 0.1, False).cpu()
    assert (force_conv.udet_debug_last_conv() & 0xff) == 9  # the family under test really ran
    assert (got - y.detach().float()).abs().max() < 1e-4 * max(1.0, float(y.abs().max()))
    dx = ops.conv2d_backward_data(dy.float().cuda(), lin.detach().float().cuda(), wt.float().cuda(), (h, w), 1, d, "none", 0.0).cpu()
    assert (force_conv.udet_debug_last_conv() & 0xff) == 9
    assert (dx - gx.float()).abs().max() < 2e-4 * max(1.0, float(gx.abs().max()))


@pytest.mark.parametrize("variant,ks", [(2, 1), (0, 1), (3, 2), (4, 1)])
def test_conv_winograd_family_at_the_largest_launch(ops, force_conv, variant, ks):
    """The Winograd family forced on the step's largest launch, pwcnet/ctxt/dc_conv21's own problem (4 x 96 x 160, 565 -> 128 channels:
    model_pwcnet.py:562): the large-grid paths -- conv_wino_ok's 32-bit byte-offset guard (a 568-channel operand of 4 x 96 x 160 pixels is
    139.6 MB), 960 tile blocks in the XCD-aware order, K slices through the split-K slabs at that size -- directly, not only through the
    whole-step tests.  Reference: the fp32 PyTorch-CPU convolution of the oracle (80 GFLOP: float64 would take minutes); the implicit-GEMM
    family on the same data must agree with both."""
    n, h, w, cin, cout = 4, 96, 160, 565, 128
    x = rnd(n, h, w, cin, seed=71)
    wt = rnd(3, 3, cin, cout, seed=72, scale=(2.0 / (9 * cin)) ** 0.5)
    b = rnd(cout, seed=73, scale=0.1)
    y = _oracle_conv(x, wt, b, 1, 1, "leaky", 0.1, False)
    dy = rnd(n, h, w, cout, seed=74)
    gx = torch.nn.grad.conv2d_input((n, cin, h, w), wt.permute(3, 2, 0, 1).contiguous(), dy.permute(0, 3, 1, 2).contiguous(),
                                    padding=1).permute(0, 2, 3, 1)  # SAME 3x3 stride 1: symmetric padding 1
    xg, wg, bg, dyg = x.cuda(), wt.cuda(), b.cuda(), dy.cuda()
    force_conv.udet_debug_force_conv((1 << 25) + variant, 0, ks)
    got = ops.conv2d(xg, wg, bg, 1, 1, "leaky", 0.1, False).cpu()
    assert (force_conv.udet_debug_last_conv() & 0xff) == 9
    assert (got - y).abs().max() < 1e-4 * max(1.0, float(y.abs().max()))
    dx = ops.conv2d_backward_data(dyg, None, wg, (h, w), 1, 1, "none", 0.0).cpu()
    assert (force_conv.udet_debug_last_conv() & 0xff) == 9
    assert (dx - gx).abs().max() < 2e-4 * max(1.0, float(gx.abs().max()))
    force_conv.udet_debug_force_conv(*FAMILIES["lds_dma"])
    ref = ops.conv2d(xg, wg, bg, 1, 1, "leaky", 0.1, False).cpu()
    assert (force_conv.udet_debug_last_conv() & 0xff) == 2
    assert (got - ref).abs().max() < 5e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("case", [(2, 13, 37, 98, 3), (1, 31, 45, 50, 5), (1, 9, 70, 386, 3), (3, 8, 8, 196, 3), (1, 40, 72, 64, 5),
                                  (2, 21, 100, 56, 5), (1, 16, 64, 36, 3)])
def test_two_channel_heads_run_the_direct_kernels(ops, force_conv, case):
    """The flow / up_feat heads (2 output channels over a deep input) and their backward-data pass (2 input channels, wide output):
    the direct kernels of conv_thin.hip are what an untuned launch runs (families 8 / 7), they agree with the oracle, and they
    agree with the implicit-GEMM kernel on the same problem to summation-order rounding."""
    n, h, w, cin, k = case
    x = rnd(n, h, w, cin, seed=51).double().requires_grad_(True)
    wt = rnd(k, k, cin, 2, seed=52, scale=(2.0 / (k * k * cin)) ** 0.5).double()
    b = rnd(2, seed=53, scale=0.1).double()
    y = O.conv2d_same(x, wt, b, 1, 1)
    dy = rnd(*y.shape, seed=54).double()
    gx, = torch.autograd.grad((y * dy).sum(), [x])
    xf, wf, bf, dyf = x.detach().float().cuda(), wt.float().cuda(), b.float().cuda(), dy.float().cuda()
    got = ops.conv2d(xf, wf, bf, 1, 1, "none", 0.0, False).cpu()
    assert (force_conv.udet_debug_last_conv() & 0xff) == 8
    assert (got - y.detach().float()).abs().max() < 1e-4 * max(1.0, float(y.abs().max()))
    dx = ops.conv2d_backward_data(dyf, None, wf, (h, w), 1, 1, "none", 0.0).cpu()
    assert (force_conv.udet_debug_last_conv() & 0xff) == 7
    assert (dx - gx.float()).abs().max() < 2e-4 * max(1.0, float(gx.abs().max()))
    force_conv.udet_debug_force_conv(*FAMILIES["wave_spec"])
    ref_y = ops.conv2d(xf, wf, bf, 1, 1, "none", 0.0, False).cpu()
    assert (force_conv.udet_debug_last_conv() & 0xff) == 1
    ref_dx = ops.conv2d_backward_data(dyf, None, wf, (h, w), 1, 1, "none", 0.0).cpu()
    assert (got - ref_y).abs().max() < 2e-5 * max(1.0, float(ref_y.abs().max()))
    assert (dx - ref_dx).abs().max() < 2e-5 * max(1.0, float(ref_dx.abs().max()))


@pytest.mark.parametrize("grid", [(2, 12, 20, 529), (1, 10, 50, 36)])
def test_transposed_two_channel_head_runs_the_direct_kernel(ops, force_conv, grid):
    """up_feat_l (models/PWCNet/model_pwcnet.py:283-286): conv2d_transpose 4x4 s2 with two output channels -- the four output
    parity classes share the staged input tile of the direct kernel."""
    n, h, w, c = grid
    x = rnd(n, h, w, c, seed=55)
    wt = rnd(4, 4, 2, c, seed=56, scale=(1.0 / (16 * c)) ** 0.5)
    b = rnd(2, seed=57, scale=0.1)
    y = ops.conv2d_transpose4x4s2(x.cuda(), wt.cuda(), b.cuda()).cpu()
    assert (force_conv.udet_debug_last_conv() & 0xff) == 8
    ref = O.conv2d_transpose_k4s2_same(x.double(), wt.double(), b.double()).float()
    assert (y - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("family", list(FAMILIES))
def test_conv2d_transpose_kernel_families(ops, force_conv, family):
    x = rnd(2, 12, 20, 64, seed=25)
    wt = rnd(4, 4, 32, 64, seed=26, scale=(1.0 / (16 * 64)) ** 0.5)
    b = rnd(32, seed=27, scale=0.1)
    ref = O.conv2d_transpose_k4s2_same(x.double(), wt.double(), b.double()).float()
    force_conv.udet_debug_force_conv(*FAMILIES[family])
    y = ops.conv2d_transpose4x4s2(x.cuda(), wt.cuda(), b.cuda()).cpu()
    assert (force_conv.udet_debug_last_conv() & 0xff) == WS_OF[family]
    assert (y - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("family", ["lds_dma", "lds_dma_ring3"])
@pytest.mark.parametrize("stride", [1, 2])
def test_conv_tail_split(ops, force_conv, family, stride):
    """Tail split (ks + 256 r in the forced split count): 288 tiles of 64x64 = one whole round of 256 + 32 tiles cut into 8 K slices
    each; with stride 2 the backward-data launch has four parity classes and its tail lies inside the last one."""
    n, h, w, cin, cout, k = 2, 96 * stride, 96, 64, 64, 3
    if stride == 2:
        w *= 2
    x = rnd(n, h, w, cin, seed=41).double().requires_grad_(True)
    wt = rnd(k, k, cin, cout, seed=42, scale=(2.0 / (k * k * cin)) ** 0.5).double()
    b = rnd(cout, seed=43, scale=0.1).double()
    y = _oracle_conv(x, wt, b, stride, 1, "leaky", 0.1, False)
    lin = O.conv2d_same(x, wt, None, stride, 1)
    dy = rnd(*y.shape, seed=44).double()
    gx, = torch.autograd.grad((lin * dy).sum(), [x])
    bm = 64 + ((1 << 17) if family == "lds_dma" else (1 << 22))
    force_conv.udet_debug_force_conv(bm, 64, 256)  # r = 1 slot per CU, slices: as many as fill the round
    got = ops.conv2d(x.detach().float().cuda(), wt.float().cuda(), b.float().cuda(), stride, 1, "leaky", 0.1, False).cpu()
    last = force_conv.udet_debug_last_conv()
    assert (last & 0xff) == WS_OF[family] and (last >> 29) & 1 == 1 and (last >> 20) & 0xff >= 2, hex(last)  # up to 256 // 32 slices
    assert (got - y.detach().float()).abs().max() < 1e-4 * max(1.0, float(y.abs().max()))
    dx = ops.conv2d_backward_data(dy.float().cuda(), lin.detach().float().cuda(), wt.float().cuda(), (h, w), stride, 1, "none", 0.0).cpu()
    last = force_conv.udet_debug_last_conv()
    assert (last & 0xff) == WS_OF[family] and (stride == 2 or (last >> 29) & 1 == 1), hex(last)  # (a 1-tap parity class cannot be split)
    assert (dx - gx.float()).abs().max() < 2e-4 * max(1.0, float(gx.abs().max()))


PAIR_CASES = [
    # na, nb, h, w, cin, cout, k, stride, dilation  -- the recover encoders' own couples (nets.py:57-75: image encoder on the B images, flow
    # encoder on the 3B samples of the batched calls) and ragged / split-K / one-tile variants
    (4, 12, 24, 48, 64, 64, 3, 1, 1),     # aconv31 + bconv31
    (4, 12, 24, 48, 64, 128, 3, 2, 1),    # aconv4 + bconv4 (stride 2, SAME padding (0, 1))
    (4, 12, 6, 12, 128, 128, 3, 1, 1),    # aconv51 + bconv51: 288 + 864 pixels -- K slices through the second pass
    (1, 3, 13, 9, 16, 24, 5, 2, 1),       # odd grid, 5x5, ragged N tile, more taps than channels per stage
    (2, 2, 17, 31, 8, 40, 3, 1, 2),       # equal batches, dilation 2, 8 channels (packed taps)
    (3, 1, 8, 8, 32, 2, 3, 1, 1),         # two output channels, first problem the larger one
]


@pytest.mark.parametrize("case", PAIR_CASES)
def test_conv_pair_launch(ops, case):
    """Two convolutions of the same geometry (separate inputs / weights / biases / outputs, different batches) in ONE launch
    (launch_conv_pair, conv_igemm_dma_pair_kernel): each result against the float64 oracle, and bit-identical to the same two problems
    launched apart on the same kernel family."""
    import ctypes
    from unsupervised_detection_amd._devel import dbg
    na, nb, h, w, cin, cout, k, s, d = case
    xs = [rnd(n, h, w, cin, seed=300 + i) for i, n in enumerate((na, nb))]
    ws = [rnd(k, k, cin, cout, seed=310 + i, scale=(2.0 / (k * k * cin)) ** 0.5) for i in range(2)]
    bs = [rnd(cout, seed=320 + i, scale=0.1) for i in range(2)]
    refs = [O.leaky_relu(O.conv2d_same(x.double(), wt.double(), b.double(), s, d), 0.2).float() for x, wt, b in zip(xs, ws, bs)]
    oh, ow = refs[0].shape[1:3]
    g = [t.cuda() for t in xs + ws + bs]
    ws_bytes = (2 * k * k * cin * ((cout + 3) // 4 * 4 + 64) + (4 << 20) + 16384) * 4
    work = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def run(force):
        ys = [torch.full((n, oh, ow, cout), float("nan"), device="cuda") for n in (na, nb)]
        try:
            dbg.udet_debug_force_pair(force)
            rc = dbg.udet_debug_conv2d_pair(g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), g[4].data_ptr(), g[5].data_ptr(),
                                            ys[0].data_ptr(), ys[1].data_ptr(), na, nb, h, w, cin, cout, k, s, d, 1, ctypes.c_float(0.2),
                                            work.data_ptr(), ws_bytes, stream)
            paired = dbg.udet_debug_last_pair()
        finally:
            dbg.udet_debug_force_pair(-1)
        assert rc == 0
        torch.cuda.synchronize()
        return [y.cpu() for y in ys], paired

    got, paired = run(1)
    assert paired == 1, "the pair did not go out as one launch"
    apart, paired0 = run(0)
    assert paired0 == 0
    for y, y0, ref in zip(got, apart, refs):
        assert float((y - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
        assert float((y0 - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))


def test_elu_accuracy(ops):
    """The ELU of the convolution epilogues (csrc/common.h elu_negative: exp(v) - 1 below -0.25, a degree-6 polynomial above) against
    float64 expm1 over the whole negative range, through a 1x1 identity convolution."""
    c = 8
    vals = torch.cat([-torch.logspace(-8, 1.9, 4000, dtype=torch.float64), torch.linspace(-0.5, 0.0, 2001, dtype=torch.float64),
                      torch.linspace(0.0, 3.0, 400, dtype=torch.float64)])
    n = (vals.numel() + c - 1) // c * c
    x = torch.zeros(n, dtype=torch.float64)
    x[:vals.numel()] = vals
    x = x.view(1, n // c, 1, c).float()
    wt = torch.eye(c).view(1, 1, c, c)
    y = ops.conv2d(x.cuda(), wt.cuda(), torch.zeros(c).cuda(), 1, 1, "elu", 0.0, False).cpu().double()
    xd = x.double()
    ref = torch.where(xd > 0, xd, torch.expm1(xd))
    err = (y - ref).abs()
    assert float(err.max()) < 3e-7, float(err.max())
    small = xd.abs() < 0.25  # relative accuracy where the result is small (the polynomial branch)
    assert float((err[small] / ref[small].abs().clamp_min(1e-30)).max()) < 4e-7


def test_bad_arguments_raise(ops):
    x = torch.zeros(1, 4, 4, 6, device="cuda")
    with pytest.raises(ValueError):
        ops.dense_image_warp(x, torch.zeros(1, 4, 4, 2, device="cuda"))  # C % 4 != 0
    with pytest.raises(ValueError):
        ops.conv2d(x, torch.zeros(3, 3, 8, 4, device="cuda"))  # channel mismatch


@pytest.mark.parametrize("h,w,oh,ow,c", [(384, 640, 192, 384, 3), (96, 160, 384, 640, 2), (6, 12, 12, 24, 8), (7, 9, 11, 5, 3),
                                         (480, 854, 384, 640, 3)])
def test_resize_bilinear_legacy(ops, h, w, oh, ow, c):
    x = rnd(2, h, w, c, seed=20)
    y = ops.resize_bilinear_legacy(x.cuda(), oh, ow).cpu()
    ref = O.resize_bilinear_legacy(x, oh, ow)
    assert torch.equal(y, ref)  # same float32 op order, no contraction -> bit-exact
    xd = x.double().requires_grad_(True)
    dy = rnd(2, oh, ow, c, seed=21)
    (O.resize_bilinear_legacy(xd, oh, ow) * dy.double()).sum().backward()
    dx = ops.resize_bilinear_legacy_backward(dy.cuda(), h, w).cpu()
    assert (dx - xd.grad.float()).abs().max() < 1e-5


def test_resize_kernels_reproduce_tensorflows_unit_test_vectors(ops):
    """image_ops_test.py (TF r1.13) ResizeImagesTest.testResizeUpAlignCornersFalse: the legacy bilinear / nearest kernels."""
    from oracle.golden_inputs import TF_RESIZE_FALSE as v
    from unsupervised_detection_amd import data as D
    x = torch.tensor(v["data"], dtype=torch.float32).reshape(1, *v["in_hw"], 1)
    oh, ow = v["out_hw"]
    got = ops.resize_bilinear_legacy(x.cuda(), oh, ow).cpu().reshape(-1).tolist()
    assert got == v["bilinear"]
    got = D.crop_flip_resize(x.cuda().contiguous(), oh, ow, None, True).cpu().reshape(-1).tolist()
    assert got == v["nearest"]


def test_same_padded_convolution_reproduces_tensorflows_unit_test_vectors(ops):
    """conv_ops_test.py (TF r1.13) Conv2DTest, the stride-2 'SAME' cases, through udet_conv2d."""
    from oracle.golden_inputs import TF_CONV_SAME
    for tin, fin, stride, expected in TF_CONV_SAME:
        if stride > 2:
            continue  # the path only has stride 1 / 2 layers
        x = torch.arange(1, int(np.prod(tin)) + 1, dtype=torch.float32).reshape(tin)
        w = torch.arange(1, int(np.prod(fin)) + 1, dtype=torch.float32).reshape(fin)
        y = ops.conv2d(x.cuda(), w.cuda(), torch.zeros(fin[3]).cuda(), stride, 1, "none", 0.0, False).cpu()
        assert y.reshape(-1).tolist() == expected  # small integers: exact in fp32
