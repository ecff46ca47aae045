"""PyTorch-CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).

PARITY (see oracle/__init__.py): pinned at the composition level by fixtures
generated from the reference's own Python on a TF-1.13 stand-in
(oracle/make_golden.py -> tests/golden/, tests/test_golden_reference.py); the
TF-1.13 C++ kernel semantics (SAME padding, legacy resizes, inference BN) are
restated here and in the stand-in, independently.

All tensors are NHWC at this API (like the reference); convolution weights are
HWIO.  ``dtype`` may be torch.float32 (the parity dtype) or torch.float64 (used
for gradient checks); index math is always done the way TF does it (float32
coordinates, int32 indices).

Every function cites the reference file:line it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# TF-1.13 kernel semantics
# --------------------------------------------------------------------------


def same_pad(in_size: int, k: int, s: int, d: int = 1):
    """TF 'SAME' padding (semantics A): out=ceil(in/s),
    total=max((out-1)*s+(k-1)*d+1-in,0), before=total//2, after=total-before."""
    out = -(-in_size // s)
    total = max((out - 1) * s + (k - 1) * d + 1 - in_size, 0)
    before = total // 2
    return before, total - before, out


def conv2d_same(x, w_hwio, b=None, stride=1, dilation=1):
    """tf.nn.conv2d / tf.layers.conv2d with padding='SAME' (semantics A, B).
    x: [N,H,W,Cin]; w: [KH,KW,Cin,Cout] (cross-correlation, no flip)."""
    kh, kw = int(w_hwio.shape[0]), int(w_hwio.shape[1])
    pt, pb, _ = same_pad(int(x.shape[1]), kh, stride, dilation)
    pl, pr, _ = same_pad(int(x.shape[2]), kw, stride, dilation)
    xn = x.permute(0, 3, 1, 2)
    xn = F.pad(xn, (pl, pr, pt, pb))
    w = w_hwio.permute(3, 2, 0, 1)
    y = F.conv2d(xn, w, b, stride=stride, dilation=dilation)
    return y.permute(0, 2, 3, 1)


def conv2d_transpose_k4s2_same(x, w_hwoi, b=None):
    """tf.layers.conv2d_transpose(x, C, 4, 2, 'same') (semantics C).
    w: [KH,KW,Cout,Cin].  Defined as the input-gradient of a SAME stride-2
    conv 2h->h with k=4 (pad 1/1):  y[oy] += x[iy]*K[ky] for oy = 2*iy+ky-1."""
    xn = x.permute(0, 3, 1, 2)
    w = w_hwoi.permute(3, 2, 0, 1)  # [Cin, Cout, KH, KW] as ConvTranspose2d wants
    y = F.conv_transpose2d(xn, w, b, stride=2, padding=1)
    return y.permute(0, 2, 3, 1)


def _legacy_interp_table(in_size: int, out_size: int):
    """TF-1.13 compute_interpolation_weights (align_corners=False, no
    half-pixel centres): scale=in/out (float32); src=i*scale (float32);
    lower=(int)src; upper=min(lower+1,in-1); lerp=src-lower."""
    scale = np.float32(in_size) / np.float32(out_size)
    i = np.arange(out_size, dtype=np.float32)
    src = (i * scale).astype(np.float32)
    lower = src.astype(np.int64)
    upper = np.minimum(lower + 1, in_size - 1)
    lerp = (src - lower.astype(np.float32)).astype(np.float32)
    return lower, upper, lerp


def resize_bilinear_legacy(x, out_h: int, out_w: int):
    """tf.image.resize_images(x,[oh,ow]) / tf.image.resize_bilinear with
    align_corners=False in TF 1.13 (semantics D).  Returns x untouched when the
    size already matches (resize_images shortcut)."""
    n, h, w, c = x.shape
    if h == out_h and w == out_w:
        return x
    ylo, yhi, yl = _legacy_interp_table(h, out_h)
    xlo, xhi, xl = _legacy_interp_table(w, out_w)
    ylo_t, yhi_t = torch.from_numpy(ylo), torch.from_numpy(yhi)
    xlo_t, xhi_t = torch.from_numpy(xlo), torch.from_numpy(xhi)
    yl_t = torch.from_numpy(yl).to(x.dtype).view(1, out_h, 1, 1)
    xl_t = torch.from_numpy(xl).to(x.dtype).view(1, 1, out_w, 1)
    top = x[:, ylo_t]
    bot = x[:, yhi_t]
    tl, tr = top[:, :, xlo_t], top[:, :, xhi_t]
    bl, br = bot[:, :, xlo_t], bot[:, :, xhi_t]
    t = tl + (tr - tl) * xl_t
    bm = bl + (br - bl) * xl_t
    return t + (bm - t) * yl_t


def resize_nearest_align_corners(x, out_h: int, out_w: int):
    """tf.image.resize_nearest_neighbor(align_corners=True) (semantics E):
    src=min(roundf(i*(in-1)/(out-1)), in-1)."""
    n, h, w, c = x.shape

    def idx(in_size, out_size):
        scale = np.float32(in_size - 1) / np.float32(out_size - 1) if out_size > 1 else np.float32(0)
        i = np.arange(out_size, dtype=np.float32)
        # roundf = round half away from zero (values are non-negative)
        src = np.floor(i * scale + np.float32(0.5)).astype(np.int64)
        return torch.from_numpy(np.minimum(src, in_size - 1))

    return x[:, idx(h, out_h)][:, :, idx(w, out_w)]


def resize_nearest_legacy(x, out_h: int, out_w: int):
    """tf.image.resize_images(method=NEAREST_NEIGHBOR), align_corners=False:
    src=min((int)floorf(i*in/out), in-1).  Used only for gt masks
    (models/adversarial_learner.py:92-94)."""
    n, h, w, c = x.shape

    def idx(in_size, out_size):
        scale = np.float32(in_size) / np.float32(out_size)
        i = np.arange(out_size, dtype=np.float32)
        return torch.from_numpy(np.minimum(np.floor(i * scale).astype(np.int64), in_size - 1))

    return x[:, idx(h, out_h)][:, :, idx(w, out_w)]


def leaky_relu(x, alpha):
    return torch.where(x > 0, x, x * alpha)


BN_EPS = 1e-3
BN_SCALE = 1.0 / math.sqrt(1.0 + BN_EPS)  # moving mean 0 / variance 1, never updated (semantics F)


# --------------------------------------------------------------------------
# PWC-Net pieces
# --------------------------------------------------------------------------


def warp_indices(flow):
    """Grid-index math of models/PWCNet/core_warp.py:99-115,189-194 in float32.
    Returns int32 floor_y, floor_x and float32 alpha_y, alpha_x, each [N,H,W]."""
    f = flow.detach().to(torch.float32)
    n, h, w, _ = f.shape
    gy = torch.arange(h, dtype=torch.float32).view(1, h, 1)
    gx = torch.arange(w, dtype=torch.float32).view(1, 1, w)
    qy = gy - f[..., 0]  # channel 0 moves rows (core_warp.py:189-194)
    qx = gx - f[..., 1]
    fy = torch.minimum(torch.clamp_min(torch.floor(qy), 0.0), torch.tensor(float(h - 2)))
    fx = torch.minimum(torch.clamp_min(torch.floor(qx), 0.0), torch.tensor(float(w - 2)))
    ay = torch.clamp(qy - fy, 0.0, 1.0)
    ax = torch.clamp(qx - fx, 0.0, 1.0)
    return fy.to(torch.int32), fx.to(torch.int32), ay, ax


def dense_image_warp(image, flow):
    """models/PWCNet/core_warp.py:153-202 + _interpolate_bilinear :42-150."""
    n, h, w, c = image.shape
    fy, fx, ay, ax = warp_indices(flow)
    fy, fx = fy.long(), fx.long()
    ay = ay.to(image.dtype).unsqueeze(-1)
    ax = ax.to(image.dtype).unsqueeze(-1)
    flat = image.reshape(n * h * w, c)
    boff = (torch.arange(n) * h * w).view(n, 1, 1)

    def gather(yc, xc):
        return flat[(boff + yc * w + xc).reshape(-1)].reshape(n, h, w, c)

    tl, tr = gather(fy, fx), gather(fy, fx + 1)
    bl, br = gather(fy + 1, fx), gather(fy + 1, fx + 1)
    top = ax * (tr - tl) + tl
    bot = ax * (br - bl) + bl
    return ay * (bot - top) + top


def cost_volume(c1, warp, search_range=4):
    """models/PWCNet/core_costvol.py:20-40: channel y*9+x = mean_c(c1*pad[h+y,w+x]);
    leaky 0.1."""
    n, h, w, c = c1.shape
    r = search_range
    p = F.pad(warp, (0, 0, r, r, r, r))
    outs = []
    for y in range(2 * r + 1):
        for x in range(2 * r + 1):
            outs.append((c1 * p[:, y:y + h, x:x + w, :]).mean(dim=3, keepdim=True))
    return leaky_relu(torch.cat(outs, dim=3), 0.1)


PWC_CH = [None, 16, 32, 64, 96, 128, 196]


def pwc_param_specs():
    """Variable list of ModelPWCNet.nn (models/PWCNet/model_pwcnet.py:599-649) in
    creation order.  (name, shape, init)."""
    specs = []
    cin = 3
    for l in range(1, 7):
        f = PWC_CH[l]
        for suf, ci in (("a", cin), ("aa", f), ("b", f)):
            specs.append((f"pwcnet/featpyr/conv{l}{suf}/kernel", (3, 3, ci, f), "he_normal"))
            specs.append((f"pwcnet/featpyr/conv{l}{suf}/bias", (f,), "zeros"))
        cin = f
    for l in range(6, 1, -1):
        x_ch = 81 if l == 6 else 81 + PWC_CH[l] + 2 + 2
        for i, co in enumerate((128, 128, 96, 64, 32)):
            specs.append((f"pwcnet/predict_flow/conv{l}_{i}/kernel", (3, 3, x_ch, co), "he_normal"))
            specs.append((f"pwcnet/predict_flow/conv{l}_{i}/bias", (co,), "zeros"))
            x_ch += co
        specs.append((f"pwcnet/predict_flow/flow{l}/kernel", (3, 3, x_ch, 2), "glorot_uniform"))
        specs.append((f"pwcnet/predict_flow/flow{l}/bias", (2,), "zeros"))
        ci = x_ch
        for i, co in enumerate((128, 128, 128, 96, 64, 32, 2)):
            specs.append((f"pwcnet/ctxt/dc_conv{l}{i + 1}/kernel", (3, 3, ci, co), "he_normal"))
            specs.append((f"pwcnet/ctxt/dc_conv{l}{i + 1}/bias", (co,), "zeros"))
            ci = co
        if l != 2:
            specs.append((f"pwcnet/upsample/up_flow{l}/kernel", (4, 4, 2, 2), "glorot_uniform"))
            specs.append((f"pwcnet/upsample/up_flow{l}/bias", (2,), "zeros"))
            specs.append((f"pwcnet/upsample/up_feat{l}/kernel", (4, 4, 2, x_ch), "glorot_uniform"))
            specs.append((f"pwcnet/upsample/up_feat{l}/bias", (2,), "zeros"))
    return specs


def pwc_forward(p, img1, img2):
    """ModelPWCNet.predict_from_img_pairs (model_pwcnet.py:61-76) -> nn (:599-649).
    img1,img2: [N,H,W,3] in [-0.5,0.5]; returns flow [N,H,W,2] and the pyramid."""
    pre = "pwcnet/"

    def conv(x, name, stride=1, dil=1, act=True):
        y = conv2d_same(x, p[pre + name + "/kernel"], p[pre + name + "/bias"], stride, dil)
        return leaky_relu(y, 0.1) if act else y

    def feats(x):  # extract_features :149-168
        pyr = [None]
        for l in range(1, 7):
            x = conv(x, f"featpyr/conv{l}a", stride=2)
            x = conv(x, f"featpyr/conv{l}aa")
            x = conv(x, f"featpyr/conv{l}b")
            pyr.append(x)
        return pyr

    c1 = feats(img1 + 0.5)  # adapt_x :39-56
    c2 = feats(img2 + 0.5)
    flow_pyr = []
    up_flow = up_feat = None
    for l in range(6, 1, -1):
        if l == 6:
            corr = cost_volume(c1[l], c2[l])
            x = corr
        else:
            scaler = 20.0 / 2 ** l
            wrp = dense_image_warp(c2[l], up_flow * scaler)
            corr = cost_volume(c1[l], wrp)
            x = torch.cat([corr, c1[l], up_flow, up_feat], dim=3)
        for i in range(5):  # predict_flow :476-506 (new activations are prepended)
            act = conv(x, f"predict_flow/conv{l}_{i}")
            x = torch.cat([act, x], dim=3)
        upfeat = x
        flow = conv(upfeat, f"predict_flow/flow{l}", act=False)
        y = upfeat  # refine_flow :559-576
        for i, d in enumerate((1, 2, 4, 8, 16, 1)):
            y = conv(y, f"ctxt/dc_conv{l}{i + 1}", dil=d)
        y = conv(y, f"ctxt/dc_conv{l}7", act=False)
        flow = flow + y
        flow_pyr.append(flow)
        if l != 2:
            up_flow = conv2d_transpose_k4s2_same(flow, p[pre + f"upsample/up_flow{l}/kernel"],
                                                 p[pre + f"upsample/up_flow{l}/bias"])
            up_feat = conv2d_transpose_k4s2_same(upfeat, p[pre + f"upsample/up_feat{l}/kernel"],
                                                 p[pre + f"upsample/up_feat{l}/bias"])
        else:
            n, h, w, _ = flow.shape
            flow_pred = resize_bilinear_legacy(flow, h * 4, w * 4) * 4.0
    return flow_pred, flow_pyr


# --------------------------------------------------------------------------
# generator / recover (models/nets.py, models/utils/convolution_utils.py)
# --------------------------------------------------------------------------

GEN_LAYERS = [
    # name, cin, cout, k, stride, rate, upsample_before, activation
    ("conv1", 5, 32, 5, 1, 1, False, "elu"),
    ("conv2_downsample", 32, 64, 3, 2, 1, False, "elu"),
    ("conv3", 64, 64, 3, 1, 1, False, "elu"),
    ("conv4_downsample", 64, 128, 3, 2, 1, False, "elu"),
    ("conv5", 128, 128, 3, 1, 1, False, "elu"),
    ("conv6", 128, 128, 3, 1, 1, False, "elu"),
    ("conv7_atrous", 128, 128, 3, 1, 2, False, "elu"),
    ("conv8_atrous", 128, 128, 3, 1, 4, False, "elu"),
    ("conv9_atrous", 128, 128, 3, 1, 8, False, "elu"),
    ("conv10_atrous", 128, 128, 3, 1, 16, False, "elu"),
    ("conv11", 128, 128, 3, 1, 1, False, "elu"),
    ("conv12", 128, 128, 3, 1, 1, False, "elu"),
    ("conv13_upsample", 128, 64, 3, 1, 1, True, "elu"),
    ("conv14", 64, 64, 3, 1, 1, False, "elu"),
    ("conv15_upsample", 64, 32, 3, 1, 1, True, "elu"),
    ("conv16", 32, 16, 3, 1, 1, False, "elu"),
    ("conv17", 16, 2, 3, 1, 1, False, "identity"),
]


def _gen_var_prefix(name):
    # gen_deconv nests a scope: conv13_upsample/conv13_upsample_conv (convolution_utils.py:70-74)
    return f"MaskNet/{name}/{name}_conv" if name.endswith("_upsample") else f"MaskNet/{name}"


def generator_param_specs():
    specs = []
    for name, cin, cout, k, s, r, up, act in GEN_LAYERS:
        pre = _gen_var_prefix(name)
        specs.append((pre + "/kernel", (k, k, cin, cout), "glorot_uniform"))
        specs.append((pre + "/bias", (cout,), "zeros"))
        specs.append((f"MaskNet/{name}/bn/gamma", (cout,), "ones"))
        specs.append((f"MaskNet/{name}/bn/beta", (cout,), "zeros"))
    return specs


def generator_net(p, images, flows):
    """models/nets.py:4-42.  gen_conv = conv(+bias) -> BN(inference; moving stats
    0/1) -> ELU (convolution_utils.py:26-53); gen_deconv = NN x2 + gen_conv
    (:55-75).  Returns mask [N,H,W,1] = softmax(logits/10)[...,0]."""
    x = torch.cat([images, flows], dim=3)
    outs = {}
    for name, cin, cout, k, s, r, up, act in GEN_LAYERS:
        pre = _gen_var_prefix(name)
        if up:
            n, h, w, c = x.shape
            x = resize_nearest_align_corners(x, 2 * h, 2 * w)
        x = conv2d_same(x, p[pre + "/kernel"], p[pre + "/bias"], s, r)
        x = p[f"MaskNet/{name}/bn/gamma"] * (x * BN_SCALE) + p[f"MaskNet/{name}/bn/beta"]
        if act == "elu":
            x = F.elu(x)
        # additive skips happen after the activation (nets.py:29,32,33)
        if name == "conv1":
            outs["x0"] = x
        elif name == "conv3":
            outs["x1"] = x
        elif name == "conv6":
            outs["x2"] = x
        elif name == "conv11":
            x = x + outs["x2"]
        elif name == "conv14":
            x = x + outs["x1"]
        elif name == "conv15_upsample":
            x = x + outs["x0"]
    x = x / 10.0
    sm = torch.softmax(x, dim=-1)
    return sm[..., 0:1]


REC_ENC = [  # name suffix, cin(None = input), cout, k, stride
    ("conv1", None, 16, 7, 2), ("conv2", 16, 32, 5, 2), ("conv3", 32, 64, 5, 2),
    ("conv31", 64, 64, 3, 1), ("conv4", 64, 128, 3, 2), ("conv41", 128, 128, 3, 1),
    ("conv5", 128, 128, 3, 2), ("conv51", 128, 128, 3, 1), ("conv6", 128, 128, 3, 2),
]
REC_DEC = [  # name, shape HWIO
    ("deconv5", (4, 4, 256, 128)), ("flow5", (3, 3, 384, 2)),
    ("deconv4", (4, 4, 384, 128)), ("upflow4", (4, 4, 2, 2)), ("flow4", (3, 3, 386, 2)),
    ("deconv3", (4, 4, 386, 64)), ("upflow3", (4, 4, 2, 2)), ("flow3", (3, 3, 194, 2)),
    ("deconv2", (4, 4, 194, 32)), ("upflow2", (4, 4, 2, 2)), ("flow2", (3, 3, 98, 2)),
    ("deconv1", (4, 4, 98, 16)), ("upflow1", (4, 4, 2, 2)), ("flow1", (5, 5, 50, 2)),
]


def recover_param_specs():
    specs = []
    for enc, cin0 in (("a", 3), ("b", 4)):
        for suf, cin, cout, k, s in REC_ENC:
            ci = cin0 if cin is None else cin
            specs.append((f"FlownetS/{enc}{suf}/weights", (k, k, ci, cout), "glorot_uniform"))
            specs.append((f"FlownetS/{enc}{suf}/biases", (cout,), "zeros"))
    for name, shape in REC_DEC:
        specs.append((f"FlownetS/{name}/weights", shape, "glorot_uniform"))
        specs.append((f"FlownetS/{name}/biases", (shape[3],), "zeros"))
    return specs


def recover_net(p, img1, flow_masked, mask):
    """models/nets.py:45-110.  conv = tf.nn.conv2d SAME + bias + leaky 0.2
    (convolution_utils.py:77-85); deconv = legacy-bilinear resize to the skip's
    size + conv 4x4 (:87-90)."""

    def conv(x, name, stride=1, act=True):
        y = conv2d_same(x, p[f"FlownetS/{name}/weights"], p[f"FlownetS/{name}/biases"], stride)
        return leaky_relu(y, 0.2) if act else y

    def deconv(x, like, name, act=True):
        return conv(resize_bilinear_legacy(x, like.shape[1], like.shape[2]), name, 1, act)

    ones_x = torch.ones_like(flow_masked[..., 0:1])
    fm = torch.cat([flow_masked, ones_x, 1.0 - mask], dim=3)
    enc = {}
    for e, x in (("a", img1), ("b", fm)):
        for suf, cin, cout, k, s in REC_ENC:
            x = conv(x, e + suf, s)
            enc[e + suf] = x
    conv6 = torch.cat([enc["aconv6"], enc["bconv6"]], dim=3)
    deconv5 = deconv(conv6, enc["bconv51"], "deconv5")
    concat5 = torch.cat([deconv5, enc["bconv51"], enc["aconv51"]], dim=3)
    flow5 = conv(concat5, "flow5", act=False)
    deconv4 = deconv(concat5, enc["bconv41"], "deconv4")
    upflow4 = deconv(flow5, enc["bconv41"], "upflow4", act=False)
    concat4 = torch.cat([deconv4, enc["bconv41"], enc["aconv41"], upflow4], dim=3)
    flow4 = conv(concat4, "flow4", act=False)
    deconv3 = deconv(concat4, enc["bconv31"], "deconv3")
    upflow3 = deconv(flow4, enc["bconv31"], "upflow3", act=False)
    concat3 = torch.cat([deconv3, enc["bconv31"], enc["aconv31"], upflow3], dim=3)
    flow3 = conv(concat3, "flow3", act=False)
    deconv2 = deconv(concat3, enc["bconv2"], "deconv2")
    upflow2 = deconv(flow3, enc["bconv2"], "upflow2", act=False)
    concat2 = torch.cat([deconv2, enc["bconv2"], enc["aconv2"], upflow2], dim=3)
    flow2 = conv(concat2, "flow2", act=False)
    deconv1 = deconv(concat2, enc["bconv1"], "deconv1")
    upflow1 = deconv(flow2, enc["bconv1"], "upflow1", act=False)
    concat1 = torch.cat([deconv1, enc["bconv1"], enc["aconv1"], upflow1], dim=3)
    flow1 = conv(concat1, "flow1", act=False)
    return resize_bilinear_legacy(flow1, img1.shape[1], img1.shape[2])


# --------------------------------------------------------------------------
# losses / optimizer (models/utils/loss_utils.py, flow_utils.py:5-12,
# models/adversarial_learner.py:87-204)
# --------------------------------------------------------------------------


def preprocess_flow_batch(flow):
    """models/utils/flow_utils.py:5-12: per sample, per channel zero-mean /
    unit-std over H,W; population variance, no epsilon."""
    mean = flow.mean(dim=(1, 2), keepdim=True)
    var = ((flow - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    return (flow - mean) / torch.sqrt(var)


def charbonnier_loss(gt_flows, pred_flows, masks, cbn=0.5):
    """models/utils/loss_utils.py:34-51 -> [B]."""
    lp = (gt_flows - pred_flows) ** 2 + 0.001 ** 2
    lp = torch.pow(lp, cbn) * masks
    return lp.sum(dim=(1, 2, 3))


class Flags:
    """Defaults of common_flags.py:6-21 that the hot path reads."""
    img_height = 192
    img_width = 384
    flow_normalizer = 80.0
    cbn = 0.5
    epsilon = 75.0
    beta1 = 0.9
    batch_size = 4


def forward_from_flow(pg, pr, image, flow, cfg=Flags, batch_size=None):
    """models/adversarial_learner.py:99-204 given image [B,192,384,3] and the
    normalised flow [B,192,384,2] (already / flow_normalizer).  Returns dict of
    tensors incl. the 8 `losses{}` entries (:196-204)."""
    B = image.shape[0] if batch_size is None else batch_size
    m = generator_net(pg, image, preprocess_flow_batch(flow))
    cm = 1.0 - m
    flow_masked = flow * (1.0 - m)
    flow_compl = flow * (1.0 - cm)
    pred = recover_net(pr, image, flow_masked, m)
    pred_c = recover_net(pr, image, flow_compl, cm)
    pred_img = recover_net(pr, image, torch.zeros_like(flow), torch.ones_like(m))
    rec = charbonnier_loss(flow, pred, m, cfg.cbn)
    rec_c = charbonnier_loss(flow, pred_c, cm, cfg.cbn)
    prior = charbonnier_loss(flow, pred_img, torch.ones_like(flow), cfg.cbn)
    num_pixels = float(cfg.img_width * cfg.img_height * B)
    recover_loss = (rec.sum() + rec_c.sum() + prior.sum()) / num_pixels
    den = charbonnier_loss(flow, pred_img, m, cfg.cbn) + cfg.epsilon
    den_c = charbonnier_loss(flow, pred_img, cm, cfg.cbn) + cfg.epsilon
    red = (1.0 - rec / den).mean(dim=0)
    red_c = (1.0 - rec_c / den_c).mean(dim=0)
    return {
        "mask": m, "pred": pred, "pred_c": pred_c, "pred_img": pred_img,
        "generator": red + red_c, "recover": recover_loss,
        "red_rate": red, "red_rate_compl": red_c,
        "reconstruction_loss": rec[0], "reconstruction_compl_loss": rec_c[0],
        "denominator_red_rate": den[0], "denominator_red_rate_compl": den_c[0],
        "rec": rec, "rec_c": rec_c, "den": den, "den_c": den_c,
    }


def prepare_inputs(pp, img1, img2, cfg=Flags):
    """models/adversarial_learner.py:83-97: PWC flow at input size, legacy
    resize of image and flow to img_height x img_width, flow / normalizer."""
    flow_full, _ = pwc_forward(pp, img1, img2)
    image = resize_bilinear_legacy(img1, cfg.img_height, cfg.img_width)
    flow = resize_bilinear_legacy(flow_full, cfg.img_height, cfg.img_width) / cfg.flow_normalizer
    return image, flow, flow_full


class TFAdam:
    """tf.train.AdamOptimizer (semantics J).  ONE optimizer object serves both
    train ops (models/adversarial_learner.py:216-237) so the beta-power
    accumulators are shared and advance on every apply."""

    def __init__(self, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.b1p, self.b2p = beta1, beta2  # TF initialises the power accumulators to beta
        self.m, self.v = {}, {}

    def apply(self, params: dict, grads: dict):
        lr_t = self.lr * math.sqrt(1.0 - self.b2p) / (1.0 - self.b1p)
        for k, g in grads.items():
            if k not in self.m:
                self.m[k] = torch.zeros_like(params[k])
                self.v[k] = torch.zeros_like(params[k])
            self.m[k] = self.m[k] + (g - self.m[k]) * (1.0 - self.b1)
            self.v[k] = self.v[k] + (g * g - self.v[k]) * (1.0 - self.b2)
            params[k] = params[k] - lr_t * self.m[k] / (torch.sqrt(self.v[k]) + self.eps)
        self.b1p *= self.b1
        self.b2p *= self.b2


def clip_or_noise(grads: dict, clip=0.2, can_change=False, noise_fn=None):
    """models/utils/loss_utils.py:12-32.  can_change: if mean_v(mean|g_v|) <
    1e-5 every grad <- abs(U(-clip,clip)) else clip(g, +-clip).  noise_fn(name,
    shape) supplies the uniform draw so that callers control the stream."""
    if can_change:
        avg = torch.stack([g.abs().mean() for g in grads.values()]).mean()
        if float(avg) < 1e-5:
            return {k: noise_fn(k, g.shape).abs() for k, g in grads.items()}, True
    return {k: g.clamp(-clip, clip) for k, g in grads.items()}, False


def grads_of(loss, params: dict):
    names = list(params.keys())
    gs = torch.autograd.grad(loss, [params[k] for k in names], retain_graph=True, allow_unused=False)
    return OrderedDict(zip(names, gs))


# --------------------------------------------------------------------------
# synthetic weights / inputs
# --------------------------------------------------------------------------


def _fans(shape):
    if len(shape) == 1:
        return shape[0], shape[0]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def init_params(specs, seed=8964, dtype=torch.float32):
    """Seeded synthetic weights with the reference's initializer families
    (he_normal = truncated normal, fan-in, scale 2; glorot/xavier uniform)."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape, init in specs:
        if init == "zeros":
            a = np.zeros(shape, np.float32)
        elif init == "ones":
            a = np.ones(shape, np.float32)
        elif init == "glorot_uniform":
            fi, fo = _fans(shape)
            lim = math.sqrt(6.0 / (fi + fo))
            a = rng.uniform(-lim, lim, size=shape).astype(np.float32)
        elif init == "he_normal":
            fi, _ = _fans(shape)
            std = math.sqrt(2.0 / fi) / 0.87962566103423978
            a = rng.standard_normal(size=shape)
            bad = np.abs(a) > 2.0
            while bad.any():
                a[bad] = rng.standard_normal(size=int(bad.sum()))
                bad = np.abs(a) > 2.0
            a = (a * std).astype(np.float32)
        else:
            raise ValueError(init)
        out[name] = torch.from_numpy(a).to(dtype)
    return out
