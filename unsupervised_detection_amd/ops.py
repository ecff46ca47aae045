"""Functional single-op wrappers over libudet.so on PyTorch-ROCm tensors (NHWC fp32).

Names and argument meaning follow the reference functions they replace; see include/udet.h."""
from __future__ import annotations

import torch

from ._ffi import check, lib

ACT = {"none": 0, "identity": 0, "leaky": 1, "elu": 2}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous float32 CUDA(HIP) tensor")
    return t


_ws_cache = {}


def _workspace(nbytes: int, device):
    key = str(device)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def dense_image_warp(image, flow, flow_scale: float = 1.0, debug: bool = False):
    """models/PWCNet/core_warp.py:153-202 applied to flow*flow_scale."""
    _chk(image, "image"); _chk(flow, "flow")
    n, h, w, c = image.shape
    if flow.shape != (n, h, w, 2):
        raise ValueError("flow must be [n,h,w,2]")
    out = torch.empty_like(image)
    if debug:
        idx = torch.empty((n, h, w, 2), dtype=torch.int32, device=image.device)
        alpha = torch.empty((n, h, w, 2), dtype=torch.float32, device=image.device)
        check(lib.udet_warp_debug(image.data_ptr(), flow.data_ptr(), flow_scale, out.data_ptr(), idx.data_ptr(),
                                  alpha.data_ptr(), n, h, w, c, _stream()))
        return out, idx, alpha
    check(lib.udet_warp(image.data_ptr(), flow.data_ptr(), flow_scale, out.data_ptr(), n, h, w, c, _stream()))
    return out


def cost_volume(c1, warp, search_range: int = 4):
    """models/PWCNet/core_costvol.py:20-40 (search_range fixed to 4 like the reference options)."""
    if search_range != 4:
        raise ValueError("only search_range=4 (model_pwcnet.py:13) is built")
    _chk(c1, "c1"); _chk(warp, "warp")
    n, h, w, c = c1.shape
    out = torch.empty((n, h, w, 81), dtype=torch.float32, device=c1.device)
    check(lib.udet_cost_volume(c1.data_ptr(), warp.data_ptr(), out.data_ptr(), n, h, w, c, _stream()))
    return out


def _same_out(n, k, s, d):
    return -(-n // s)


def conv2d(x, w_hwio, bias=None, stride=1, dilation=1, act="none", alpha=0.0, upsample2x=False):
    """tf.nn.conv2d(padding='SAME') + bias + activation; optional fused NN x2 upsample of x."""
    _chk(x, "x"); _chk(w_hwio, "w")
    n, h, w, cin = x.shape
    kh, kw, cin2, cout = w_hwio.shape
    if cin2 != cin:
        raise ValueError("weight/input channel mismatch")
    us = 2 if upsample2x else 1
    oh, ow = _same_out(h * us, kh, stride, dilation), _same_out(w * us, kw, stride, dilation)
    y = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=x.device)
    nb = lib.udet_conv2d_workspace_bytes(n, h, w, cin, cout, kh, kw, int(upsample2x))
    ws = _workspace(nb, x.device)
    check(lib.udet_conv2d(x.data_ptr(), w_hwio.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                          n, h, w, cin, cout, kh, kw, stride, dilation, int(upsample2x), ACT[act], alpha, ws.data_ptr(),
                          ws.numel(), _stream()))
    return y


def conv2d_backward_data(dy, y_saved, w_hwio, in_hw, stride=1, dilation=1, act="none", alpha=0.0):
    _chk(dy, "dy"); _chk(w_hwio, "w")
    kh, kw, cin, cout = w_hwio.shape
    n = dy.shape[0]
    h, w = in_hw
    dx = torch.empty((n, h, w, cin), dtype=torch.float32, device=dy.device)
    nb = lib.udet_conv2d_workspace_bytes(n, h, w, cin, cout, kh, kw, 0)
    ws = _workspace(nb, dy.device)
    check(lib.udet_conv2d_backward_data(dy.data_ptr(), y_saved.data_ptr() if y_saved is not None else None,
                                        w_hwio.data_ptr(), dx.data_ptr(), n, h, w, cin, cout, kh, kw, stride, dilation,
                                        ACT[act], alpha, ws.data_ptr(), ws.numel(), _stream()))
    return dx


def conv2d_backward_filter(x, dy, y_saved, ksize, stride=1, dilation=1, act="none", alpha=0.0, upsample2x=False):
    _chk(x, "x"); _chk(dy, "dy")
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    kh, kw = ksize
    dw = torch.empty((kh, kw, cin, cout), dtype=torch.float32, device=x.device)
    db = torch.empty((cout,), dtype=torch.float32, device=x.device)
    nb = lib.udet_conv2d_workspace_bytes(n, h, w, cin, cout, kh, kw, int(upsample2x))
    ws = _workspace(nb, x.device)
    check(lib.udet_conv2d_backward_filter(x.data_ptr(), dy.data_ptr(), y_saved.data_ptr() if y_saved is not None else None,
                                          dw.data_ptr(), db.data_ptr(), n, h, w, cin, cout, kh, kw, stride, dilation,
                                          int(upsample2x), ACT[act], alpha, ws.data_ptr(), ws.numel(), _stream()))
    return dw, db


def conv2d_transpose4x4s2(x, w_hwoi, bias=None):
    """tf.layers.conv2d_transpose(x, cout, 4, 2, 'same') (model_pwcnet.py:283-286); w [4,4,cout,cin]."""
    _chk(x, "x"); _chk(w_hwoi, "w")
    n, h, w, cin = x.shape
    cout = w_hwoi.shape[2]
    y = torch.empty((n, 2 * h, 2 * w, cout), dtype=torch.float32, device=x.device)
    nb = lib.udet_conv2d_workspace_bytes(n, 2 * h, 2 * w, cin, cout, 4, 4, 0)
    ws = _workspace(nb, x.device)
    check(lib.udet_conv2d_transpose4x4s2(x.data_ptr(), w_hwoi.data_ptr(), bias.data_ptr() if bias is not None else None,
                                         y.data_ptr(), n, h, w, cin, cout, ws.data_ptr(), ws.numel(), _stream()))
    return y


def resize_bilinear_legacy(x, out_h: int, out_w: int):
    """tf.image.resize_images(x, [out_h, out_w]) with TF-1.13 legacy bilinear sampling."""
    _chk(x, "x")
    n, h, w, c = x.shape
    if (h, w) == (out_h, out_w):
        return x
    y = torch.empty((n, out_h, out_w, c), dtype=torch.float32, device=x.device)
    check(lib.udet_resize_bilinear_legacy_fwd(x.data_ptr(), y.data_ptr(), n, h, w, c, out_h, out_w, _stream()))
    return y


def resize_bilinear_legacy_backward(dy, in_h: int, in_w: int):
    _chk(dy, "dy")
    n, oh, ow, c = dy.shape
    dx = torch.empty((n, in_h, in_w, c), dtype=torch.float32, device=dy.device)
    check(lib.udet_resize_bilinear_legacy_bwd(dy.data_ptr(), dx.data_ptr(), n, in_h, in_w, c, oh, ow, _stream()))
    return dx


def warp_cost_volume(c1, c2, flow=None, flow_scale: float = 1.0, return_warped: bool = False):
    """cost_volume(c1, dense_image_warp(c2, flow*flow_scale)) in one launch (model_pwcnet.py:616-623); flow=None: no warp."""
    _chk(c1, "c1"); _chk(c2, "c2")
    n, h, w, c = c1.shape
    if c2.shape != c1.shape or (flow is not None and tuple(flow.shape) != (n, h, w, 2)):
        raise ValueError("c2 must match c1 and flow must be [n,h,w,2]")
    if flow is not None:
        _chk(flow, "flow")
    corr = torch.empty((n, h, w, 81), dtype=torch.float32, device=c1.device)
    warped = torch.zeros_like(c2) if return_warped else None
    check(lib.udet_warp_cost_volume(c1.data_ptr(), c2.data_ptr(), flow.data_ptr() if flow is not None else None, flow_scale,
                                    corr.data_ptr(), warped.data_ptr() if warped is not None else None, n, h, w, c, _stream()))
    return (corr, warped) if return_warped else corr


def _stage_ws(n: int, device):
    return _workspace(int(lib.udet_stage_workspace_bytes(n)), device)


def preprocess_flow_batch(flow):
    """models/utils/flow_utils.py:5-12: per sample and channel zero-mean / unit-std over H,W (population variance)."""
    _chk(flow, "flow")
    n, h, w, c = flow.shape
    if c != 2:
        raise ValueError("flow must be [n,h,w,2]")
    out = torch.empty_like(flow)
    ws = _stage_ws(n, flow.device)
    check(lib.udet_flow_normalize(flow.data_ptr(), out.data_ptr(), n, h, w, ws.data_ptr(), ws.numel(), _stream()))
    return out


def charbonnier_loss(gt_flows, pred_flows, masks=None, cbn: float = 0.5):
    """models/utils/loss_utils.py:34-51 -> [B] (masks: [B,H,W,1] broadcast, [B,H,W,2], or None = ones)."""
    _chk(gt_flows, "gt_flows"); _chk(pred_flows, "pred_flows")
    n, h, w, c = gt_flows.shape
    if c != 2 or pred_flows.shape != gt_flows.shape:
        raise ValueError("gt_flows / pred_flows must be [n,h,w,2]")
    mc = 1
    if masks is not None:
        _chk(masks, "masks")
        mc = masks.shape[3]
        if tuple(masks.shape[:3]) != (n, h, w) or mc not in (1, 2):
            raise ValueError("masks must be [n,h,w,1] or [n,h,w,2]")
    out = torch.empty((n,), dtype=torch.float32, device=gt_flows.device)
    ws = _stage_ws(n, gt_flows.device)
    check(lib.udet_charbonnier_loss(gt_flows.data_ptr(), pred_flows.data_ptr(), masks.data_ptr() if masks is not None else None, mc,
                                    n, h, w, cbn, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out


LOSS_KEYS = ("generator", "recover", "red_rate", "red_rate_compl", "reconstruction_loss", "reconstruction_compl_loss",
             "denominator_red_rate", "denominator_red_rate_compl")  # models/adversarial_learner.py:196-204


def losses_forward(flow, mask, pred3, cbn: float = 0.5, epsilon: float = 75.0):
    """The losses{} dictionary of adversarial_learner.py:141-204 as a device tensor [8] (order LOSS_KEYS) + the per-sample
    coefficients [B,4] the generator-loss backward consumes."""
    _chk(flow, "flow"); _chk(mask, "mask"); _chk(pred3, "pred3")
    b, h, w, _ = flow.shape
    if tuple(mask.shape) != (b, h, w, 1) or tuple(pred3.shape) != (3 * b, h, w, 2):
        raise ValueError("mask must be [b,h,w,1] and pred3 [3b,h,w,2]")
    losses = torch.empty((8,), dtype=torch.float32, device=flow.device)
    coef = torch.empty((b, 4), dtype=torch.float32, device=flow.device)
    ws = _stage_ws(b, flow.device)
    check(lib.udet_losses_forward(flow.data_ptr(), mask.data_ptr(), pred3.data_ptr(), b, h, w, cbn, epsilon, losses.data_ptr(),
                                  coef.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return losses, coef


def losses_backward(flow, mask, pred3, which: str, coef=None, cbn: float = 0.5):
    """tf.gradients of losses['recover'] (which='recover' -> dpred [3b,h,w,2]) or losses['generator'] (which='generator' ->
    dpred [2b,h,w,2] and the direct mask term dmask [b,h,w,1]) w.r.t. the recover predictions."""
    _chk(flow, "flow"); _chk(mask, "mask"); _chk(pred3, "pred3")
    b, h, w, _ = flow.shape
    if which == "recover":
        dpred = torch.empty_like(pred3)
        check(lib.udet_losses_backward(flow.data_ptr(), mask.data_ptr(), pred3.data_ptr(), None, 2, b, h, w, cbn, dpred.data_ptr(), None,
                                       _stream()))
        return dpred
    if which != "generator" or coef is None:
        raise ValueError("which must be 'recover' or 'generator' (the latter with coef from losses_forward)")
    dpred = torch.empty((2 * b, h, w, 2), dtype=torch.float32, device=flow.device)
    dmask = torch.empty((b, h, w, 1), dtype=torch.float32, device=flow.device)
    check(lib.udet_losses_backward(flow.data_ptr(), mask.data_ptr(), pred3.data_ptr(), _chk(coef, "coef").data_ptr(), 1, b, h, w, cbn,
                                   dpred.data_ptr(), dmask.data_ptr(), _stream()))
    return dpred, dmask


def clip_or_noise_(g, clip: float = 0.2, flag2=None, seed: int = 8964, step: int = 0):
    """In place: g <- |U(-clip,clip)| if flag2[1] != 0 else clip(g, +-clip)  (loss_utils.py:22-31)."""
    _chk(g, "g")
    check(lib.udet_clip_or_noise(g.data_ptr(), g.numel(), clip, flag2.data_ptr() if flag2 is not None else None, seed, step, _stream()))
    return g


def adam_step_(w, g, m, v, t: int, lr: float = 1e-4, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
    """In place tf.train.AdamOptimizer apply on flat buffers; t = 1-based count of applies of the shared optimizer."""
    for n_, x in (("w", w), ("g", g), ("m", m), ("v", v)):
        _chk(x, n_)
    check(lib.udet_adam_step(w.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), w.numel(), lr, beta1, beta2, eps, t, _stream()))
    return w
