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
