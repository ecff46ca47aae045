"""Flat fp32 parameter buffers in TF variable order (tables come from libudet.so) and seeded
synthetic initialisation with the reference's initializer families:
  PWC-Net convs  he_normal (truncated normal, fan-in, scale 2)   models/PWCNet/model_pwcnet.py:153,477,560
  PWC flow{l} heads, transposed convs  glorot_uniform (tf.layers default)   :504,286
  generator      glorot_uniform + zero bias, BN gamma=1 beta=0   models/utils/convolution_utils.py:28,46-50
  recover        xavier_initializer_conv2d (uniform) + zero bias  models/utils/convolution_utils.py:78
Pretrained checkpoints are external downloads (README.md:59-64); see INTEGRATION.md for import."""
from __future__ import annotations

import ctypes
import math
from collections import OrderedDict

import numpy as np
import torch

from ._ffi import check, lib

NET_PWC, NET_GEN, NET_REC = 0, 1, 2
NET_NAMES = {"pwc": NET_PWC, "pwcnet": NET_PWC, "gen": NET_GEN, "generator": NET_GEN, "rec": NET_REC, "recover": NET_REC}

lib.udet_param_count.restype = ctypes.c_int
lib.udet_param_count.argtypes = [ctypes.c_int]
lib.udet_param_total.restype = ctypes.c_size_t
lib.udet_param_total.argtypes = [ctypes.c_int]
lib.udet_param_info.restype = ctypes.c_int
lib.udet_param_info.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int),
                                ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_size_t)]


def param_table(net: int):
    """[(name, shape, offset_floats)] for net in TF variable creation order."""
    out = []
    for i in range(lib.udet_param_count(net)):
        name = ctypes.c_char_p()
        rank = ctypes.c_int()
        shape = (ctypes.c_int * 4)()
        off = ctypes.c_size_t()
        check(lib.udet_param_info(net, i, ctypes.byref(name), ctypes.byref(rank), shape, ctypes.byref(off)))
        out.append((name.value.decode(), tuple(shape[k] for k in range(rank.value)), off.value))
    return out


def param_total(net: int) -> int:
    return int(lib.udet_param_total(net))


def _fans(shape):
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def _init_kind(net, name):
    if name.endswith(("/bias", "/biases", "/beta")):
        return "zeros"
    if name.endswith("/gamma"):
        return "ones"
    if net == NET_PWC and ("/flow" in name.split("predict_flow")[-1] or "/upsample/" in name):
        return "glorot_uniform"
    return "he_normal" if net == NET_PWC else "glorot_uniform"


def init_flat(net: int, seed: int = 8964) -> torch.Tensor:
    """Seeded synthetic weights (CPU float32 flat tensor)."""
    rng = np.random.default_rng(seed + 1000 * net)
    flat = np.zeros(param_total(net), np.float32)
    for name, shape, off in param_table(net):
        n = int(np.prod(shape))
        kind = _init_kind(net, name)
        if kind == "zeros":
            continue
        if kind == "ones":
            flat[off:off + n] = 1.0
        elif kind == "glorot_uniform":
            fi, fo = _fans(shape)
            lim = math.sqrt(6.0 / (fi + fo))
            flat[off:off + n] = rng.uniform(-lim, lim, size=n).astype(np.float32)
        else:
            fi, _ = _fans(shape)
            std = math.sqrt(2.0 / fi) / 0.87962566103423978
            a = rng.standard_normal(n)
            bad = np.abs(a) > 2.0
            while bad.any():
                a[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(a) > 2.0
            flat[off:off + n] = (a * std).astype(np.float32)
    return torch.from_numpy(flat)


def as_dict(flat: torch.Tensor, net: int) -> "OrderedDict[str, torch.Tensor]":
    """name -> view of the flat buffer (TF names)."""
    d = OrderedDict()
    for name, shape, off in param_table(net):
        n = int(np.prod(shape))
        d[name] = flat[off:off + n].view(*shape)
    return d


def from_dict(d, net: int) -> torch.Tensor:
    flat = torch.zeros(param_total(net), dtype=torch.float32)
    for name, shape, off in param_table(net):
        t = d[name]
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {tuple(t.shape)} != {shape}")
        flat[off:off + t.numel()] = t.detach().reshape(-1).to(torch.float32)
    return flat


# --------------------------------------------------------------------------------------------------------------------
# TF variable names.  The reference builds the generator inside `tf.variable_scope("MaskNet/")` (the *name scope* string,
# adversarial_learner.py:99-105) and the recover net inside "FlownetS/", so checkpoint names carry a double slash
# ("MaskNet//conv1/kernel", "FlownetS//aconv1/weights"), and tf.layers auto-names the generator's batch-norm layers per
# enclosing scope ("MaskNet//batch_normalization_3/gamma", "MaskNet//conv13_upsample/batch_normalization/beta").
# tests/golden/names.json holds the full list as created by the reference's own code.
# --------------------------------------------------------------------------------------------------------------------
_GEN_TOP_BN_ORDER = ("conv1", "conv2_downsample", "conv3", "conv4_downsample", "conv5", "conv6", "conv7_atrous", "conv8_atrous",
                     "conv9_atrous", "conv10_atrous", "conv11", "conv12", "conv14", "conv16", "conv17")


def canonical_name(tf_name: str):
    """TF-1 variable name (optionally with a ':0' suffix) -> the name used by param_table(), or None for variables the
    path does not read (BN moving statistics -- never updated by the reference, see SURVEY 8c-F --, optimizer slots,
    global_step)."""
    n = tf_name.split(":")[0].replace("//", "/")
    parts = n.split("/")
    if parts[-1] in ("moving_mean", "moving_variance", "Adam", "Adam_1") or parts[0] not in ("MaskNet", "FlownetS", "pwcnet"):
        return None
    if parts[0] == "MaskNet" and len(parts) >= 3 and parts[-2].startswith("batch_normalization"):
        if len(parts) == 3:  # k-th default-named layer of the top scope = k-th gen_conv outside the upsample scopes
            suf = parts[1][len("batch_normalization"):]
            k = int(suf[1:]) if suf else 0
            if k >= len(_GEN_TOP_BN_ORDER):
                raise KeyError(tf_name)
            layer = _GEN_TOP_BN_ORDER[k]
        else:
            layer = parts[1]
        return "MaskNet/%s/bn/%s" % (layer, parts[-1])
    return n


def from_tf_dict(d, net: int) -> torch.Tensor:
    """Flat buffer from {TF variable name: array} as exported from a reference checkpoint (e.g. with
    tf.train.load_checkpoint(...).get_tensor on a machine that has TensorFlow); names are canonicalised first.
    Raises KeyError naming the first variable of `net` that the mapping lacks."""
    canon = {}
    for k, v in d.items():
        c = canonical_name(k)
        if c is not None:
            canon[c] = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
    return from_dict(canon, net)
