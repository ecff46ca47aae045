"""ctypes binding of libudet.so (include/udet.h).

The HIP library is the product: there is no CPU / PyTorch fallback.  Importing this
module without a built ``libudet.so`` raises, and every call that fails raises with
``udet_last_error()``.
"""
from __future__ import annotations

import ctypes
import os

# PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so); libudet.so names the runtime by soname.  Loaded AFTER torch the
# library binds to the runtime torch already initialised; loaded first it would bring /opt/rocm's copy in, torch would add its own, and
# whichever initialises second sees "no ROCm-capable device".  Hence: torch first, always.
import torch  # noqa: F401  (load order, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libudet.so")

if not os.path.exists(LIB_PATH):
    raise RuntimeError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C unsupervised_detection_amd/csrc`).  There is no CPU fallback.")

lib = ctypes.CDLL(LIB_PATH)

c_f = ctypes.c_float
c_i = ctypes.c_int
c_p = ctypes.c_void_p
c_sz = ctypes.c_size_t

lib.udet_version.restype = c_i
lib.udet_last_error.restype = ctypes.c_char_p


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_sig("udet_warp", c_i, c_p, c_p, c_f, c_p, c_i, c_i, c_i, c_i, c_p)
_sig("udet_warp_debug", c_i, c_p, c_p, c_f, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p)
_sig("udet_cost_volume", c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p)
_sig("udet_resize_bilinear_legacy_fwd", c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p)
_sig("udet_resize_bilinear_legacy_bwd", c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p)
_sig("udet_conv2d_workspace_bytes", c_sz, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i)
_sig("udet_conv2d", c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_sz, c_p)
_sig("udet_conv2d_backward_data", c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p,
     c_sz, c_p)
_sig("udet_conv2d_backward_filter", c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
     c_f, c_p, c_sz, c_p)
_sig("udet_conv2d_transpose4x4s2", c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p)


_sig("udet_warp_cost_volume", c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_i, c_i, c_i, c_p)
_sig("udet_stage_workspace_bytes", c_sz, c_i)
_sig("udet_flow_normalize", c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_sz, c_p)
_sig("udet_charbonnier_loss", c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_sz, c_p)
_sig("udet_losses_forward", c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_p, c_p, c_p, c_sz, c_p)
_sig("udet_losses_backward", c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p)
_sig("udet_clip_or_noise", c_i, c_p, c_sz, c_f, c_p, ctypes.c_ulonglong, ctypes.c_long, c_p)
_sig("udet_adam_step", c_i, c_p, c_p, c_p, c_p, c_sz, c_f, c_f, c_f, c_f, ctypes.c_long, c_p)
_sig("udet_generator_layers", c_i, c_p, c_p, c_p)
_sig("udet_generator_backward", c_i, c_p, c_p, c_p, c_p, c_p)
_sig("udet_recover_backward", c_i, c_p, c_p, c_p, c_p, c_p)
_sig("udet_grad_absmean", c_i, c_p, c_i, c_p, c_p, c_p, c_p)
_sig("udet_stream_wait_grads", c_i, c_p, c_i, c_p)
_sig("udet_tune_rejected", c_i)
_sig("udet_tune_save", c_i, ctypes.c_char_p)
_sig("udet_tune_load", c_i, ctypes.c_char_p)


class UdetError(RuntimeError):
    pass


class UdetOverflow(OverflowError):
    """UDET_ERR_OVERFLOW (-6): an fp16-mode optimizer update was dropped on the device (include/udet.h, udet_config.conv_fp16).
    Its own type, so that callers which retry on it (trainer.train_step) do not swallow Python's / ctypes' own OverflowError
    (argument marshalling of a too-large integer, for example)."""


def check(status: int):
    if status != 0:
        msg = lib.udet_last_error().decode("utf-8", "replace")
        if status in (-1, -2, -5):
            raise ValueError(f"libudet error {status}: {msg}")
        if status == -6:
            raise UdetOverflow(f"libudet error {status}: {msg}")
        raise UdetError(f"libudet error {status}: {msg}")
