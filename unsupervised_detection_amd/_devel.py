"""Loader of libudet_debug.so (include/udet_debug.h): TEST-ONLY hooks that pin a convolution kernel family / tile / split-K
mode and report which one ran.  Used by tests/ and tools/ only -- nothing on the product path imports this module; the
hooks are not exported by libudet.so."""
from __future__ import annotations

import ctypes
import os

from ._ffi import lib as _lib  # libudet.so first: the debug library resolves its internal symbols against it  # noqa: F401

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libudet_debug.so")
if not os.path.exists(_PATH):
    raise RuntimeError(f"{_PATH} is missing: build it with `make -C unsupervised_detection_amd/csrc`")
dbg = ctypes.CDLL(_PATH, mode=ctypes.RTLD_GLOBAL)
dbg.udet_debug_force_conv.restype = None
dbg.udet_debug_force_conv.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
dbg.udet_debug_last_conv.restype = ctypes.c_int
dbg.udet_debug_last_conv.argtypes = []
dbg.udet_debug_conv_fp16.restype = None
dbg.udet_debug_conv_fp16.argtypes = [ctypes.c_int]
dbg.udet_debug_force_wgrad.restype = None
dbg.udet_debug_force_wgrad.argtypes = [ctypes.c_int, ctypes.c_int]
dbg.udet_debug_upb_min_pixels.restype = None
dbg.udet_debug_upb_min_pixels.argtypes = [ctypes.c_long]
dbg.udet_debug_set_tuning.restype = None
dbg.udet_debug_set_tuning.argtypes = [ctypes.c_int]
dbg.udet_debug_last_wgrad.restype = ctypes.c_int
dbg.udet_debug_last_wgrad.argtypes = []
dbg.udet_debug_force_pair.restype = None
dbg.udet_debug_force_pair.argtypes = [ctypes.c_int]
dbg.udet_debug_last_pair.restype = ctypes.c_int
dbg.udet_debug_last_pair.argtypes = []
dbg.udet_debug_conv2d_pair.restype = ctypes.c_int
dbg.udet_debug_conv2d_pair.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_int] * 10 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
