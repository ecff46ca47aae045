"""Host wrapper of the step plan (udet_plan_* in include/udet.h): owns the workspace as one
torch byte tensor and exposes its regions as zero-copy float32 views."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import torch

from ._ffi import c_f, c_i, c_p, c_sz, check, lib
from .weights import NET_GEN, NET_REC  # noqa: F401  (re-exported: the net ids udet_apply takes)


class _Cfg(ctypes.Structure):
    _fields_ = [("batch", c_i), ("in_h", c_i), ("in_w", c_i), ("img_h", c_i), ("img_w", c_i), ("flow_normalizer", c_f),
                ("cbn", c_f), ("epsilon", c_f), ("lr", c_f), ("beta1", c_f), ("beta2", c_f), ("adam_eps", c_f), ("clip", c_f),
                ("noise_seed", ctypes.c_ulonglong), ("conv_fp16", c_i)]


lib.udet_plan_create.restype = c_i
lib.udet_plan_create.argtypes = [ctypes.POINTER(_Cfg), ctypes.POINTER(c_p)]
lib.udet_plan_destroy.restype = None
lib.udet_plan_destroy.argtypes = [c_p]
lib.udet_workspace_bytes.restype = c_sz
lib.udet_workspace_bytes.argtypes = [c_p]
lib.udet_plan_init.restype = c_i
lib.udet_plan_init.argtypes = [c_p, c_p, c_p]
lib.udet_buffer_count.restype = c_i
lib.udet_buffer_count.argtypes = [c_p]
lib.udet_buffer_info.restype = c_i
lib.udet_buffer_info.argtypes = [c_p, c_i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_sz), ctypes.POINTER(c_i)]
lib.udet_plan_set_concurrent.restype = c_i
lib.udet_plan_set_concurrent.argtypes = [c_p, c_i]
lib.udet_plan_lane_queues.restype = c_i
lib.udet_plan_lane_queues.argtypes = [c_p, c_p, ctypes.POINTER(c_i)]
lib.udet_plan_pin_lanes.restype = c_i
lib.udet_plan_pin_lanes.argtypes = [c_p, c_p, ctypes.POINTER(c_p), c_i]
lib.udet_fp16_overflow_count.restype = ctypes.c_long
lib.udet_fp16_overflow_count.argtypes = [c_p]
lib.udet_get_adam_step.restype = ctypes.c_long
lib.udet_get_adam_step.argtypes = [c_p]
lib.udet_set_adam_step.restype = None
lib.udet_set_adam_step.argtypes = [c_p, ctypes.c_long]
for _n, _a in (("udet_pack_pwc", [c_p, c_p, c_p, c_p]), ("udet_pack_trainable", [c_p, c_p, c_p, c_p, c_p]),
               ("udet_pwc_forward", [c_p, c_p, c_p, c_p, c_p]), ("udet_forward", [c_p, c_p, c_p, c_i, c_p, c_p]),
               ("udet_forward_from_flow", [c_p, c_i, c_p, c_p]), ("udet_generator_forward", [c_p, c_p, c_p]),
               ("udet_recover_forward", [c_p, c_i, c_p, c_p]),
               ("udet_prefetch_flow", [c_p, c_p, c_p, c_p, c_p]), ("udet_forward_prefetched", [c_p, c_i, c_p, c_p]),
               ("udet_prefetch_consume", [c_p, c_p, c_p]),
               ("udet_backward", [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
               ("udet_apply", [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
               ("udet_train_step", [c_p, c_i] + [c_p] * 12)):
    getattr(lib, _n).restype = c_i
    getattr(lib, _n).argtypes = _a

lib.udet_autotune.restype = c_i
lib.udet_autotune.argtypes = [c_p] * 7
lib.udet_tuned_shapes.restype = c_i
lib.udet_tuned_shapes.argtypes = []
lib.udet_profile_begin.restype = c_i
lib.udet_profile_begin.argtypes = [c_p]
lib.udet_profile_end.restype = c_i
lib.udet_profile_end.argtypes = [c_p, ctypes.POINTER(ctypes.c_double), c_i, c_p]

GEN, REC, BOTH = 1, 2, 3
LOSS_KEYS = ("generator", "recover", "red_rate", "red_rate_compl", "reconstruction_loss", "reconstruction_compl_loss",
             "denominator_red_rate", "denominator_red_rate_compl")  # models/adversarial_learner.py:196-204


@dataclass
class EngineConfig:
    """Hot-path flags of common_flags.py:6-21 (same names, same defaults) + the PWC input size the readers produce."""
    batch_size: int = 4
    in_height: int = 384
    in_width: int = 640
    img_height: int = 192
    img_width: int = 384
    flow_normalizer: float = 80.0
    cbn: float = 0.5
    epsilon: float = 75.0
    beta1: float = 0.9
    lr: float = 1e-4
    beta2: float = 0.999
    adam_eps: float = 1e-8
    clip: float = 0.2
    noise_seed: int = 8964
    conv_fp16: bool = False  # BASELINE.json configs[4]: fp16 multiplication (fp32 accumulation) in the convolution GEMMs


def _ptr(t):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("expected a contiguous float32 CUDA(HIP) tensor")
    return t.data_ptr()


class Engine:
    def __init__(self, cfg: EngineConfig, device="cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("unsupervised_detection_amd needs an MI355X (no CPU fallback exists)")
        self.cfg = cfg
        self.device = torch.device(device)
        c = _Cfg(cfg.batch_size, cfg.in_height, cfg.in_width, cfg.img_height, cfg.img_width, cfg.flow_normalizer, cfg.cbn,
                 cfg.epsilon, cfg.lr, cfg.beta1, cfg.beta2, cfg.adam_eps, cfg.clip, cfg.noise_seed, 1 if cfg.conv_fp16 else 0)
        h = c_p()
        check(lib.udet_plan_create(ctypes.byref(c), ctypes.byref(h)))
        self._h = h
        self.workspace_bytes = int(lib.udet_workspace_bytes(h))
        self.ws = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=self.device)
        self._bufs = {}
        for i in range(lib.udet_buffer_count(h)):
            name = ctypes.c_char_p()
            off = c_sz()
            dims = (c_i * 4)()
            check(lib.udet_buffer_info(h, i, ctypes.byref(name), ctypes.byref(off), dims))
            self._bufs[name.value.decode()] = (off.value, tuple(dims))
        check(lib.udet_plan_init(h, self.ws.data_ptr(), self._stream()))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.udet_plan_destroy(h)
            self._h = None

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def buffer_names(self):
        return list(self._bufs)

    def buffer(self, name: str) -> torch.Tensor:
        """float32 view [n,h,w,ld] of a workspace region (ld >= logical channels: padding channels are zero)."""
        off, (n, h, w, ld) = self._bufs[name]
        cnt = n * h * w * ld
        return self.ws[off:off + 4 * cnt].view(torch.float32).view(n, h, w, ld)

    # ---- weights -------------------------------------------------------------------------
    def pack_pwc(self, w_pwc):
        check(lib.udet_pack_pwc(self._h, _ptr(w_pwc), self.ws.data_ptr(), self._stream()))

    def pack_trainable(self, w_gen=None, w_rec=None):
        check(lib.udet_pack_trainable(self._h, _ptr(w_gen), _ptr(w_rec), self.ws.data_ptr(), self._stream()))

    # ---- forward -------------------------------------------------------------------------
    def _check_pair(self, img1, img2):
        """The plan is specialised to one batch shape and the kernels read exactly that many bytes: a short last batch of a
        one-pass reader (or any other shape) must be rejected here, not read out of bounds on the device."""
        c = self.cfg
        want = (c.batch_size, c.in_height, c.in_width, 3)
        for name, t in (("img1", img1), ("img2", img2)):
            if tuple(t.shape) != want:
                raise ValueError("%s has shape %s, this plan takes %s (pad the last batch of a one-pass reader: "
                                 "learner.pad_batch)" % (name, tuple(t.shape), want))

    def pwc_forward(self, img1, img2):
        self._check_pair(img1, img2)
        check(lib.udet_pwc_forward(self._h, _ptr(img1), _ptr(img2), self.ws.data_ptr(), self._stream()))
        return self.buffer("flow_full")

    def forward(self, img1, img2, ncalls=3):
        self._check_pair(img1, img2)
        check(lib.udet_forward(self._h, _ptr(img1), _ptr(img2), ncalls, self.ws.data_ptr(), self._stream()))

    def prefetch_flow(self, img1, img2):
        """PWC flow + resizes of the NEXT step's pair on the plan's side streams (PWC-Net is frozen); overlaps whatever
        is enqueued next.  Keep img1/img2 alive until forward_prefetched()."""
        self._check_pair(img1, img2)
        self._prefetch_keep = (img1, img2)
        check(lib.udet_prefetch_flow(self._h, _ptr(img1), _ptr(img2), self.ws.data_ptr(), self._stream()))

    def prefetch_consume(self):
        """join the pending prefetch and move its flow / image into place (then: prefetch_flow(next), forward_in_place())"""
        check(lib.udet_prefetch_consume(self._h, self.ws.data_ptr(), self._stream()))
        self._prefetch_keep = None

    def forward_in_place(self, ncalls=3):
        """generator + recover forward + losses from the "image" / "flow" buffers as they are"""
        check(lib.udet_forward_from_flow(self._h, ncalls, self.ws.data_ptr(), self._stream()))

    def forward_prefetched(self, ncalls=3):
        check(lib.udet_forward_prefetched(self._h, ncalls, self.ws.data_ptr(), self._stream()))
        self._prefetch_keep = None

    def forward_from_flow(self, image, flow, ncalls=3):
        if tuple(image.shape) != tuple(self.buffer("image").shape) or tuple(flow.shape) != tuple(self.buffer("flow").shape):
            raise ValueError("image / flow must be %s / %s" % (tuple(self.buffer("image").shape), tuple(self.buffer("flow").shape)))
        self.buffer("image").copy_(image)
        self.buffer("flow").copy_(flow)
        check(lib.udet_forward_from_flow(self._h, ncalls, self.ws.data_ptr(), self._stream()))

    def losses(self):
        v = self.buffer("losses").view(-1).cpu().tolist()
        return dict(zip(LOSS_KEYS, v))

    # ---- backward / optimizer ------------------------------------------------------------
    def backward(self, which, w_gen, w_rec, g_gen, g_rec):
        check(lib.udet_backward(self._h, which, _ptr(w_gen), _ptr(w_rec), _ptr(g_gen), _ptr(g_rec), self.ws.data_ptr(),
                                self._stream()))

    def apply(self, net, w, g, m, v):
        check(lib.udet_apply(self._h, net, _ptr(w), _ptr(g), _ptr(m), _ptr(v), self.ws.data_ptr(), self._stream()))

    def train_step(self, which, img1, img2, w_gen, w_rec, g_gen, g_rec, m_gen, v_gen, m_rec, v_rec):
        self._check_pair(img1, img2)
        check(lib.udet_train_step(self._h, which, _ptr(img1), _ptr(img2), _ptr(w_gen), _ptr(w_rec), _ptr(g_gen), _ptr(g_rec),
                                  _ptr(m_gen), _ptr(v_gen), _ptr(m_rec), _ptr(v_rec), self.ws.data_ptr(), self._stream()))

    def autotune(self, w_gen, w_rec, g_gen, g_rec):
        """Time the candidate kernel configurations of every convolution of the plan once (process-wide cache).
        Weights must be packed; g_gen / g_rec are scratch; activation buffers are re-zeroed."""
        check(lib.udet_autotune(self._h, _ptr(w_gen), _ptr(w_rec), _ptr(g_gen), _ptr(g_rec), self.ws.data_ptr(), self._stream()))
        return int(lib.udet_tuned_shapes())

    def profile(self, fn):
        """Run fn() with per-category kernel timing (udet_profile_begin/end). Returns {category: {...}}."""
        cats = ("conv_fwd", "conv_dgrad", "conv_wgrad", "warp", "cost_volume")
        check(lib.udet_profile_begin(self._h))
        try:
            fn()
        finally:
            out = (ctypes.c_double * (5 * len(cats)))()
            check(lib.udet_profile_end(self._h, out, len(cats), self._stream()))
        # ms: sum of the kernels' own start -> stop times; bracket_ms: hipEventRecord around each launch group (adds the event
        # packets and dispatch gaps)
        return {c: dict(groups=out[5 * i], ms=out[5 * i + 1], flops=out[5 * i + 2], bytes=out[5 * i + 3], bracket_ms=out[5 * i + 4])
                for i, c in enumerate(cats)}

    def set_concurrent(self, on: bool):
        """False: every lane of the plan collapses onto the caller's stream (plain program order, bit-identical results)."""
        check(lib.udet_plan_set_concurrent(self._h, 1 if on else 0))

    def lane_queues(self):
        """(independent hardware queues in use, queue-group index of each of the plan's six lanes) for the current stream; places the
        lanes (a probe that synchronises the device once) if this stream has not driven the plan yet -- include/udet.h."""
        q = (c_i * 6)()
        n = int(lib.udet_plan_lane_queues(self._h, self._stream(), q))
        if n < 0:
            check(n)
        return n, [int(v) for v in q]

    def pin_lanes(self, side_streams):
        """Pin the plan's lane layout for the current stream to up to three torch streams the caller knows to sit on distinct hardware
        queues (include/udet.h: udet_plan_pin_lanes) -- no timing probe.  The streams are kept alive by this engine."""
        side_streams = list(side_streams)
        arr = (c_p * max(1, len(side_streams)))(*[s.cuda_stream for s in side_streams])
        check(lib.udet_plan_pin_lanes(self._h, self._stream(), arr, len(side_streams)))
        self._pinned_streams = getattr(self, "_pinned_streams", []) + side_streams

    def fp16_overflow_count(self) -> int:
        """conv_fp16 plans: optimizer updates dropped so far because their gradients were not finite (synchronises)."""
        return int(lib.udet_fp16_overflow_count(self._h))

    @property
    def adam_step(self):
        return int(lib.udet_get_adam_step(self._h))

    @adam_step.setter
    def adam_step(self, t):
        lib.udet_set_adam_step(self._h, int(t))
