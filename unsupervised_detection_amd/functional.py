"""Function-level surface of the reference's hot path, backed by libudet.so.

The reference's entry scripts and its learner import these names:
    generator_net, recover_net                      models/nets.py:4,45
    ModelPWCNet().predict_from_img_pairs            models/PWCNet/model_pwcnet.py:22,61
    train_op, charbonnier_loss                      models/utils/loss_utils.py:12,34
    preprocess_flow_batch                           models/utils/flow_utils.py:5
`unsupervised_detection_amd.models.*` re-exports them under the reference's module paths.

TF-1 builds a graph whose variables live in a process-wide collection keyed by scope ("MaskNet/", "FlownetS/", "pwcnet");
here the same role is played by `variables(net)`: one flat fp32 device buffer per network in TF variable order
(weights.param_table), created on first use with the reference's initializer families -- exactly when tf.get_variable
would create them -- or assigned from a checkpoint (`assign_variables`).  Tensors are NHWC float32 on the GPU; every
function below is a driver over the HIP library (there is no torch / CPU arithmetic on the path).
"""
from __future__ import annotations

import torch

from . import ops
from . import weights as W
from ._ffi import check, lib
from .engine import GEN, REC, Engine, EngineConfig

_SCOPE_NET = {"MaskNet": W.NET_GEN, "FlownetS": W.NET_REC, "pwcnet": W.NET_PWC}
_store = {}      # net -> {"w": flat device tensor, "version": int}
_engines = {}    # (batch, img_hw, in_hw, flags) -> {"engine": Engine, "packed": {net: version}}
_seed = 8964


def _net_of(scope, default):
    if scope is None:
        return default
    key = str(scope).strip("/").split("/")[0]
    if key not in _SCOPE_NET:
        raise ValueError("unknown variable scope %r (the path has MaskNet/, FlownetS/ and pwcnet)" % (scope,))
    return _SCOPE_NET[key]


def set_seed(seed: int):
    """Seed of the initializers used when a scope's variables are created on first use (tf.set_random_seed analogue)."""
    global _seed
    _seed = int(seed)


def variables(net, device="cuda") -> torch.Tensor:
    """The flat weight buffer of a network (W.NET_PWC / NET_GEN / NET_REC or a scope string); created on first use with
    the reference's initializers (convolution_utils.py:28,46-50,78; model_pwcnet.py:153,286,477,504)."""
    if isinstance(net, str):
        net = _net_of(net, None)
    if net not in _store:
        _store[net] = {"w": W.init_flat(net, _seed).to(device), "version": 0}
    return _store[net]["w"]


def assign_variables(net, flat: torch.Tensor):
    """Saver.restore analogue: replace a network's variables by a flat buffer (weights.from_dict / from_tf_dict)."""
    if isinstance(net, str):
        net = _net_of(net, None)
    if flat.numel() != W.param_total(net):
        raise ValueError("expected %d values for net %d, got %d" % (W.param_total(net), net, flat.numel()))
    cur = _store.get(net)
    if cur is None:
        _store[net] = {"w": flat.detach().to("cuda", torch.float32).contiguous().clone(), "version": 0}
    else:
        cur["w"].copy_(flat)
        cur["version"] += 1


def mark_updated(net):
    """Call after modifying variables(net) in place (an optimizer step): engines re-pack the weights before their next use."""
    _store[net]["version"] += 1


def _engine(batch, img_hw=(64, 64), in_hw=(64, 64), nets=(), **flags):
    key = (batch, tuple(img_hw), tuple(in_hw), tuple(sorted(flags.items())))
    ent = _engines.get(key)
    if ent is None:
        eng = Engine(EngineConfig(batch_size=batch, in_height=in_hw[0], in_width=in_hw[1], img_height=img_hw[0],
                                  img_width=img_hw[1], **flags))
        ent = _engines[key] = {"engine": eng, "packed": {}}
    eng = ent["engine"]
    for net in nets:
        w = variables(net)
        ver = _store[net]["version"]
        if ent["packed"].get(net) != ver:
            if net == W.NET_PWC:
                eng.pack_pwc(w)
            else:
                eng.pack_trainable(w if net == W.NET_GEN else None, w if net == W.NET_REC else None)
            ent["packed"][net] = ver
    return eng


def _nhwc(t, c, name):
    if t.dim() != 4 or t.shape[3] != c:
        raise ValueError("%s must be [B,H,W,%d]" % (name, c))
    return t.contiguous()


# ------------------------------------------------------------------------------------------------ networks ----
def generator_net(images, flows, scope="MaskNet/", reuse=None, training=True):
    """Mask network (models/nets.py:4-42): images [B,H,W,3] in [-0.5,0.5], flows [B,H,W,2] ALREADY standardised by
    preprocess_flow_batch (adversarial_learner.py:99-105 passes preprocess_flow_batch(flow)); returns the mask [B,H,W,1]
    (channel 0 of softmax(logits / 10)).  BatchNorm runs in inference mode with moving statistics (0, 1) exactly as the
    reference graph does, whatever `training` says."""
    images, flows = _nhwc(images, 3, "images"), _nhwc(flows, 2, "flows")
    B, H, Wd, _ = images.shape
    e = _engine(B, (H, Wd), nets=(_net_of(scope, W.NET_GEN),))
    gin = e.buffer("gen.in")
    gin[..., 0:3] = images
    gin[..., 3:5] = flows
    check(lib.udet_generator_layers(e._h, e.ws.data_ptr(), e._stream()))
    return e.buffer("mask").clone()


def recover_net(img1, flow_masked, mask, scope="FlownetS/", reuse=None, f=0.25, training=True):
    """Inpainter (models/nets.py:45-110): img1 [B,H,W,3], flow_masked [B,H,W,2], mask [B,H,W,1] -> flow [B,H,W,2]."""
    img1, flow_masked, mask = _nhwc(img1, 3, "img1"), _nhwc(flow_masked, 2, "flow_masked"), _nhwc(mask, 1, "mask")
    B, H, Wd, _ = img1.shape
    e = _engine(B, (H, Wd), nets=(_net_of(scope, W.NET_REC),))
    fin, imgin = e.buffer("rec.fin"), e.buffer("rec.imgin")
    fin[:B, ..., 0:2] = flow_masked
    fin[:B, ..., 2:3] = 1.0          # ones_x (nets.py:50)
    fin[:B, ..., 3:4] = 1.0 - mask   # 1 - mask (nets.py:51-52)
    imgin[:B, ..., 0:3] = img1
    check(lib.udet_recover_forward(e._h, 1, e.ws.data_ptr(), e._stream()))
    return e.buffer("pred")[:B].clone()


class ModelPWCNet(object):
    """models/PWCNet/model_pwcnet.py:22-76 (the lg-6-2 test configuration the learner instantiates, :18)."""

    def __init__(self, name="pwcnet", options=None):
        self.name = name
        self.opts = options
        self.dbg = False

    def adapt_x(self, img1s, img2s):
        """:39-59: images from [-0.5,0.5] to [0,1], stacked as [N,2,H,W,3]."""
        return torch.cat(((img1s + 0.5).unsqueeze(1), (img2s + 0.5).unsqueeze(1)), dim=1)

    def nn(self, x_adapt):
        """:599-649 on an adapt_x result; returns (flow_pred [N,H,W,2], None) -- the pyramid is not materialised."""
        return self._flow(x_adapt[:, 0] - 0.5, x_adapt[:, 1] - 0.5), None

    def predict_from_img_pairs(self, img1s, img2s):
        """:61-76: flow from img1 to img2, [N,H,W,2] at the input resolution (H, W multiples of 64)."""
        return self._flow(img1s, img2s)

    def _flow(self, img1s, img2s):
        img1s, img2s = _nhwc(img1s, 3, "img1s"), _nhwc(img2s, 3, "img2s")
        B, H, Wd, _ = img1s.shape
        e = _engine(B, in_hw=(H, Wd), nets=(_net_of(self.name, W.NET_PWC),))
        return e.pwc_forward(img1s, img2s).clone()


# ------------------------------------------------------------------------------------------ losses / train ----
preprocess_flow_batch = ops.preprocess_flow_batch
charbonnier_loss = ops.charbonnier_loss


class AdamOptimizer(object):
    """tf.train.AdamOptimizer(learning_rate, beta1): ONE object serves both train ops (adversarial_learner.py:216), so
    its beta-power accumulators advance on every apply_gradients, whichever network it updates."""

    def __init__(self, learning_rate=1e-4, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.beta1, self.beta2, self.epsilon = learning_rate, beta1, beta2, epsilon
        self.t = 0
        self._slots = {}

    def slots(self, net):
        if net not in self._slots:
            w = variables(net)
            self._slots[net] = (torch.zeros_like(w), torch.zeros_like(w))
        return self._slots[net]


class AdversarialGraph(object):
    """The tensors of build_train_graph (adversarial_learner.py:72-258) for one batch shape: run(img1, img2) executes the
    forward pass and returns the losses{} dictionary of :196-204 whose 'generator' / 'recover' entries are handles that
    train_op() differentiates."""

    class Loss(object):
        def __init__(self, graph, which, name):
            self.graph, self.which, self.name = graph, which, name

        def value(self):
            return self.graph.engine.losses()[self.name]

        def __float__(self):
            return float(self.value())

    def __init__(self, batch_size, img_hw=(192, 384), in_hw=(384, 640), group=None, **flags):
        self._key = dict(batch=batch_size, img_hw=img_hw, in_hw=in_hw, **flags)
        self.group = group
        n_gen, n_rec = W.param_total(W.NET_GEN), W.param_total(W.NET_REC)
        off = (n_gen + 63) // 64 * 64
        self.g_all = torch.zeros(off + n_rec, dtype=torch.float32, device="cuda")  # one all-reduce payload (trainer.py)
        self.grads = {W.NET_GEN: self.g_all[:n_gen], W.NET_REC: self.g_all[off:off + n_rec]}

    @property
    def engine(self):
        k = dict(self._key)
        return _engine(k.pop("batch"), k.pop("img_hw"), k.pop("in_hw"), nets=(W.NET_PWC, W.NET_GEN, W.NET_REC), **k)

    def run(self, img1, img2):
        e = self.engine
        e.forward(img1, img2, 3)
        out = {k: None for k in ops.LOSS_KEYS}
        out["generator"] = AdversarialGraph.Loss(self, GEN, "generator")
        out["recover"] = AdversarialGraph.Loss(self, REC, "recover")
        for k in ops.LOSS_KEYS[2:]:
            out[k] = AdversarialGraph.Loss(self, 0, k)
        return out


def train_op(loss, var_list, optimizer, gradient_clip_value=0.1, can_change=False):
    """Train operation (models/utils/loss_utils.py:12-32).  `loss` is losses['generator'] or losses['recover'] of an
    AdversarialGraph.run(); `var_list` names the network (a scope string, a net id, or the dict / list of its variables).
    Returns (train_operation, clipped_grad_and_vars): calling train_operation() runs compute_gradients (the HIP backward
    pass of that loss), the data-parallel mean over ranks, the clip / escape-noise rule and optimizer.apply_gradients, stage
    by stage through the library's per-stage entry points; clipped_grad_and_vars lists (gradient view, variable view) per
    variable in TF creation order (valid after the operation ran)."""
    if not isinstance(loss, AdversarialGraph.Loss) or loss.which not in (GEN, REC):
        raise ValueError("train_op: loss must be losses['generator'] or losses['recover'] of AdversarialGraph.run()")
    net = W.NET_GEN if loss.which == GEN else W.NET_REC
    if isinstance(var_list, (str, int)):
        vnet = _net_of(var_list, None) if isinstance(var_list, str) else var_list
        if vnet != net:
            raise ValueError("train_op: var_list names another network than the loss trains (adversarial_learner.py:224-234)")
    graph = loss.graph
    w, g = variables(net), graph.grads[net]
    gv = list(zip(W.as_dict(g, net).values(), W.as_dict(w, net).values()))

    def train_operation():
        from .trainer import allreduce_mean_
        e = graph.engine
        w_gen = variables(W.NET_GEN)
        w_rec = variables(W.NET_REC)
        # grads_and_vars = optimizer.compute_gradients(loss, var_list)
        if net == W.NET_GEN:
            check(lib.udet_generator_backward(e._h, w_gen.data_ptr(), g.data_ptr(), e.ws.data_ptr(), e._stream()))
        else:
            check(lib.udet_recover_backward(e._h, w_rec.data_ptr(), g.data_ptr(), e.ws.data_ptr(), e._stream()))
        allreduce_mean_(g, graph.group)
        optimizer.t += 1
        flag = None
        if can_change:  # grad_avg_value < 1e-5 -> abs(randomize(grad))
            flag = torch.empty(2, dtype=torch.float32, device=g.device)
            check(lib.udet_grad_absmean(e._h, net, g.data_ptr(), flag.data_ptr(), e.ws.data_ptr(), e._stream()))
        ops.clip_or_noise_(g, gradient_clip_value, flag, seed=e.cfg.noise_seed, step=optimizer.t)
        m, v = optimizer.slots(net)
        ops.adam_step_(w, g, m, v, optimizer.t, optimizer.lr, optimizer.beta1, optimizer.beta2, optimizer.epsilon)
        mark_updated(net)

    return train_operation, gv
