"""A minimal eager stand-in for the TensorFlow-1.13 Python API, backed by torch-CPU tensors (TEST INFRASTRUCTURE ONLY).

Purpose: TensorFlow 1.13.1 cannot be installed here, so the reference's graph code cannot run as shipped.  With this
module registered as ``tensorflow`` (``install()``), the reference's OWN Python -- ``models/nets.py``,
``models/PWCNet/{model_pwcnet,core_warp,core_costvol}.py``, ``models/utils/{loss_utils,flow_utils,general_utils}.py`` and
``AdversarialLearner.build_train_graph`` -- executes unmodified from /root/reference, op by op, on torch tensors:
``oracle/make_golden.py`` uses that to produce the fixtures under ``tests/golden/``.  What this pins is the reference's
composition (wiring, variable names and shapes, concat orders, strides, loss algebra, clip / noise rule, the gather
arithmetic of the warp, the slicing of the cost volume); what it cannot pin are the TF C++ kernels behind the handful of
primitives below, which are restated here from the TF-1.13 documentation / kernel sources, independently of
``oracle/oracle_torch.py``:

  * conv2d / conv2d_transpose with 'SAME' padding  (tensorflow/core/framework/common_shape_fns.cc GetWindowedOutputSize:
    out = ceil(in/stride), pad_total = max((out-1)*stride + (k-1)*dil + 1 - in, 0), pad_before = pad_total // 2)
  * resize_bilinear / resize_nearest_neighbor, align_corners False ("legacy": src = dst * in/out, no half-pixel
    centres) and True (src = dst * (in-1)/(out-1))            (tensorflow/core/kernels/resize_bilinear_op.cc, image_resizer_state.h)
  * batch_normalization in inference mode with the freshly initialised moving statistics (mean 0, variance 1, eps 1e-3)

Only what the files above call is implemented; anything else raises NotImplementedError when called.
Gradients (``optimizer.compute_gradients``) are torch autograd through the same eager ops.
"""
from __future__ import annotations

import contextlib
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------------ state


class _State:
    def __init__(self):
        self.provider = None          # callable(name, shape) -> torch tensor (leaf, may require grad)
        self.variables = OrderedDict()  # full TF variable name -> tensor, in creation order
        self.trainable = OrderedDict()
        self.scope = []               # variable-scope stack (strings, verbatim)
        self.default_counts = {}      # (scope name, base) -> next suffix for default-named layers
        self.dtype = torch.float32
        self.rng = torch.Generator().manual_seed(0)
        self.raw_grads = []           # one [(grad, var)] list per optimizer.compute_gradients call
        self.draws = []               # every random draw, in order: (op, value) -- TF's own generator cannot be reproduced


STATE = _State()


def reset(provider, dtype=torch.float32):
    """Start a fresh 'graph': variables are fetched from provider(full_name, shape) the first time they are created."""
    STATE.__init__()
    STATE.provider = provider
    STATE.dtype = dtype


def _scope_name():
    return "/".join(STATE.scope)


def _get_variable(name, shape, trainable=True):
    sc = _scope_name()
    full = sc + "/" + name if sc else name
    if full in STATE.variables:
        v = STATE.variables[full]
        if list(v.shape) != [int(s) for s in shape]:
            raise ValueError("variable %s re-requested with shape %s, has %s" % (full, list(shape), list(v.shape)))
        return v
    v = STATE.provider(full, tuple(int(s) for s in shape))
    if tuple(v.shape) != tuple(int(s) for s in shape):
        raise ValueError("provider returned shape %s for %s %s" % (tuple(v.shape), full, tuple(shape)))
    STATE.variables[full] = v
    if trainable:
        STATE.trainable[full] = v
    return v


class _VarScope:
    def __init__(self, name):
        self.name = name


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, reuse=None, **kw):
    name = name_or_scope.name if isinstance(name_or_scope, _VarScope) else name_or_scope
    if isinstance(name_or_scope, _VarScope):  # re-entering a captured scope: absolute
        saved = STATE.scope
        STATE.scope = [name]
        try:
            yield name_or_scope
        finally:
            STATE.scope = saved
        return
    if name is None:
        name = _unique_default(default_name)
    STATE.scope.append(name)
    try:
        yield _VarScope(_scope_name())
    finally:
        STATE.scope.pop()


def _unique_default(base):
    key = (_scope_name(), base)
    n = STATE.default_counts.get(key, 0)
    STATE.default_counts[key] = n + 1
    return base if n == 0 else "%s_%d" % (base, n)


@contextlib.contextmanager
def name_scope(name, *a, **kw):
    # `with tf.name_scope("MaskNet") as scope` hands the string "MaskNet/" to tf.variable_scope (adversarial_learner.py:99)
    yield (name + "/") if name else ""


# ------------------------------------------------------------------------------------------------ tensors

class _Shape(list):
    def as_list(self):
        return list(self)


def _get_shape(self):
    return _Shape(int(s) for s in self.shape)


torch.Tensor.get_shape = _get_shape  # the reference calls x.get_shape().as_list() / x.get_shape()[i]
torch.Tensor.set_shape = lambda self, shape: None


class FlipTensor(torch.Tensor):
    """torch tensor that accepts numpy/TF style reversed slices (x[:, ::-1, :], data/aug_flips.py:3-9)."""

    def __getitem__(self, idx):
        if isinstance(idx, tuple) and any(isinstance(i, slice) and i.step is not None and i.step < 0 for i in idx):
            dims, clean = [], []
            for d, i in enumerate(idx):
                if isinstance(i, slice) and i.step is not None and i.step < 0:
                    if i != slice(None, None, -1):
                        raise NotImplementedError(i)
                    dims.append(d)
                    clean.append(slice(None))
                else:
                    clean.append(i)
            return torch.flip(torch.Tensor.__getitem__(self, tuple(clean)), dims)
        return torch.Tensor.__getitem__(self, idx)


def as_tf(x):
    """numpy array / tensor -> tensor with TF-style indexing."""
    return torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x).as_subclass(FlipTensor)


def _t(x, like=None):
    if isinstance(x, torch.Tensor):
        return x
    dt = like.dtype if isinstance(like, torch.Tensor) else STATE.dtype
    if isinstance(x, (bool, np.bool_)):
        return torch.tensor(bool(x))
    if isinstance(x, (int, np.integer)) and like is None:
        return torch.tensor(int(x))
    return torch.as_tensor(np.asarray(x), dtype=dt) if not isinstance(x, (int, float)) else torch.tensor(x, dtype=dt)


float32, float64, float16, int32, int64, bool_ = torch.float32, torch.float64, torch.float16, torch.int32, torch.int64, torch.bool


def _map_dtype(d):
    return STATE.dtype if d is torch.float32 else d  # "float32" means the graph's float type (float64 golden runs)


def cast(x, dtype, name=None):
    return _t(x).to(_map_dtype(dtype))


def constant(value, dtype=None, shape=None, name=None):
    if dtype is None:
        dtype = STATE.dtype if isinstance(value, float) or (isinstance(value, np.ndarray) and value.dtype.kind == "f") else None
    t = torch.as_tensor(np.asarray(value))
    if dtype is not None:
        t = t.to(_map_dtype(dtype))
    return t


def convert_to_tensor(x, dtype=None, name=None):
    return _t(x)


def shape(x, name=None):
    return _Shape(int(s) for s in x.shape)


def unstack(x, num=None, axis=0, name=None):
    if isinstance(x, (list, tuple)):
        return list(x)
    return list(torch.unbind(x, dim=axis))


def stack(values, axis=0, name=None):
    return torch.stack([_t(v) for v in values], dim=axis)


def concat(values, axis, name=None):
    return torch.cat(list(values), dim=axis)


def expand_dims(x, axis=None, name=None, dim=None):
    return torch.unsqueeze(x, axis if axis is not None else dim)


def reshape(x, shp, name=None):
    return torch.reshape(x, [int(s) for s in shp])


def identity(x, name=None):
    return x


def ones_like(x, dtype=None, name=None):
    return torch.ones_like(x)


def zeros_like(x, dtype=None, name=None):
    return torch.zeros_like(x)


def add(a, b, name=None):
    return a + b


def divide(a, b, name=None):
    return a / b


def square(x, name=None):
    return x * x


def sqrt(x, name=None):
    return torch.sqrt(x)


def tf_pow(x, y, name=None):
    return torch.pow(x, y)


def tf_abs(x, name=None):
    return torch.abs(x)


def floor(x, name=None):
    return torch.floor(x)


def minimum(a, b, name=None):
    return torch.minimum(_t(a, b if isinstance(b, torch.Tensor) else None), _t(b, a if isinstance(a, torch.Tensor) else None))


def maximum(a, b, name=None):
    return torch.maximum(_t(a, b if isinstance(b, torch.Tensor) else None), _t(b, a if isinstance(a, torch.Tensor) else None))


def tf_range(*args, **kw):
    return torch.arange(*[int(a) for a in args])


def meshgrid(*xs, **kw):
    indexing = kw.get("indexing", "xy")
    return list(torch.meshgrid(*xs, indexing=indexing))


def gather(params, indices, name=None, axis=0):
    return params[indices.long()]


def clip_by_value(x, lo, hi, name=None):
    return torch.clamp(x, min=float(lo), max=float(hi))


def _axes(axis):
    if axis is None:
        return None
    return tuple(int(a) for a in axis) if isinstance(axis, (list, tuple)) else int(axis)


def _reduce(fn, x, axis, keep):
    if isinstance(x, (list, tuple)):
        x = torch.stack([_t(v) for v in x])
    ax = _axes(axis)
    if ax is None:
        return fn(x)
    return fn(x, dim=ax, keepdim=bool(keep))


def reduce_sum(x, axis=None, keepdims=None, name=None, keep_dims=None, reduction_indices=None):
    return _reduce(torch.sum, x, axis if axis is not None else reduction_indices, keepdims or keep_dims)


def reduce_mean(x, axis=None, keepdims=None, name=None, keep_dims=None, reduction_indices=None):
    return _reduce(torch.mean, x, axis if axis is not None else reduction_indices, keepdims or keep_dims)


def moments(x, axes, shift=None, name=None, keep_dims=False):
    # tf.nn.moments: population variance  mean((x - mean)^2)
    m = torch.mean(x, dim=_axes(axes), keepdim=True)
    v = torch.mean((x - m) * (x - m), dim=_axes(axes), keepdim=True)
    if not keep_dims:
        m, v = m.squeeze(), v.squeeze()
    return m, v


def cond(pred, true_fn=None, false_fn=None, name=None, fn1=None, fn2=None):
    t, f = (true_fn or fn1), (false_fn or fn2)
    return t() if bool(pred) else f()


def logical_or(a, b, name=None):
    return torch.logical_or(a, b)


def logical_and(a, b, name=None):
    return torch.logical_and(a, b)


def pad(x, paddings, mode="CONSTANT", name=None, constant_values=0):
    flat = []
    for lo, hi in reversed([tuple(p) for p in paddings]):
        flat += [int(lo), int(hi)]
    return F.pad(x, flat, value=float(constant_values))


def tf_slice(x, begin, size, name=None):
    idx = []
    for b, s, d in zip(begin, size, x.shape):
        b = int(b)
        idx.append(slice(b, d if int(s) == -1 else b + int(s)))
    return x[tuple(idx)]


def random_uniform(shape, minval=0, maxval=None, dtype=None, seed=None, name=None):
    shp = [int(v) for v in shape]
    if dtype in (torch.int32, torch.int64):
        lo, hi = int(minval), int(maxval)
        v = torch.randint(lo, hi, shp, generator=STATE.rng, dtype=dtype)
    else:
        hi = 1.0 if maxval is None else float(maxval)
        v = torch.rand(shp, generator=STATE.rng, dtype=STATE.dtype) * (hi - float(minval)) + float(minval)
    if v.numel() <= 16:
        STATE.draws.append(("random_uniform", v.clone().numpy()))
    return v


def equal(a, b, name=None):
    return torch.eq(_t(a), _t(b))


def random_crop(value, size, seed=None, name=None):
    """tf.random_crop: a window of `size` at a uniformly drawn offset (offset recorded in STATE.draws)."""
    size = [int(v) for v in size]
    off = [int(torch.randint(0, int(d) - sz + 1, [], generator=STATE.rng)) for d, sz in zip(value.shape, size)]
    STATE.draws.append(("random_crop", np.array(off + size, np.int64)))
    return value[tuple(slice(o, o + sz) for o, sz in zip(off, size))]


def central_crop(image, central_fraction):
    """tf.image.central_crop (image_ops_impl.py, TF 1.13): start = int32((size - size*fraction) / 2), extent = size - 2*start."""
    if central_fraction >= 1.0:
        return image
    hd, wd = (0, 1) if image.dim() == 3 else (1, 2)
    h, w = int(image.shape[hd]), int(image.shape[wd])
    y0, x0 = int((h - h * central_fraction) / 2), int((w - w * central_fraction) / 2)
    idx = [slice(None)] * image.dim()
    idx[hd], idx[wd] = slice(y0, h - y0), slice(x0, w - x0)
    return image[tuple(idx)]


def placeholder(dtype, shape=None, name=None):
    return torch.tensor(True)  # only `is_training` exists on the path; it selects the training batch


# ------------------------------------------------------------------------------------------------ nn primitives (restated)

def leaky_relu(x, alpha=0.2, name=None):
    return torch.where(x > 0, x, x * alpha)


def elu(x, name=None):
    return torch.where(x > 0, x, torch.expm1(x))


def softmax(x, axis=-1, name=None, dim=None):
    return torch.softmax(x, dim=axis if dim is None else dim)


def _same_pads(n, k, s, d):
    out = -(-n // s)
    total = max((out - 1) * s + (k - 1) * d + 1 - n, 0)
    return total // 2, total - total // 2


def _conv2d(x, w, stride, dilation, padding):
    """x NHWC, w HWIO, cross-correlation."""
    kh, kw = int(w.shape[0]), int(w.shape[1])
    xn = x.permute(0, 3, 1, 2)
    if str(padding).upper() == "SAME":
        pt, pb = _same_pads(int(x.shape[1]), kh, stride, dilation)
        pl, pr = _same_pads(int(x.shape[2]), kw, stride, dilation)
        xn = F.pad(xn, [pl, pr, pt, pb])
    elif str(padding).upper() != "VALID":
        raise NotImplementedError(padding)
    y = F.conv2d(xn, w.permute(3, 2, 0, 1), None, stride=stride, dilation=dilation)
    return y.permute(0, 2, 3, 1)


def nn_conv2d(input, filter, strides, padding, name=None, **kw):
    if strides[0] != 1 or strides[3] != 1 or strides[1] != strides[2]:
        raise NotImplementedError(strides)
    return _conv2d(input, filter, int(strides[1]), 1, padding)


def bias_add(x, b, name=None):
    return x + b


def _pair(v):
    return (int(v), int(v)) if isinstance(v, (int, np.integer)) else (int(v[0]), int(v[1]))


def layers_conv2d(inputs, filters, kernel_size, strides=(1, 1), padding="valid", data_format="channels_last",
                  dilation_rate=(1, 1), activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None,
                  name=None, reuse=None, trainable=True, **kw):
    kh, kw_ = _pair(kernel_size)
    s, d = _pair(strides), _pair(dilation_rate)
    if s[0] != s[1] or d[0] != d[1]:
        raise NotImplementedError
    with variable_scope(name, default_name="conv2d"):
        w = _get_variable("kernel", (kh, kw_, int(inputs.shape[3]), int(filters)))
        y = _conv2d(inputs, w, s[0], d[0], padding)
        if use_bias:
            y = y + _get_variable("bias", (int(filters),))
    return activation(y) if activation is not None else y


def layers_conv2d_transpose(inputs, filters, kernel_size, strides=(1, 1), padding="valid", activation=None, use_bias=True,
                            kernel_initializer=None, name=None, reuse=None, **kw):
    """tf.layers.conv2d_transpose, 'same': output = stride * input; defined as the gradient of the SAME forward convolution
    (kernel [kh, kw, out_channels, in_channels]) with respect to its input (conv2d_backprop_input)."""
    kh, kw_ = _pair(kernel_size)
    s = _pair(strides)
    if str(padding).lower() != "same" or s[0] != s[1]:
        raise NotImplementedError
    cin = int(inputs.shape[3])
    with variable_scope(name, default_name="conv2d_transpose"):
        w = _get_variable("kernel", (kh, kw_, int(filters), cin))
        H, W = int(inputs.shape[1]) * s[0], int(inputs.shape[2]) * s[1]
        # forward conv being differentiated: [N,H,W,filters] -> [N,H/s,W/s,cin] with SAME pads (pt, pl)
        pt, _ = _same_pads(H, kh, s[0], 1)
        pl, _ = _same_pads(W, kw_, s[1], 1)
        g = inputs.permute(0, 3, 1, 2)                 # gradient w.r.t. the forward conv's output
        wt = w.permute(3, 2, 0, 1)                      # forward conv weight as [out=cin, in=filters, kh, kw]
        full = F.conv_transpose2d(g, wt, None, stride=s)  # size (in-1)*s + k, indexed on the padded forward input
        y = full[:, :, pt:pt + H, pl:pl + W]
        if y.shape[2] != H or y.shape[3] != W:
            y = F.pad(y, [0, W - y.shape[3], 0, H - y.shape[2]])
        y = y.permute(0, 2, 3, 1)
        if use_bias:
            y = y + _get_variable("bias", (int(filters),))
    return activation(y) if activation is not None else y


def layers_batch_normalization(inputs, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, training=False,
                               name=None, reuse=None, **kw):
    if training is not False:
        raise NotImplementedError("batch statistics")
    c = int(inputs.shape[-1])
    with variable_scope(name, default_name="batch_normalization"):
        gamma = _get_variable("gamma", (c,))
        beta = _get_variable("beta", (c,))
        mean = _get_variable("moving_mean", (c,), trainable=False)
        var = _get_variable("moving_variance", (c,), trainable=False)
    return (inputs - mean) / torch.sqrt(var + epsilon) * gamma + beta


def model_variable(name, shape=None, initializer=None, trainable=True, **kw):
    return _get_variable(name, shape, trainable=bool(trainable))


def _resize_coords(n_in, n_out, align_corners, dtype):
    if align_corners and n_out > 1:
        scale = (n_in - 1) / (n_out - 1)
    else:
        scale = n_in / n_out
    # the TF kernel computes the scale and the source coordinate in float32
    return torch.arange(n_out, dtype=torch.float32) * torch.tensor(scale, dtype=torch.float32)


def resize_bilinear(images, size, align_corners=False, name=None):
    oh, ow = int(size[0]), int(size[1])
    n, h, w, c = images.shape
    ys, xs = _resize_coords(h, oh, align_corners, images.dtype), _resize_coords(w, ow, align_corners, images.dtype)
    y0 = torch.floor(ys).long()
    x0 = torch.floor(xs).long()
    y1 = torch.clamp(y0 + 1, max=h - 1)
    x1 = torch.clamp(x0 + 1, max=w - 1)
    ly = (ys - y0.to(torch.float32)).to(images.dtype).view(1, oh, 1, 1)
    lx = (xs - x0.to(torch.float32)).to(images.dtype).view(1, 1, ow, 1)
    top = images[:, y0][:, :, x0] + (images[:, y0][:, :, x1] - images[:, y0][:, :, x0]) * lx
    bot = images[:, y1][:, :, x0] + (images[:, y1][:, :, x1] - images[:, y1][:, :, x0]) * lx
    return top + (bot - top) * ly


def resize_nearest_neighbor(images, size, align_corners=False, name=None):
    oh, ow = int(size[0]), int(size[1])
    n, h, w, c = images.shape
    ys, xs = _resize_coords(h, oh, align_corners, images.dtype), _resize_coords(w, ow, align_corners, images.dtype)
    if align_corners:  # roundf: half away from zero; the coordinates are >= 0
        yi, xi = torch.floor(ys + 0.5), torch.floor(xs + 0.5)
    else:
        yi, xi = torch.floor(ys), torch.floor(xs)
    yi = torch.clamp(yi.long(), max=h - 1)
    xi = torch.clamp(xi.long(), max=w - 1)
    return images[:, yi][:, :, xi]


class ResizeMethod:
    BILINEAR, NEAREST_NEIGHBOR, BICUBIC, AREA = 0, 1, 2, 3


def resize_images(images, size, method=ResizeMethod.BILINEAR, align_corners=False, **kw):
    if images.dim() == 3:  # a single HWC image
        return resize_images(images.unsqueeze(0), size, method, align_corners).squeeze(0)
    size = [int(size[0]), int(size[1])]
    if [int(images.shape[1]), int(images.shape[2])] == size:
        return images  # tf.image.resize_images returns the input when the size already matches
    if method == ResizeMethod.BILINEAR:
        return resize_bilinear(images, size, align_corners)
    if method == ResizeMethod.NEAREST_NEIGHBOR:
        return resize_nearest_neighbor(images, size, align_corners)
    raise NotImplementedError(method)


# ------------------------------------------------------------------------------------------------ training-graph pieces

class GraphKeys:
    TRAINABLE_VARIABLES, UPDATE_OPS, GLOBAL_VARIABLES = "trainable_variables", "update_ops", "variables"


def get_collection(key, scope=None):
    if key == GraphKeys.TRAINABLE_VARIABLES:
        return [v for n, v in STATE.trainable.items() if scope is None or n.startswith(scope)]
    return []


def trainable_names(scope=None):
    return [n for n in STATE.trainable if scope is None or n.startswith(scope)]


class AdamOptimizer:
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, **kw):
        self.hp = dict(lr=learning_rate, beta1=beta1, beta2=beta2, eps=epsilon)
        self.applied = []

    def compute_gradients(self, loss, var_list=None, **kw):
        grads = torch.autograd.grad(loss, list(var_list), retain_graph=True, allow_unused=False)
        STATE.raw_grads.append(list(zip(grads, var_list)))  # before the reference's clip / noise rule
        return list(zip(grads, var_list))

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
        self.applied.append(list(grads_and_vars))
        return None


def Variable(initial_value, name=None, trainable=True, **kw):
    return torch.tensor(initial_value)


def group(*a, **kw):
    return None


def assign(ref, value, **kw):
    return None


# ------------------------------------------------------------------------------------------------ module assembly

class _Missing:
    def __init__(self, path):
        self._path = path

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Missing(self._path + "." + k)

    def __call__(self, *a, **kw):
        raise NotImplementedError("tf1_shim: %s is not on the restated surface" % self._path)


class _Mod(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Missing(self.__name__ + "." + k)


def _mod(name, **attrs):
    m = _Mod(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _initializer(*a, **kw):
    return None  # values always come from the provider


def install():
    """Register the stand-in under the module names the reference imports."""
    nn = _mod("tensorflow.nn", conv2d=nn_conv2d, bias_add=bias_add, leaky_relu=leaky_relu, elu=elu, softmax=softmax,
              moments=moments)
    layers = _mod("tensorflow.layers", conv2d=layers_conv2d, conv2d_transpose=layers_conv2d_transpose,
                  batch_normalization=layers_batch_normalization)
    image = _mod("tensorflow.image", resize_bilinear=resize_bilinear, resize_nearest_neighbor=resize_nearest_neighbor,
                 resize_images=resize_images, ResizeMethod=ResizeMethod, central_crop=central_crop)
    contrib = _mod("tensorflow.contrib",
                   layers=_mod("tensorflow.contrib.layers", xavier_initializer_conv2d=_initializer, xavier_initializer=_initializer),
                   framework=_mod("tensorflow.contrib.framework", model_variable=model_variable))
    keras = _mod("tensorflow.keras", initializers=_mod("tensorflow.keras.initializers", he_normal=_initializer))
    train = _mod("tensorflow.train", AdamOptimizer=AdamOptimizer)
    tf = _mod(
        "tensorflow", nn=nn, layers=layers, image=image, contrib=contrib, keras=keras, train=train,
        float32=float32, float64=float64, half=float16, float16=float16, int32=int32, int64=int64, bool=bool_, uint8=torch.uint8,
        Tensor=torch.Tensor, AUTO_REUSE="auto_reuse", GraphKeys=GraphKeys,
        cast=cast, constant=constant, convert_to_tensor=convert_to_tensor, shape=shape, unstack=unstack, stack=stack,
        concat=concat, expand_dims=expand_dims, reshape=reshape, identity=identity, ones_like=ones_like, zeros_like=zeros_like,
        add=add, divide=divide, square=square, sqrt=sqrt, pow=tf_pow, abs=tf_abs, floor=floor, minimum=minimum, maximum=maximum,
        range=tf_range, meshgrid=meshgrid, gather=gather, clip_by_value=clip_by_value, reduce_sum=reduce_sum,
        reduce_mean=reduce_mean, cond=cond, logical_or=logical_or, logical_and=logical_and, pad=pad, slice=tf_slice,
        random_uniform=random_uniform, random_crop=random_crop, equal=equal, placeholder=placeholder, variable_scope=variable_scope, name_scope=name_scope,
        constant_initializer=_initializer, get_collection=get_collection, Variable=Variable, group=group, assign=assign,
        device=lambda *a, **k: contextlib.nullcontext(),
    )
    fw_ops = _mod("tensorflow.python.framework.ops", name_scope=name_scope, convert_to_tensor=convert_to_tensor)
    fw_const = _mod("tensorflow.python.framework.constant_op", constant=constant)
    fw_dtypes = _mod("tensorflow.python.framework.dtypes", int32=int32, float32=float32, int64=int64)
    array_ops = _mod("tensorflow.python.ops.array_ops", unstack=unstack, shape=shape, expand_dims=expand_dims, reshape=reshape,
                     gather=gather, stack=stack, meshgrid=meshgrid)
    math_ops = _mod("tensorflow.python.ops.math_ops", cast=cast, minimum=minimum, maximum=maximum, floor=floor, range=tf_range)
    framework = _mod("tensorflow.python.framework", ops=fw_ops, constant_op=fw_const, dtypes=fw_dtypes)
    pyops = _mod("tensorflow.python.ops", array_ops=array_ops, math_ops=math_ops)
    python = _mod("tensorflow.python", framework=framework, ops=pyops)
    tf.python = python
    mods = {
        "tensorflow": tf, "tensorflow.python": python, "tensorflow.python.framework": framework,
        "tensorflow.python.framework.ops": fw_ops, "tensorflow.python.framework.constant_op": fw_const,
        "tensorflow.python.framework.dtypes": fw_dtypes, "tensorflow.python.ops": pyops,
        "tensorflow.python.ops.array_ops": array_ops, "tensorflow.python.ops.math_ops": math_ops,
        "tensorflow.contrib": contrib, "tensorflow.nn": nn, "tensorflow.layers": layers, "tensorflow.image": image,
        # imported at module level by reference files on the path, never called there
        "cv2": _mod("cv2"),
        "keras": _mod("keras"), "keras.utils": _mod("keras.utils"),
        "keras.utils.generic_utils": _mod("keras.utils.generic_utils", Progbar=object),
    }
    sys.modules.update(mods)
    return tf
