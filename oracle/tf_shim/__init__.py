"""oracle.tf_shim -- an eager numpy stand-in for the ~70 TensorFlow-1.x symbols the reference imports, so that the
reference's OWN Python (utils/pointasnl_util.py, utils/pointnet_util.py, utils/tf_util.py, tf_ops/*/tf_*.py,
models/pointasnl_*.py) can be imported and executed unmodified where TensorFlow cannot be installed.

TEST INFRASTRUCTURE ONLY.  It exists to produce reference-derived fixtures (tests/golden/make_golden.py cells):
every slice, axis, scope name, activation default and op order in those fixtures is the reference's code, not a
reading of it; what this file supplies is the documented semantics of the TF ops themselves
(tf.nn.conv2d NHWC/VALID, tf.matmul with batch dims, tf.nn.softmax(axis), tf.gather_nd, tf.contrib.layers.batch_norm
at inference = gamma * (x - moving_mean) / sqrt(moving_variance + 0.001) + beta, tf.losses.* reductions, ...).

    from oracle import tf_shim
    with tf_shim.session(seed=7, dtype=np.float64) as tfs:     # installs sys.modules['tensorflow'] + the op libraries
        import pointasnl_util                                   # the REFERENCE module, from /root/reference
        out = pointasnl_util.PointNonLocalCell(tfs.constant(x), ...)
        tfs.variables                                           # {'scope/conv_kv/weights': array, ...} in creation order

The custom-op libraries the reference loads with tf.load_op_library (tf_sampling_so.so, tf_grouping_so.so,
tf_interpolate_so.so) and its Cython kNN module are backed by oracle/_ref (the reference's own C++ compiled where it
lies: nanoflann kNN, threenn_cpu, threeinterpolate_cpu) and, for the ops the reference only has as CUDA kernels,
by oracle/pasnl_oracle.c (itself pinned to those kernels compiled by hipcc: tests/golden/ref_tfops_hip.npz).

Everything is evaluated in ONE float type per session (`dtype`): tf.float32 resolves to it, so the same reference
code yields the fp32 result and its fp64 cross-check.  Index-producing ops always see float32 coordinates, like
the reference's kernels.
"""
import contextlib
import os
import sys
import types

import numpy as np

from .. import weights as _weights

REFERENCE = os.environ.get("PASNL_REFERENCE", "/root/reference")


# ------------------------------------------------------------------------------------------------ tensors

class Dimension(int):
    """tf.Dimension: an int with .value (the reference mixes `shape[i].value` and bare dimensions in arithmetic)."""

    @property
    def value(self):
        return int(self)


class TensorShape(list):
    def as_list(self):
        return [int(d) for d in self]

    @property
    def dims(self):
        return list(self)

    def with_rank(self, rank):
        assert len(self) == rank
        return self


class Tensor(np.ndarray):
    """ndarray + get_shape()/set_shape().  Augmented assignments build NEW tensors, as in TF (`grouped_xyz -= ...`
    at pointasnl_util.py:248 must not write through the views handed out by AdaptiveSampling's slices)."""

    name = None

    def get_shape(self):
        return TensorShape(Dimension(d) for d in self.shape)

    def set_shape(self, shape):
        assert tuple(int(s) for s in shape) == self.shape, (shape, self.shape)

    def __isub__(self, other):
        return self - other

    def __iadd__(self, other):
        return self + other

    def __imul__(self, other):
        return self * other

    def __itruediv__(self, other):
        return self / other


def _t(x):
    return np.asarray(x).view(Tensor)


class DType:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return f"tf.{self.name}"


float32, float16, float64 = DType("float32"), DType("float16"), DType("float64")
int32, int64, bool_ = DType("int32"), DType("int64"), DType("bool")


class _State:
    def __init__(self):
        self.reset(0, np.float32)

    def reset(self, seed, dtype):
        self.seed = seed
        self.float = np.dtype(dtype)
        self.scopes = []
        self.variables = {}      # full name -> Tensor, creation order
        self.collections = {}
        self.py_func_calls = 0


S = _State()


def _np_dtype(dt):
    if dt is None:
        return None
    if isinstance(dt, DType):
        if dt.name in ("float32", "float16", "float64"):
            return S.float
        return np.dtype({"int32": np.int32, "int64": np.int64, "bool": np.bool_}[dt.name])
    return np.dtype(dt)


def _axis(axis):
    if axis is None:
        return None
    if isinstance(axis, (list, tuple)):
        return tuple(int(a) for a in axis)
    return int(axis)


# ------------------------------------------------------------------------------------------------ graph plumbing

class _Scope:
    def __init__(self, name):
        self.name = name


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, reuse=None):
    name = name_or_scope.name.split("/")[-1] if isinstance(name_or_scope, _Scope) else str(name_or_scope)
    S.scopes.append(name)
    try:
        yield _Scope("/".join(S.scopes))
    finally:
        S.scopes.pop()


def get_variable_scope():
    return _Scope("/".join(S.scopes))


@contextlib.contextmanager
def device(name):
    yield


@contextlib.contextmanager
def control_dependencies(deps):
    yield


def constant_initializer(value=0.0, dtype=None):
    return ("constant", float(value))


def truncated_normal_initializer(mean=0.0, stddev=1.0, seed=None, dtype=None):
    return ("truncated_normal", float(stddev))


def _xavier_initializer(uniform=True, seed=None, dtype=None):
    return ("xavier", None)


_INIT_OF_LEAF = {"weights": ("xavier", "truncated_normal"), "biases": ("constant",), "beta": ("constant",),
                 "gamma": ("constant",), "moving_mean": ("constant",), "moving_variance": ("constant",)}


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, **kw):
    full = "/".join(S.scopes + [name])
    if full in S.variables:
        raise ValueError(f"Variable {full} already exists, disallowed. Did you mean to set reuse=True?")
    assert initializer is not None and initializer[0] in _INIT_OF_LEAF[name], (full, initializer)
    shape = [int(s) for s in shape]
    v = _t(_weights.make(S.seed, full, shape).astype(S.float))
    v.name = full + ":0"
    S.variables[full] = v
    return v


def global_variables():
    return list(S.variables.values())


def add_to_collection(name, value):
    S.collections.setdefault(name, []).append(value)


def get_collection(name, scope=None):
    return list(S.collections.get(name, []))


def placeholder(dtype, shape=None, name=None):
    raise NotImplementedError("the shim is eager: pass tensors, not placeholders")


def cond(pred, true_fn=None, false_fn=None, **kw):
    return true_fn() if bool(np.asarray(pred)) else false_fn()


def no_op(*a, **k):
    return None


def py_func(func, inp, Tout, stateful=True, name=None):
    S.py_func_calls += 1
    out = func(*[np.asarray(i) for i in inp])
    return _t(np.asarray(out).astype(_np_dtype(Tout)))


def RegisterGradient(op_type):
    return lambda fn: fn


# ------------------------------------------------------------------------------------------------ array ops

def constant(value, dtype=None, shape=None, name=None):
    a = np.asarray(value)
    dt = _np_dtype(dtype)
    if dt is None and a.dtype == np.float64 and not isinstance(value, np.ndarray):
        dt = S.float  # python floats are tf.float32 constants
    a = a.astype(dt) if dt is not None else a
    return _t(a.reshape(shape) if shape is not None else a)


def convert_to_tensor(value, dtype=None, name=None):
    return constant(value, dtype)


def zeros(shape, dtype=float32, name=None):
    return _t(np.zeros([int(s) for s in shape], _np_dtype(dtype)))


def identity(x, name=None):
    return _t(np.array(x))


def range(*args, **kw):  # noqa: A001  (tf.range)
    return _t(np.arange(*[int(a) for a in args], dtype=np.int32))


def reshape(tensor, shape, name=None):
    return _t(np.reshape(np.asarray(tensor), [int(s) for s in shape]))


def tile(input, multiples, name=None):  # noqa: A002
    a = np.asarray(input)
    assert len(multiples) == a.ndim, "tf.tile: multiples must have one entry per dimension"
    return _t(np.tile(a, [int(m) for m in multiples]))


def concat(values, axis, name=None):
    return _t(np.concatenate([np.asarray(v) for v in values], axis=int(axis)))


def expand_dims(input, axis=None, name=None, dim=None):  # noqa: A002
    return _t(np.expand_dims(np.asarray(input), int(axis if axis is not None else dim)))


def squeeze(input, axis=None, name=None, squeeze_dims=None):  # noqa: A002
    axis = axis if axis is not None else squeeze_dims
    a = np.asarray(input)
    if axis is not None:
        for ax in (_axis(axis) if isinstance(_axis(axis), tuple) else (_axis(axis),)):
            assert a.shape[ax] == 1, f"tf.squeeze: dimension {ax} of {a.shape} is not 1"
    return _t(np.squeeze(a, axis=_axis(axis)))


def transpose(a, perm=None, name=None):
    return _t(np.transpose(np.asarray(a), perm))


def slice(input_, begin, size, name=None):  # noqa: A001
    a = np.asarray(input_)
    idx = tuple(np.s_[int(b):(None if int(s) == -1 else int(b) + int(s))] for b, s in zip(begin, size))
    return _t(a[idx])


def gather_nd(params, indices, name=None):
    p, i = np.asarray(params), np.asarray(indices)
    return _t(p[tuple(i[..., d] for d in np.arange(i.shape[-1]))])


def cast(x, dtype, name=None):
    return _t(np.asarray(x).astype(_np_dtype(dtype)))


def _binary(fn):
    def op(x, y, name=None):
        return _t(fn(np.asarray(x), np.asarray(y)))
    return op


add, subtract, multiply, divide = _binary(np.add), _binary(np.subtract), _binary(np.multiply), _binary(np.divide)
maximum, minimum = _binary(np.maximum), _binary(np.minimum)


def _unary(fn):
    def op(x, name=None):
        return _t(fn(np.asarray(x)))
    return op


sqrt, exp, log, square, negative, abs = _unary(np.sqrt), _unary(np.exp), _unary(np.log), _unary(np.square), _unary(np.negative), _unary(np.abs)  # noqa: A001


def add_n(inputs, name=None):
    inputs = list(inputs)
    if not inputs:
        raise ValueError("inputs must be a list of at least one Tensor/IndexedSlices with the same dtype and shape")
    out = np.asarray(inputs[0])
    for t in inputs[1:]:
        out = out + np.asarray(t)
    return _t(out)


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    a, b = np.asarray(a), np.asarray(b)
    if transpose_a:
        a = np.swapaxes(a, -1, -2)
    if transpose_b:
        b = np.swapaxes(b, -1, -2)
    assert a.shape[:-2] == b.shape[:-2], "tf.matmul does not broadcast batch dimensions"
    return _t(np.matmul(a, b))


def _reduce(fn):
    def op(input_tensor, axis=None, keepdims=None, name=None, reduction_indices=None, keep_dims=None):
        keep = bool(keepdims) or bool(keep_dims)
        axis = axis if axis is not None else reduction_indices
        return _t(fn(np.asarray(input_tensor), axis=_axis(axis), keepdims=keep))
    return op


reduce_max, reduce_sum, reduce_mean, reduce_min = _reduce(np.max), _reduce(np.sum), _reduce(np.mean), _reduce(np.min)


def norm(tensor, ord="euclidean", axis=None, keepdims=None, name=None, keep_dims=None):  # noqa: A002
    assert ord in ("euclidean", 2)
    return _t(np.sqrt(np.sum(np.square(np.asarray(tensor)), axis=_axis(axis), keepdims=bool(keepdims) or bool(keep_dims))))


# ------------------------------------------------------------------------------------------------ tf.nn

nn = types.ModuleType("tensorflow.nn")


def _conv2d(input, filter, strides, padding, use_cudnn_on_gpu=True, data_format="NHWC", name=None):  # noqa: A002
    """NHWC, stride 1.  VALID: out[b,h,w,:] = sum_{i,j} x[b,h+i,w+j,:] @ k[i,j]  (cross-correlation, as TF)."""
    x, k = np.asarray(input), np.asarray(filter)
    assert data_format == "NHWC" and list(strides) == [1, 1, 1, 1], (data_format, strides)
    kh, kw, cin, cout = k.shape
    assert x.shape[-1] == cin
    assert padding == "VALID" or (kh == 1 and kw == 1), "SAME padding only for 1x1 kernels"
    b, h, w, _ = x.shape
    oh, ow = h - kh + 1, w - kw + 1
    out = np.zeros((b, oh, ow, cout), x.dtype)
    for i in np.arange(kh):
        for j in np.arange(kw):
            out += x[:, i:i + oh, j:j + ow, :] @ k[i, j]
    return _t(out)


def _conv1d(value, filters, stride, padding, use_cudnn_on_gpu=None, data_format=None, name=None):
    x, k = np.asarray(value), np.asarray(filters)
    assert data_format in (None, "NHWC", "NWC") and int(stride) == 1
    kw, cin, cout = k.shape
    assert padding == "VALID" or kw == 1
    ow = x.shape[1] - kw + 1
    out = np.zeros((x.shape[0], ow, cout), x.dtype)
    for j in np.arange(kw):
        out += x[:, j:j + ow, :] @ k[j]
    return _t(out)


def _bias_add(value, bias, data_format=None, name=None):
    assert data_format in (None, "NHWC")
    return _t(np.asarray(value) + np.asarray(bias))


def _softmax(logits, axis=None, name=None, dim=None):
    axis = -1 if axis is None and dim is None else int(axis if axis is not None else dim)
    x = np.asarray(logits)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return _t(e / e.sum(axis=axis, keepdims=True))


def _leaky_relu(features, alpha=0.2, name=None):
    x = np.asarray(features)
    return _t(np.maximum(x * x.dtype.type(alpha), x))


def _top_k(input, k=1, sorted=True, name=None):  # noqa: A002
    x = np.asarray(input)
    order = np.argsort(-x, axis=-1, kind="stable")[..., :int(k)]  # descending, lower index first among equals
    return _t(np.take_along_axis(x, order, axis=-1)), _t(order.astype(np.int32))


def _xent(labels=None, logits=None, name=None, _sentinel=None):
    z = np.asarray(logits)
    z = z - z.max(axis=-1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=-1))
    return _t(lse - np.take_along_axis(z, np.asarray(labels)[..., None].astype(np.int64), axis=-1)[..., 0])


def _dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
    raise NotImplementedError("dropout is only reachable with is_training=True")


nn.conv2d, nn.conv1d, nn.bias_add, nn.softmax, nn.leaky_relu, nn.top_k = _conv2d, _conv1d, _bias_add, _softmax, _leaky_relu, _top_k
nn.relu = _unary(lambda x: np.maximum(x, 0))
nn.sigmoid = _unary(lambda x: 1 / (1 + np.exp(-x)))
nn.l2_loss = lambda t, name=None: _t(np.sum(np.square(np.asarray(t))) / 2)  # noqa: E731
nn.sparse_softmax_cross_entropy_with_logits = _xent
nn.dropout = _dropout


# ------------------------------------------------------------------------------------------------ tf.contrib.layers

def _batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, activation_fn=None, updates_collections=None,
                is_training=True, reuse=None, scope=None, data_format="NHWC", **kw):
    """tf.contrib.layers.batch_norm at inference: variables <scope>/{beta,gamma,moving_mean,moving_variance} over the
    last axis, y = (x - moving_mean) * gamma / sqrt(moving_variance + epsilon) + beta, epsilon default 0.001."""
    if bool(np.asarray(is_training)):
        raise NotImplementedError("the shim evaluates the inference graph (is_training=False)")
    assert data_format == "NHWC" and activation_fn is None
    x = np.asarray(inputs)
    c = [x.shape[-1]]
    with variable_scope(scope or "BatchNorm"):
        beta = get_variable("beta", c, initializer=constant_initializer(0.0)) if center else 0
        gamma = get_variable("gamma", c, initializer=constant_initializer(1.0)) if scale else 1
        mean = get_variable("moving_mean", c, initializer=constant_initializer(0.0), trainable=False)
        var = get_variable("moving_variance", c, initializer=constant_initializer(1.0), trainable=False)
    inv = np.asarray(gamma) / np.sqrt(np.asarray(var) + x.dtype.type(epsilon))
    return _t((x - np.asarray(mean)) * inv + np.asarray(beta))


contrib = types.ModuleType("tensorflow.contrib")
contrib.layers = types.ModuleType("tensorflow.contrib.layers")
contrib.layers.xavier_initializer = _xavier_initializer
contrib.layers.batch_norm = _batch_norm


# ------------------------------------------------------------------------------------------------ tf.losses / summary

losses = types.ModuleType("tensorflow.losses")


def _sparse_softmax_cross_entropy(labels, logits, weights=1.0, scope=None, loss_collection="losses", reduction=None):
    """tf.losses.sparse_softmax_cross_entropy: Reduction.SUM_BY_NONZERO_WEIGHTS = sum(loss * w) / #(w != 0), and the
    result is ADDED TO tf.GraphKeys.LOSSES ('losses') -- the same collection tf_util's weight decays go to."""
    ce = np.asarray(_xent(labels=labels, logits=logits))
    w = np.broadcast_to(np.asarray(weights).astype(ce.dtype), ce.shape)
    present = np.count_nonzero(w)
    loss = _t(np.sum(ce * w) / ce.dtype.type(present) if present else ce.dtype.type(0))
    if loss_collection:
        add_to_collection(loss_collection, loss)
    return loss


losses.sparse_softmax_cross_entropy = _sparse_softmax_cross_entropy
summary = types.ModuleType("tensorflow.summary")
summary.scalar = summary.histogram = lambda *a, **k: None


# ------------------------------------------------------------------------------------------------ custom-op libraries

def _f32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _bgather(points, idx):
    p, i = np.asarray(points), np.asarray(idx)
    bi = np.arange(p.shape[0]).reshape((-1,) + (1,) * (i.ndim - 1))
    return p[bi, i]


def _ref_or_oracle(name, so):
    from .. import ops, ref
    return getattr(ref, name) if ref.available(so) else getattr(ops, name)


def _sampling_library():
    from .. import ops
    m = types.SimpleNamespace()
    m.farthest_point_sample = lambda inp, npoint: _t(ops.farthest_point_sample(int(npoint), _f32(inp)))
    m.gather_point = lambda inp, idx: _t(_bgather(inp, idx))
    m.gather_point_grad = lambda inp, idx, out_g: _t(ops.gather_point_grad(_f32(inp), idx, _f32(out_g)))
    m.prob_sample = lambda inp, inpr: _t(ops.prob_sample(_f32(inp), _f32(inpr)))
    return m


def _grouping_library():
    from .. import ops
    m = types.SimpleNamespace()

    def query_ball_point(xyz1, xyz2, radius, nsample):
        idx, cnt = ops.query_ball_point(float(radius), int(nsample), _f32(xyz1), _f32(xyz2))
        return _t(idx), _t(cnt)

    def selection_sort(dist, k):
        outi, out = ops.select_top_k(int(k), _f32(dist))
        return _t(outi), _t(out.astype(S.float))

    m.query_ball_point, m.selection_sort = query_ball_point, selection_sort
    m.group_point = lambda points, idx: _t(_bgather(points, idx))
    m.group_point_grad = lambda points, idx, g: _t(ops.group_point_grad(_f32(points), idx, _f32(g)))
    return m


def _interpolate_library():
    m = types.SimpleNamespace()

    def three_nn(xyz1, xyz2):
        dist, idx = _ref_or_oracle("three_nn", "libref_interp.so")(_f32(xyz1), _f32(xyz2))
        return _t(dist.astype(S.float)), _t(idx)

    def three_interpolate(points, idx, weight):
        p, w = np.asarray(points), np.asarray(weight)
        if p.dtype == np.float32:
            return _t(_ref_or_oracle("three_interpolate", "libref_interp.so")(p, idx, _f32(w)))
        g = _bgather(p, idx)  # (b,n,3,c); tf_interpolate.cpp:107-127: (p1*w1 + p2*w2) + p3*w3
        return _t((g[:, :, 0] * w[:, :, 0:1] + g[:, :, 1] * w[:, :, 1:2]) + g[:, :, 2] * w[:, :, 2:3])

    m.three_nn, m.three_interpolate = three_nn, three_interpolate
    m.three_interpolate_grad = lambda p, i, w, g: _t(_ref_or_oracle("three_interpolate_grad", "libref_interp.so")(_f32(p), i, _f32(w), _f32(g)))
    return m


_LIBRARIES = {"tf_sampling_so.so": _sampling_library, "tf_grouping_so.so": _grouping_library,
              "tf_interpolate_so.so": _interpolate_library}


def load_op_library(path):
    return _LIBRARIES[os.path.basename(path)]()


def _knn_module():
    """nearest_neighbors.lib.python.nearest_neighbors (the reference's Cython binding, knn.pyx:71-109) on the reference's
    own knn_.cxx + nanoflann (oracle/_ref/libref_knn.so); the C restatement where that build is absent."""
    m = types.ModuleType("nearest_neighbors.lib.python.nearest_neighbors")
    fn = _ref_or_oracle("knn_batch", "libref_knn.so")
    m.knn_batch = lambda pts, queries, K, omp=False: np.asarray(fn(_f32(pts), _f32(queries), int(K), omp=bool(omp))).astype(np.int64)
    return m


# ------------------------------------------------------------------------------------------------ install / session

_ops_mod = types.ModuleType("tensorflow.python.framework.ops")
_ops_mod.NoGradient = _ops_mod.NotDifferentiable = lambda op_type: None
_ops_mod.RegisterGradient = RegisterGradient

_REF_MODULES = ("tensorflow", "tf_util", "pointasnl_util", "pointnet_util", "tf_sampling", "tf_grouping", "tf_interpolate",
                "nearest_neighbors", "pointasnl_cls", "pointasnl_sem_seg", "pointasnl_sem_seg_res")
_REF_PATHS = ("utils", "models", "tf_ops/sampling", "tf_ops/grouping", "tf_ops/3d_interpolation")


def install():
    """Register this module as `tensorflow` (+ submodules the reference imports) and the kNN binding; put the
    reference's directories on sys.path the way its own files do."""
    if not os.path.isdir(REFERENCE):
        raise FileNotFoundError(f"{REFERENCE}: the reference tree is needed to run the reference's Python")
    me = sys.modules[__name__]
    py = types.ModuleType("tensorflow.python")
    fw = types.ModuleType("tensorflow.python.framework")
    py.framework, fw.ops = fw, _ops_mod
    me.python = py
    mods = {"tensorflow": me, "tensorflow.nn": nn, "tensorflow.contrib": contrib, "tensorflow.contrib.layers": contrib.layers,
            "tensorflow.losses": losses, "tensorflow.summary": summary, "tensorflow.python": py,
            "tensorflow.python.framework": fw, "tensorflow.python.framework.ops": _ops_mod}
    knn = _knn_module()
    pk = [types.ModuleType(n) for n in ("nearest_neighbors", "nearest_neighbors.lib", "nearest_neighbors.lib.python")]
    pk[0].lib, pk[1].python, pk[2].nearest_neighbors = pk[1], pk[2], knn
    mods.update({p.__name__: p for p in pk})
    mods[knn.__name__] = knn
    sys.modules.update(mods)
    for rel in _REF_PATHS:
        p = os.path.join(REFERENCE, rel)
        if p not in sys.path:
            sys.path.insert(0, p)


def uninstall():
    for name in list(sys.modules):
        if name.split(".")[0] in _REF_MODULES:
            del sys.modules[name]
    for rel in _REF_PATHS:
        p = os.path.join(REFERENCE, rel)
        while p in sys.path:
            sys.path.remove(p)


class _Session:
    constant = staticmethod(constant)

    @property
    def variables(self):
        return {k: np.asarray(v) for k, v in S.variables.items()}

    @property
    def collections(self):
        return S.collections

    def reset_graph(self):
        """tf.reset_default_graph(): forget variables and collections (a new model in the same session)."""
        S.reset(S.seed, S.float)


@contextlib.contextmanager
def session(seed=0, dtype=np.float32):
    """Install, evaluate in `dtype` with variables drawn from oracle.weights.make(seed, name, shape), uninstall."""
    S.reset(seed, dtype)
    install()
    try:
        yield _Session()
    finally:
        uninstall()
        S.reset(0, np.float32)
