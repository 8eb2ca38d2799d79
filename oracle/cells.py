"""oracle.cells -- numpy restatement of the PointASNL cells and of the three model graphs around them.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED: every function below is checked, in fp64, against the
outputs of the REFERENCE'S OWN PYTHON (utils/pointasnl_util.py, utils/pointnet_util.py, utils/tf_util.py,
models/pointasnl_*.py) imported and executed under oracle/tf_shim -- tests/golden/make_golden.py {cells,models,losses}
-> tests/golden/ref_{cells,models,losses}.npz, asserted by tests/test_oracle_cells_pinned.py at 1e-9 (measured
~1e-13).  What remains outside the pin is TensorFlow's own arithmetic inside tf.nn.conv2d / tf.matmul / tf.nn.softmax /
tf.contrib.layers.batch_norm (an un-vendored dependency): the shim implements their documented semantics.
The restatement is evaluated in fp32 like the reference and in fp64 as a cross-check; the tolerance against the HIP
path is 1e-5.

Weights are passed in as ``params``: a mapping  scope -> {"w": (cin,cout), "b": (cout,), and when the layer
has batch norm "gamma","beta","mean","var"}  (utils/tf_util.py:120-185 conv2d = conv, bias, BN, activation;
BN inference = gamma*(x-mean)/sqrt(var+1e-3)+beta, tf_util.py:527-531 with the contrib default epsilon).
"""
import numpy as np

from . import ops

BN_EPS = 1e-3

TRACE = None  # set to a list: every layer-level call below appends (function, scope, inputs, outputs) -- the tests use it to
              # feed the product's layers the oracle's own intermediate tensors (layer-wise comparison without the
              # amplification of an fp32-ulp difference by a later FPS / kNN decision)


def _traced(fn):
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kw):
        out = fn(*args, **kw)
        if TRACE is not None:
            TRACE.append((fn.__name__, args, kw, out))
        return out
    return wrapper


def _layer(x, p, act):
    """tf_util.conv2d / conv1d / fully_connected with a 1x1 kernel: matmul, bias, [BN], [activation]."""
    dt = x.dtype
    y = x @ p["w"].astype(dt) + p["b"].astype(dt)
    if "gamma" in p:
        y = (y - p["mean"].astype(dt)) / np.sqrt(p["var"].astype(dt) + dt.type(BN_EPS)) * p["gamma"].astype(dt) + p["beta"].astype(dt)
    if act == "relu":
        y = np.maximum(y, 0)
    elif act == "sigmoid":
        y = 1 / (1 + np.exp(-y))
    elif act == "leaky_relu":  # tf.nn.leaky_relu, alpha 0.2
        y = np.where(y > 0, y, y * dt.type(0.2))
    return y


def _softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def batched_gather(points, idx):
    """tf.gather_nd with a leading batch index (pointasnl_util.py:43-49, 63-71): points (B,N,C), idx (B,...)"""
    b = points.shape[0]
    bi = np.arange(b).reshape((b,) + (1,) * (idx.ndim - 1))
    return points[bi, idx]


def _join(outer, scope):
    """TF variable scopes nest: every cell re-opens tf.variable_scope(scope) with the SAME scope string it was
    handed (pointasnl_util.py:119,159,182,233), so variables live under layer1/layer1/layer1/conv_kv_ds etc."""
    return scope if not outer else outer + "/" + scope


def sample_weights(new_point, grouped_xyz, mlps, params, scope, outer=""):
    """SampleWeights, pointasnl_util.py:112-156.  new_point (B,P,as,ch), grouped_xyz (B,P,as,3)."""
    scope = _join(outer, scope)
    ch = new_point.shape[-1]
    cb = max(32, ch // 2)  # :121
    normalized_xyz = grouped_xyz - grouped_xyz[:, :, :1, :]  # :122
    x = np.concatenate([normalized_xyz, new_point], axis=-1)  # :123
    kv = _layer(x, params[scope + "/conv_kv_ds"], None)  # :125-129
    q = _layer(x, params[scope + "/conv_query_ds"], None)  # :130-134
    k, v = kv[..., :cb], kv[..., cb:]  # :136-137
    w = q @ np.swapaxes(k, -1, -2)  # :139
    w = w / np.sqrt(x.dtype.type(cb))  # :141
    w = _softmax(w, -1)  # :142
    g = w @ v  # :145
    for i, _c in enumerate(mlps):  # :147-153
        g = _layer(g, params[scope + "/mlp2_%d" % i], "relu" if i < len(mlps) - 1 else None)
    return _softmax(g, 2)  # :154


def adaptive_sampling(group_xyz, group_feature, num_neighbor, params, scope, outer=""):
    """AdaptiveSampling, pointasnl_util.py:158-173."""
    path = _join(outer, scope)
    if num_neighbor == 0:
        return group_xyz[:, :, 0, :], group_feature[:, :, 0, :]
    num_channel = group_feature.shape[-1]
    sxyz = group_xyz[:, :, :num_neighbor, :]
    sfeat = group_feature[:, :, :num_neighbor, :]
    w = sample_weights(sfeat, sxyz, [32, 1 + num_channel], params, scope, outer=path)
    new_xyz = (sxyz * w[..., :1]).sum(axis=2)
    new_feature = (sfeat * w[..., 1:]).sum(axis=2)
    return new_xyz, new_feature


def point_nonlocal_cell(feature, new_point, mlp, params, scope, outer=""):
    """PointNonLocalCell mode='dot', pointasnl_util.py:175-219.  feature (B,N,C), new_point (B,P,C')."""
    scope = _join(outer, scope)
    cb = mlp[0]
    kv = _layer(feature, params[scope + "/conv_kv"], None)  # :187-190
    q = _layer(new_point, params[scope + "/conv_query"], None)  # :191-194
    k, v = kv[..., :cb], kv[..., cb:]  # :196-197
    att = q @ np.swapaxes(k, -1, -2)  # :199
    att = att / np.sqrt(feature.dtype.type(cb))  # :201
    att = _softmax(att, -1)  # :211
    y = att @ v  # :212
    return _layer(y, params[scope + "/conv_back_project"], "relu")  # :213-216 (conv2d default activation)


def nl_attention_core(q, kv, cb):
    """Just the part the fused HIP kernel computes: softmax(q k^T / sqrt(cb)) v."""
    k, v = kv[..., :cb], kv[..., cb:]
    att = q @ np.swapaxes(k, -1, -2) / np.sqrt(q.dtype.type(cb))
    return _softmax(att, -1) @ v


def knn_query(k, support, query):
    """pointasnl_util.py:22-30"""
    return ops.knn_batch(support, query, k, omp=True).astype(np.int32)


@_traced
def set_abstraction(xyz, feature, npoint, nsample, mlp, params, scope, as_neighbor=8, NL=True, knn_idx=None):
    """PointASNLSetAbstraction, pointasnl_util.py:221-292 (use_knn=True, bn=True).  knn_idx: neighbour lists to use instead
    of the oracle's kNN (tests on clouds with equidistant neighbours pass the reference nanoflann's own lists)."""
    dt = feature.dtype
    num_points, num_channel = feature.shape[1:]
    if num_points == npoint:  # :236-240
        new_xyz, new_feature = xyz, feature
    else:
        fps_idx = ops.farthest_point_sample(npoint, xyz.astype(np.float32))
        new_xyz, new_feature = batched_gather(xyz, fps_idx), batched_gather(feature, fps_idx)
    idx = knn_query(nsample, xyz.astype(np.float32), new_xyz.astype(np.float32)) if knn_idx is None else knn_idx  # :242 -> :62
    grouped_xyz = batched_gather(xyz, idx)
    new_point = np.concatenate([grouped_xyz, batched_gather(feature, idx)], axis=-1)  # :71-74
    if num_points != npoint:  # :246-247
        new_xyz, new_feature = adaptive_sampling(grouped_xyz, new_point, as_neighbor, params, scope, outer=scope)
    grouped_xyz = grouped_xyz - new_xyz[:, :, None, :]  # :248
    new_point = np.concatenate([grouped_xyz, new_point], axis=-1)  # :249
    if NL:  # :252-255
        nonlocal_pt = point_nonlocal_cell(feature, new_feature, [max(32, num_channel // 2), mlp[-1]], params, scope,
                                          outer=scope)
    skip = _layer(new_point.max(axis=2), params[scope + "/skip"], "relu")  # :258-261
    for i in range(len(mlp) - 1):  # :264-269
        new_point = _layer(new_point, params[scope + "/conv%d" % i], "relu")
    weight = _layer(grouped_xyz, params[scope + "/weight_net/wconv0"], "relu")  # :272
    new_point = np.swapaxes(new_point, 2, 3) @ weight  # :273-274  (B,P,C',32)
    b, p = new_point.shape[:2]
    # :275-278 conv2d with kernel [1, C'] over a (B,P,C',32) map == one matmul over the flattened (C',32) window
    new_point = _layer(new_point.reshape(b, p, -1), params[scope + "/after_conv"], "relu")
    new_point = new_point + skip  # :282
    if NL:
        new_point = new_point + nonlocal_pt  # :285
    new_point = _layer(new_point, params[scope + "/aggregation"], "relu")  # :288-290
    return new_xyz.astype(dt), new_point


def _three_weights(dist, dtype):
    """pointasnl_util.py:308-311 / pointnet_util.py:212-215 on the op's float32 squared distances, in the graph's dtype"""
    if dtype == np.float32:
        return ops.three_weights(dist)
    d = np.maximum(dist.astype(dtype), 1e-10)
    return (1.0 / d) / (1.0 / d).sum(axis=2, keepdims=True)


def _three_interpolate(points, idx, w):
    """tf_interpolate.cpp:107-127: (p[i1]*w1 + p[i2]*w2) + p[i3]*w3"""
    if points.dtype == np.float32:
        return ops.three_interpolate(points, idx, w)
    g = batched_gather(points, idx)
    return (g[:, :, 0] * w[:, :, 0:1] + g[:, :, 1] * w[:, :, 1:2]) + g[:, :, 2] * w[:, :, 2:3]


@_traced
def decoding_layer(xyz1, xyz2, points1, points2, nsample, mlp, params, scope):
    """PointASNLDecodingLayer, pointasnl_util.py:294-351 (NL=False, use_xyz=True, use_knn=True)."""
    dist, idx = ops.three_nn(xyz1.astype(np.float32), xyz2.astype(np.float32))  # :307
    w = _three_weights(dist, points2.dtype)  # :308-311
    interpolated = _three_interpolate(points2, idx, w)  # :320
    kidx = knn_query(nsample, xyz1.astype(np.float32), xyz1.astype(np.float32))  # :323
    grouped_xyz = batched_gather(xyz1, kidx)
    grouped_feature = np.concatenate([grouped_xyz, batched_gather(interpolated, kidx)], axis=-1)
    grouped_xyz = grouped_xyz - xyz1[:, :, None, :]  # :324
    weight = _layer(grouped_xyz, params[scope + "/decode_weight_net/wconv0"], "relu")  # :326
    new_points = np.swapaxes(grouped_feature, 2, 3) @ weight  # :328-331
    b, p = new_points.shape[:2]
    new_points = _layer(new_points.reshape(b, p, -1), params[scope + "/decode_after_conv"], "relu")  # :333-336
    if points1 is not None:
        new_points = np.concatenate([new_points, points1], axis=-1)  # :338-341
    for i in range(1, len(mlp)):  # :343-348
        new_points = _layer(new_points, params[scope + "/conv_%d" % i], "relu")
    return new_points


@_traced
def fp_module(xyz1, xyz2, points1, points2, mlp, params, scope):
    """pointnet_fp_module, utils/pointnet_util.py:199-229."""
    dist, idx = ops.three_nn(xyz1.astype(np.float32), xyz2.astype(np.float32))
    w = _three_weights(dist, points2.dtype)
    interpolated = _three_interpolate(points2, idx, w)
    x = np.concatenate([interpolated, points1], axis=2) if points1 is not None else interpolated
    for i in range(len(mlp)):
        x = _layer(x, params[scope + "/conv_%d" % i], "relu")
    return x


def _split_input(pc, feature_channel):
    """tf.slice of the input into coordinates and features (pointasnl_sem_seg.py:22-27); feature_channel == 0: both = pc"""
    if feature_channel > 0:
        return pc[:, :, 0:3], pc[:, :, 3:3 + feature_channel]
    return pc, pc


def sem_seg_forward(point_cloud, params, num_class, dtype=np.float32, feature_channel=0, return_l1=False):
    """models/pointasnl_sem_seg.py:18-50."""
    pc = point_cloud.astype(dtype)
    n = pc.shape[1]
    nps = [n // 8, n // 32, n // 128, n // 256]
    l0_xyz, l0_points = _split_input(pc, feature_channel)
    l1_xyz, l1_points = set_abstraction(l0_xyz, l0_points, nps[0], 32, [32, 32, 64], params, "layer1", 8)
    l2_xyz, l2_points = set_abstraction(l1_xyz, l1_points, nps[1], 32, [64, 64, 128], params, "layer2", 4)
    l3_xyz, l3_points = set_abstraction(l2_xyz, l2_points, nps[2], 32, [128, 128, 256], params, "layer3", 0)
    l4_xyz, l4_points = set_abstraction(l3_xyz, l3_points, nps[3], 32, [256, 256, 512], params, "layer4", 0)
    l3_points = decoding_layer(l3_xyz, l4_xyz, l3_points, l4_points, 16, [512, 512], params, "fa_layer1")
    l2_points = decoding_layer(l2_xyz, l3_xyz, l2_points, l3_points, 16, [256, 256], params, "fa_layer2")
    l1_points = decoding_layer(l1_xyz, l2_xyz, l1_points, l2_points, 16, [256, 128], params, "fa_layer3")
    l0_points = decoding_layer(l0_xyz, l1_xyz, l0_points, l1_points, 16, [128, 128, 128], params, "fa_layer4")
    net = _layer(l0_points, params["fc1"], "relu")
    net = _layer(net, params["fc2"], None)
    return (net, l1_xyz) if return_l1 else net


def sem_seg_res_forward(point_cloud, params, num_class, dtype=np.float32, feature_channel=0, return_l1=False):
    """models/pointasnl_sem_seg_res.py:19-68."""
    pc = point_cloud.astype(dtype)
    n = pc.shape[1]
    nps = [n // 8, n // 32, n // 128, n // 256]
    l0_xyz, l0_feat = _split_input(pc, feature_channel)
    _, l0_points = set_abstraction(l0_xyz, l0_feat, n, 32, [16, 16, 32], params, "layer0", 0, NL=False)
    l1_xyz, l1_1 = set_abstraction(l0_xyz, l0_points, nps[0], 32, [32, 32, 64], params, "layer1_1", 8)
    _, l1_2 = set_abstraction(l0_xyz, l0_points, nps[0], 32, [64, 64], params, "layer1_2", 0, NL=False)
    l1_2 = l1_2 + l1_1
    l2_xyz, l2_1 = set_abstraction(l1_xyz, l1_2, nps[1], 32, [64, 64, 128], params, "layer2_1", 4)
    _, l2_2 = set_abstraction(l2_xyz, l2_1, nps[1], 32, [128, 128], params, "layer2_2", 0, NL=False)
    l2_2 = l2_2 + l2_1
    l3_xyz, l3_1 = set_abstraction(l2_xyz, l2_2, nps[2], 32, [128, 128, 256], params, "layer3_1", 0)
    _, l3_2 = set_abstraction(l3_xyz, l3_1, nps[2], 32, [256, 256], params, "layer3_2", 0, NL=False)
    l3_2 = l3_2 + l3_1
    l4_xyz, l4_1 = set_abstraction(l3_xyz, l3_1, nps[3], 32, [256, 256, 512], params, "layer4_1", 0)  # sic :50
    _, l4_2 = set_abstraction(l4_xyz, l4_1, nps[3], 32, [512, 512], params, "layer4_2", 0, NL=False)
    l4_2 = l4_2 + l4_1
    l3_points = fp_module(l3_xyz, l4_xyz, l3_2, l4_2, [512, 512], params, "fa_layer1")
    l2_points = fp_module(l2_xyz, l3_xyz, l2_2, l3_points, [256, 256], params, "fa_layer2")
    l1_points = fp_module(l1_xyz, l2_xyz, l1_2, l2_points, [256, 128], params, "fa_layer3")
    l0_points = fp_module(l0_xyz, l1_xyz, l0_points, l1_points, [128, 128, 128], params, "fa_layer4")
    net = _layer(l0_points, params["fc1"], "leaky_relu")
    net = _layer(net, params["fc0"], None)
    return (net, l1_xyz) if return_l1 else net


@_traced
def sa_group_all(xyz, points, mlp, params, scope):
    """pointnet_sa_module(group_all=True), utils/pointnet_util.py:59-84,109-125: concat xyz, MLP, max."""
    x = np.concatenate([xyz, points], axis=2)
    for i in range(len(mlp)):
        x = _layer(x, params[scope + "/conv%d" % i], "relu")
    return x.max(axis=1)


def cls_forward(point_cloud, params, adaptive_sample=False, dtype=np.float32):
    """models/pointasnl_cls.py:17-52, use_normal=False, inference (dropout = identity, tf_util.py:612-614)."""
    pc = point_cloud.astype(dtype)
    l0_xyz, l0_points = pc, pc
    as_neighbor = [12, 12] if adaptive_sample else [0, 0]
    l1_xyz, l1_points = set_abstraction(l0_xyz, l0_points, 512, 32, [64, 64, 128], params, "layer1", as_neighbor[0])
    l2_xyz, l2_points = set_abstraction(l1_xyz, l1_points, 128, 64, [128, 128, 256], params, "layer2", as_neighbor[1])
    res = sa_group_all(l1_xyz, l1_points, [128, 256, 512], params, "layer3_1")
    top = sa_group_all(l2_xyz, l2_points, [256, 512, 1024], params, "layer3_2")
    net = np.concatenate([top, res], axis=-1)
    net = _layer(net, params["fc1"], "relu")
    net = _layer(net, params["fc2"], "relu")
    net = _layer(net, params["fc3"], None)
    return net, {"l1_xyz": l1_xyz, "l2_xyz": l2_xyz, "l1_points": l1_points, "l2_points": l2_points}


def repulsion_loss(pred, nsample=20, radius=0.07, dtype=np.float64):
    """utils/pointasnl_util.py:361-378: ball query -> group -> centre -> squared distances -> 5 smallest, drop the
    first (the point itself) -> clamp -> mean(radius - dist * exp(-d2/h^2)).  Indices from the C oracle's ball query."""
    idx, _ = ops.query_ball_point(radius, nsample, pred, pred)
    grouped = batched_gather(pred, idx).astype(dtype) - pred[:, :, None, :].astype(dtype)
    d2 = (grouped ** 2).sum(-1)
    d2 = np.sort(d2, axis=-1)[:, :, 1:5]  # tf.nn.top_k(-d2, 5) then [:, :, 1:]
    d2 = np.maximum(1e-12, d2)
    h = 0.03
    return float(np.mean(radius - np.sqrt(d2) * np.exp(-d2 / h ** 2)))


def params_from_tf(variables):
    """{TF variable name: array} (names as the reference graph creates them, e.g. 'layer1/layer1/conv_kv/bn/gamma';
    weights [kh,kw,cin,cout] or [k,cin,cout] or [cin,cout]) -> the scope -> {"w","b","gamma","beta","mean","var"} mapping
    used above.  Kernels are flattened to (kh*kw*cin, cout): a [1,W] VALID kernel over a (B,P,W,C) map is one matmul over
    the row-major (W,C) window (tf_util.py:170-175)."""
    out = {}
    for full, a in variables.items():
        parts = full.split("/")
        a = np.asarray(a)
        if len(parts) >= 2 and parts[-2] == "bn":
            sc = "/".join(parts[:-2])
            key = {"gamma": "gamma", "beta": "beta", "moving_mean": "mean", "moving_variance": "var"}[parts[-1]]
        else:
            sc = "/".join(parts[:-1])
            key = {"weights": "w", "biases": "b"}[parts[-1]]
            if key == "w":
                a = a.reshape(-1, a.shape[-1])
        out.setdefault(sc, {})[key] = a
    return out


def _xent(logits, labels):
    z = logits - logits.max(axis=-1, keepdims=True)
    return np.log(np.exp(z).sum(axis=-1)) - np.take_along_axis(z, labels[..., None].astype(np.int64), axis=-1)[..., 0]


def _l2_all_weights(params, dtype):
    """sum of tf.nn.l2_loss(v) over tf.global_variables() with 'weights' in the name"""
    return sum((p["w"].astype(dtype) ** 2).sum() / 2 for p in params.values() if "w" in p)


def cls_loss(pred, label, l1_xyz, params, uniform_weight=0, weights_decay=1e-4, dtype=np.float64):
    """models/pointasnl_cls.py:55-70.  uniform_weight == 0: the 'uniform' term is the classify loss times zero."""
    classify = _xent(pred.astype(dtype), label).mean()
    uniform = repulsion_loss(l1_xyz, nsample=20, radius=0.07, dtype=dtype) if uniform_weight > 0 else classify
    return classify + uniform_weight * uniform + weights_decay * _l2_all_weights(params, dtype)


def seg_loss(pred, label, l1_xyz, params, weight_decay, smpw=1.0, uniform_weight=0.01, weights_decay=1e-4, radius=0.07,
             dtype=np.float64, decayed=lambda scope: True):
    """models/pointasnl_sem_seg.py:53-68 == pointasnl_sem_seg_res.py:70-85.  tf.losses.sparse_softmax_cross_entropy
    reduces with SUM_BY_NONZERO_WEIGHTS and ADDS ITS RESULT to the 'losses' collection -- the collection tf_util's
    wd * l2_loss(weights) terms live in (tf_util.py:47-48) -- so tf.add_n(tf.get_collection('losses')) (:63) contains the
    classify loss once more: total = 2 * classify + sum(wd * l2) + uniform_weight * uniform + weights_decay * sum(l2).
    `decayed(scope)`: which layers were built with the weight_decay -- all of them in pointasnl_sem_seg.py; in
    pointasnl_sem_seg_res.py the pointnet_fp_module decoders take none (:57-60), so 'fa_layer*' is excluded there."""
    ce = _xent(pred.astype(dtype), label)
    w = np.broadcast_to(np.asarray(smpw, dtype=dtype), ce.shape)
    classify = (ce * w).sum() / np.count_nonzero(w)
    l2 = _l2_all_weights(params, dtype)
    l2_decayed = sum((p["w"].astype(dtype) ** 2).sum() / 2 for sc, p in params.items() if "w" in p and decayed(sc))
    weight_reg = (weight_decay * l2_decayed if weight_decay is not None else 0.0) + classify
    uniform = repulsion_loss(l1_xyz, nsample=20, radius=radius, dtype=dtype)
    return classify + weight_reg + uniform_weight * uniform + weights_decay * l2
