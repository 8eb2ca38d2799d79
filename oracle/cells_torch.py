"""oracle.cells_torch -- the classification forward of oracle/cells.py with the dense layers on torch-CPU (fp32, proper 2-D
GEMMs on all host threads) instead of numpy broadcasting matmuls.  BASELINE ONLY: bench.py's `cpu_baseline` leg times it
(reference kNN with OpenMP + C ports of the GPU-only ops + torch-CPU for the dense / attention part);
tests/test_oracle_cells_pinned.py checks it against oracle/cells.py.  Never imported by the product.

Index ops: kNN = the reference's own knn_.cxx + nanoflann (oracle/_ref/libref_knn.so, omp=True) when that build is present,
else the C port; FPS = the C port (the reference has no CPU kernel for it), OpenMP over the batch.
"""
import time

import numpy as np
import torch

from . import ops, ref

BN_EPS = 1e-3
TIMES = None  # dict: piece -> seconds, filled when set


def _tick(piece, t0):
    if TIMES is not None:
        TIMES[piece] = TIMES.get(piece, 0.0) + time.perf_counter() - t0


def _fold(p):
    w, b = torch.from_numpy(p["w"]), torch.from_numpy(p["b"])
    if "gamma" in p:
        s = torch.from_numpy(p["gamma"]) / torch.sqrt(torch.from_numpy(p["var"]) + BN_EPS)
        w, b = w * s, (b - torch.from_numpy(p["mean"])) * s + torch.from_numpy(p["beta"])
    return w.contiguous(), b.contiguous()


def _layer(x, p, act):
    w, b = _fold(p)
    y = torch.addmm(b, x.reshape(-1, x.shape[-1]), w).reshape(*x.shape[:-1], w.shape[1])
    if act == "relu":
        y = torch.relu_(y)
    return y


def _gather(points, idx):
    b = points.shape[0]
    bi = torch.arange(b).reshape((b,) + (1,) * (idx.ndim - 1))
    return points[bi, idx]


def knn(k, support, query):
    t0 = time.perf_counter()
    fn = ref.knn_batch if ref.available("libref_knn.so") else ops.knn_batch
    out = torch.from_numpy(np.asarray(fn(support.numpy(), query.numpy(), k, omp=True)).astype(np.int64))
    _tick("knn (reference nanoflann, OpenMP over batch)" if ref.available("libref_knn.so") else "knn (C port, OpenMP)", t0)
    return out


def fps(npoint, xyz):
    t0 = time.perf_counter()
    out = torch.from_numpy(ops.farthest_point_sample(npoint, xyz.numpy()).astype(np.int64))
    _tick("fps (C port, OpenMP over batch)", t0)
    return out


def set_abstraction(xyz, feature, npoint, nsample, mlp, params, scope, as_neighbor, NL=True):
    """pointasnl_util.py:221-292, as oracle/cells.py:set_abstraction."""
    num_channel = feature.shape[-1]
    fi = fps(npoint, xyz)
    new_xyz, new_feature = _gather(xyz, fi), _gather(feature, fi)
    idx = knn(nsample, xyz, new_xyz)
    t0 = time.perf_counter()
    grouped_xyz = _gather(xyz, idx)
    new_point = torch.cat([grouped_xyz, _gather(feature, idx)], -1)
    if as_neighbor == 0:
        new_xyz, new_feature = grouped_xyz[:, :, 0], new_point[:, :, 0]
    else:
        s3 = scope + "/" + scope + "/" + scope
        sx, sf = grouped_xyz[:, :, :as_neighbor], new_point[:, :, :as_neighbor]
        ch = sf.shape[-1]
        cb = max(32, ch // 2)
        x = torch.cat([sx - sx[:, :, :1], sf], -1)
        kv, q = _layer(x, params[s3 + "/conv_kv_ds"], None), _layer(x, params[s3 + "/conv_query_ds"], None)
        w = torch.softmax(q @ kv[..., :cb].transpose(-1, -2) / float(np.sqrt(np.float32(cb))), -1)
        g = _layer(_layer(w @ kv[..., cb:], params[s3 + "/mlp2_0"], "relu"), params[s3 + "/mlp2_1"], None)
        w = torch.softmax(g, 2)
        new_xyz, new_feature = (sx * w[..., :1]).sum(2), (sf * w[..., 1:]).sum(2)
    grouped_xyz = grouped_xyz - new_xyz[:, :, None]
    new_point = torch.cat([grouped_xyz, new_point], -1)
    if NL:
        s2 = scope + "/" + scope
        cb = max(32, num_channel // 2)
        kv, q = _layer(feature, params[s2 + "/conv_kv"], None), _layer(new_feature, params[s2 + "/conv_query"], None)
        att = torch.softmax(q @ kv[..., :cb].transpose(-1, -2) / float(np.sqrt(np.float32(cb))), -1)
        nonlocal_pt = _layer(att @ kv[..., cb:], params[s2 + "/conv_back_project"], "relu")
    skip = _layer(new_point.max(dim=2).values, params[scope + "/skip"], "relu")
    for i in range(len(mlp) - 1):
        new_point = _layer(new_point, params[scope + "/conv%d" % i], "relu")
    weight = _layer(grouped_xyz, params[scope + "/weight_net/wconv0"], "relu")
    new_point = new_point.transpose(2, 3) @ weight
    new_point = _layer(new_point.reshape(new_point.shape[0], new_point.shape[1], -1), params[scope + "/after_conv"], "relu") + skip
    if NL:
        new_point = new_point + nonlocal_pt
    out = _layer(new_point, params[scope + "/aggregation"], "relu")
    _tick("dense + attention (torch CPU fp32)", t0)
    return new_xyz.contiguous(), out


def cls_forward(point_cloud, params, adaptive_sample=False):
    """models/pointasnl_cls.py:17-52 at inference -> logits (B,40) float32 numpy"""
    pc = torch.from_numpy(np.ascontiguousarray(point_cloud, dtype=np.float32))
    a = [12, 12] if adaptive_sample else [0, 0]
    with torch.no_grad():
        l1_xyz, l1_points = set_abstraction(pc, pc, 512, 32, [64, 64, 128], params, "layer1", a[0])
        l2_xyz, l2_points = set_abstraction(l1_xyz, l1_points, 128, 64, [128, 128, 256], params, "layer2", a[1])
        t0 = time.perf_counter()
        res, top = torch.cat([l1_xyz, l1_points], 2), torch.cat([l2_xyz, l2_points], 2)
        for i in range(3):
            res, top = _layer(res, params["layer3_1/conv%d" % i], "relu"), _layer(top, params["layer3_2/conv%d" % i], "relu")
        net = torch.cat([top.max(dim=1).values, res.max(dim=1).values], -1)
        net = _layer(_layer(net, params["fc1"], "relu"), params["fc2"], "relu")
        net = _layer(net, params["fc3"], None)
        _tick("dense + attention (torch CPU fp32)", t0)
    return net.numpy()
