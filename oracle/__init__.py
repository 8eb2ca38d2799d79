"""oracle -- CPU restatement of the reference's set-abstraction hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
package; nothing under ``pointasnl_amd/`` does (tests/test_boundary.py greps for it).

``oracle.ops``   numpy front-ends of oracle/pasnl_oracle.c (index/byte-exact ops)
``oracle.cells`` numpy fp32/fp64 restatement of the cells, layers, model graphs and losses (pinned to the reference's own
                 Python: tests/test_oracle_cells_pinned.py)
``oracle.cells_torch`` the classification forward with torch-CPU dense layers (bench.py's cpu_baseline leg)
``oracle.ref``   the reference's own C++ sources compiled into oracle/_ref (when built)
``oracle.tf_shim`` numpy stand-in for the TensorFlow symbols the reference imports: runs the reference's own Python
                 (tests/golden/make_golden.py cells|models|losses; needs /root/reference)
``oracle.weights`` seeded values for the reference's TF variables, regenerated on both sides of a fixture
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "pasnl_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "_build/liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/tf_ops"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class ops:
    """numpy in, numpy out; argument order follows the reference's Python wrappers."""

    @staticmethod
    def set_threads(t):
        lib().oracle_set_threads(ctypes.c_int(int(t)))

    @staticmethod
    def farthest_point_sample(npoint, inp):
        inp = _f32(inp)
        b, n, _ = inp.shape
        out = np.zeros((b, npoint), np.int32)
        lib().oracle_fps(b, n, int(npoint), _p(inp), _p(out))
        return out

    @staticmethod
    def gather_point(inp, idx):
        inp, idx = _f32(inp), _i32(idx)
        b, n, _ = inp.shape
        m = idx.shape[1]
        out = np.zeros((b, m, 3), np.float32)
        lib().oracle_gather_point(b, n, m, _p(inp), _p(idx), _p(out))
        return out

    @staticmethod
    def gather_point_grad(inp, idx, out_g):
        inp, idx, out_g = _f32(inp), _i32(idx), _f32(out_g)
        b, n, _ = inp.shape
        m = idx.shape[1]
        g = np.zeros((b, n, 3), np.float32)
        lib().oracle_gather_point_grad(b, n, m, _p(out_g), _p(idx), _p(g))
        return g

    @staticmethod
    def cumsum(inp):
        inp = _f32(inp)
        b, n = inp.shape
        out = np.zeros((b, n), np.float32)
        lib().oracle_cumsum(b, n, _p(inp), _p(out))
        return out

    @staticmethod
    def prob_sample(inp, inpr):
        inp, inpr = _f32(inp), _f32(inpr)
        b, n = inp.shape
        m = inpr.shape[1]
        temp = np.zeros((b, n), np.float32)
        out = np.zeros((b, m), np.int32)
        lib().oracle_prob_sample(b, n, m, _p(inp), _p(inpr), _p(temp), _p(out))
        return out

    @staticmethod
    def query_ball_point(radius, nsample, xyz1, xyz2):
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        idx = np.zeros((b, m, nsample), np.int32)
        cnt = np.zeros((b, m), np.int32)
        lib().oracle_query_ball_point(b, n, m, ctypes.c_float(radius), int(nsample), _p(xyz1), _p(xyz2), _p(idx), _p(cnt))
        return idx, cnt

    @staticmethod
    def group_point(points, idx):
        points, idx = _f32(points), _i32(idx)
        b, n, c = points.shape
        _, m, ns = idx.shape
        out = np.zeros((b, m, ns, c), np.float32)
        lib().oracle_group_point(b, n, c, m, ns, _p(points), _p(idx), _p(out))
        return out

    @staticmethod
    def group_point_grad(points, idx, grad_out):
        points, idx, grad_out = _f32(points), _i32(idx), _f32(grad_out)
        b, n, c = points.shape
        _, m, ns = idx.shape
        g = np.zeros((b, n, c), np.float32)
        lib().oracle_group_point_grad(b, n, c, m, ns, _p(grad_out), _p(idx), _p(g))
        return g

    @staticmethod
    def select_top_k(k, dist):
        dist = _f32(dist)
        b, m, n = dist.shape
        outi = np.zeros((b, m, n), np.int32)
        out = np.zeros((b, m, n), np.float32)
        lib().oracle_select_top_k(b, n, m, int(k), _p(dist), _p(outi), _p(out))
        return outi, out

    @staticmethod
    def knn_point(k, xyz1, xyz2):
        """tf_grouping.py:48-73: (b,m,n) squared distances by broadcasting, then selection sort."""
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        diff = xyz1[:, None, :, :] - xyz2[:, :, None, :]
        sq = diff * diff
        dist = sq[..., 0]
        for c in range(1, sq.shape[-1]):  # tf.reduce_sum over the last axis, left to right
            dist = dist + sq[..., c]
        outi, out = ops.select_top_k(k, dist)
        return out[:, :, :k].copy(), outi[:, :, :k].copy()

    @staticmethod
    def knn_batch(pts, queries, K, omp=False, return_dist=False):
        pts, queries = _f32(pts), _f32(queries)
        b, n, _ = pts.shape
        m = queries.shape[1]
        idx = np.zeros((b, m, K), np.int64)
        d = np.zeros((b, m, K), np.float32)
        lib().oracle_knn(b, n, m, int(K), _p(pts), _p(queries), _p(idx), _p(d))
        return (idx, d) if return_dist else idx

    @staticmethod
    def mt19937(seed, count):
        """`count` successive outputs of std::mt19937(seed) (numpy's MT19937 with legacy seeding is the same generator)"""
        bg = np.random.MT19937()
        bg._legacy_seeding(int(seed))
        return bg.random_raw(int(count)).astype(np.uint32)

    @staticmethod
    def knn_batch_distance_pick(pts, nqueries, K, seed):
        """knn_.cxx:136-200 with an explicit seed -> (indices (B,nq,K) int64, queries (B,nq,3) float32)"""
        pts = _f32(pts)
        b, n, _ = pts.shape
        rnd = np.ascontiguousarray(ops.mt19937(seed, b * nqueries))
        idx = np.zeros((b, nqueries, K), np.int64)
        q = np.zeros((b, nqueries, 3), np.float32)
        lib().oracle_knn_distance_pick(b, n, int(nqueries), int(K), _p(pts), _p(rnd), _p(idx), _p(q))
        return idx, q

    @staticmethod
    def three_nn(xyz1, xyz2):
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        dist = np.zeros((b, n, 3), np.float32)
        idx = np.zeros((b, n, 3), np.int32)
        lib().oracle_three_nn(b, n, m, _p(xyz1), _p(xyz2), _p(dist), _p(idx))
        return dist, idx

    @staticmethod
    def three_interpolate(points, idx, weight):
        points, idx, weight = _f32(points), _i32(idx), _f32(weight)
        b, m, c = points.shape
        n = idx.shape[1]
        out = np.zeros((b, n, c), np.float32)
        lib().oracle_three_interpolate(b, m, c, n, _p(points), _p(idx), _p(weight), _p(out))
        return out

    @staticmethod
    def three_interpolate_grad(points, idx, weight, grad_out):
        points, idx, weight, grad_out = _f32(points), _i32(idx), _f32(weight), _f32(grad_out)
        b, m, c = points.shape
        n = idx.shape[1]
        g = np.zeros((b, m, c), np.float32)
        lib().oracle_three_interpolate_grad(b, n, c, m, _p(grad_out), _p(idx), _p(weight), _p(g))
        return g

    @staticmethod
    def three_weights(dist):
        dist = _f32(dist)
        w = np.zeros_like(dist)
        lib().oracle_three_weights(ctypes.c_long(dist.size // 3), _p(dist), _p(w))
        return w

    @staticmethod
    def grid_subsample(points, features=None, classes=None, sampleDl=0.1):
        """-> (points[, features][, classes]) in ascending voxel-key order (oracle_grid_subsample)"""
        points = _f32(points)
        n = points.shape[0]
        feats = _f32(features) if features is not None else np.zeros((n, 0), np.float32)
        cls = _i32(classes) if classes is not None else np.zeros((n, 0), np.int32)
        fdim, ldim = feats.shape[1], cls.shape[1]
        op, of, oc = np.zeros((n, 3), np.float32), np.zeros((n, fdim), np.float32), np.zeros((n, ldim), np.int32)
        fn = lib().oracle_grid_subsample
        m = fn(ctypes.c_long(n), fdim, ldim, _p(points), _p(feats), _p(cls), ctypes.c_float(sampleDl), _p(op), _p(of), _p(oc))
        out = [op[:m]]
        if features is not None:
            out.append(of[:m])
        if classes is not None:
            out.append(oc[:m])
        return out[0] if len(out) == 1 else tuple(out)



    @staticmethod
    def crop_d2(points, centre):
        """sklearn EuclideanDistance.rdist on the KDTree's float64 copy of the float32 scan (what `search_tree.query` /
        `query_radius` rank by, semantic_kitti_dataset_grid.py:269-271): d = 0; d += t*t per axis, in order, in double."""
        p = _f32(points).astype(np.float64)
        c = _f32(np.asarray(centre).reshape(3)).astype(np.float64)
        d = p - c
        return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]

    @staticmethod
    def knn_crop(points, centre, k):
        """The k nearest points of a scan to one centre: indices in ASCENDING INDEX order + their squared distances (f64).
        A tie at the k-th distance goes to the lowest indices (sklearn: tree visit order; pinned against sklearn in
        tests/test_oracle_crop.py).  k is clamped to [0, n]."""
        d2 = ops.crop_d2(points, centre)
        n = d2.shape[0]
        k = max(0, min(int(k), n))
        order = np.lexsort((np.arange(n), d2))[:k]  # by (d2, index)
        sel = np.sort(order).astype(np.int32)
        return sel, d2[sel]

    @staticmethod
    def radius_crop(points, centre, radius):
        """Every point with d2 <= radius*radius (inclusive, both in double: sklearn query_radius), ascending index."""
        d2 = ops.crop_d2(points, centre)
        sel = np.nonzero(d2 <= np.float64(radius) * np.float64(radius))[0].astype(np.int32)
        return sel, d2[sel]
