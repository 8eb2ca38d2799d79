"""oracle.ref -- the reference's OWN code, compiled from /root/reference into oracle/_ref by oracle/Makefile.

TEST INFRASTRUCTURE ONLY.  CPU pieces (kNN = nanoflann, three_nn / three_interpolate) run anywhere the
prebuilt .so travelled to; the HIP pieces (the reference .cu files compiled unchanged by hipcc) need a GPU.
"""
import ctypes
import os

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def available(name):
    return os.path.exists(os.path.join(_DIR, name))


def _load(name):
    path = os.path.join(_DIR, name)
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not built (run `make -C oracle ref` where /root/reference exists)")
    return ctypes.CDLL(path)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_knn = None
_interp = None


def knn_batch(pts, queries, K, omp=False):
    """reference cpp_knn_batch / cpp_knn_batch_omp (knn_.cxx:72-135) -> (B,M,K) int64"""
    global _knn
    if _knn is None:
        _knn = _load("libref_knn.so")
    pts, queries = _f32(pts), _f32(queries)
    b, n, dim = pts.shape
    m = queries.shape[1]
    out = np.zeros((b, m, K), np.int64)
    fn = _knn.ref_knn_batch_omp if omp else _knn.ref_knn_batch
    sz = ctypes.c_size_t
    fn(_p(pts), sz(b), sz(n), sz(dim), _p(queries), sz(m), sz(K), _p(out))
    return out


def knn_batch_distance_pick(pts, nqueries, K, seed):
    """reference cpp_knn_batch_distance_pick (knn_.cxx:136-200) with its time(0) seed pinned to `seed` (the shim overrides
    time() inside libref_knn.so) -> (indices (B,nq,K) int64, queries (B,nq,3))"""
    global _knn
    if _knn is None:
        _knn = _load("libref_knn.so")
    pts = _f32(pts)
    b, n, dim = pts.shape
    out = np.zeros((b, nqueries, K), np.int64)
    q = np.zeros((b, nqueries, dim), np.float32)
    sz = ctypes.c_size_t
    _knn.ref_knn_batch_distance_pick(ctypes.c_long(int(seed)), _p(pts), sz(b), sz(n), sz(dim), _p(q), sz(nqueries), sz(K), _p(out))
    return out, q


def three_nn(xyz1, xyz2):
    """reference threenn_cpu (tf_interpolate.cpp:60-103)"""
    global _interp
    if _interp is None:
        _interp = _load("libref_interp.so")
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    _interp.threenn_cpu(b, n, m, _p(xyz1), _p(xyz2), _p(dist), _p(idx))
    return dist, idx


def three_interpolate(points, idx, weight):
    """reference threeinterpolate_cpu (tf_interpolate.cpp:107-127)"""
    global _interp
    if _interp is None:
        _interp = _load("libref_interp.so")
    points, weight = _f32(points), _f32(weight)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.zeros((b, n, c), np.float32)
    _interp.threeinterpolate_cpu(b, m, c, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def three_interpolate_grad(points, idx, weight, grad_out):
    """reference threeinterpolate_grad_cpu (tf_interpolate.cpp:131-153)"""
    global _interp
    if _interp is None:
        _interp = _load("libref_interp.so")
    points, weight, grad_out = _f32(points), _f32(weight), _f32(grad_out)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    b, m, c = points.shape
    n = idx.shape[1]
    g = np.zeros((b, m, c), np.float32)
    _interp.threeinterpolate_grad_cpu(b, n, c, m, _p(grad_out), _p(idx), _p(weight), _p(g))
    return g


class HipRef:
    """The reference CUDA launchers (tf_sampling_g.cu:194-211, tf_grouping_g.cu:125-141) on torch CUDA tensors.

    ``nofma=True`` loads the -ffp-contract=off build (canonical arithmetic); the default build lets hipcc
    contract as it would for a user of the reference."""

    def __init__(self, nofma=True):
        import torch  # noqa: F401  (device memory only)

        self.lib = _load("libref_tfops_hip_nofma.so" if nofma else "libref_tfops_hip.so")

    @staticmethod
    def _ptr(t):
        return ctypes.c_void_p(t.data_ptr())

    def _sync(self):
        import torch

        torch.cuda.synchronize()
        self.lib.ref_sync()

    def farthest_point_sample(self, npoint, inp):
        import torch

        b, n, _ = inp.shape
        temp = torch.empty((32, n), dtype=torch.float32, device=inp.device)  # tf_sampling.cpp:115
        out = torch.zeros((b, npoint), dtype=torch.int32, device=inp.device)
        torch.cuda.synchronize()
        self.lib.ref_fps(b, n, int(npoint), self._ptr(inp), self._ptr(temp), self._ptr(out))
        self._sync()
        return out

    def gather_point(self, inp, idx):
        import torch

        b, n, _ = inp.shape
        m = idx.shape[1]
        out = torch.zeros((b, m, 3), dtype=torch.float32, device=inp.device)
        torch.cuda.synchronize()
        self.lib.ref_gather_point(b, n, m, self._ptr(inp), self._ptr(idx), self._ptr(out))
        self._sync()
        return out

    def prob_sample(self, inp, inpr):
        import torch

        b, n = inp.shape
        m = inpr.shape[1]
        temp = torch.zeros((b, n), dtype=torch.float32, device=inp.device)
        out = torch.zeros((b, m), dtype=torch.int32, device=inp.device)
        torch.cuda.synchronize()
        self.lib.ref_prob_sample(b, n, m, self._ptr(inp), self._ptr(inpr), self._ptr(temp), self._ptr(out))
        self._sync()
        return out, temp

    def query_ball_point(self, radius, nsample, xyz1, xyz2):
        import torch

        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        idx = torch.zeros((b, m, nsample), dtype=torch.int32, device=xyz1.device)  # zero-hit rows stay 0
        cnt = torch.zeros((b, m), dtype=torch.int32, device=xyz1.device)
        torch.cuda.synchronize()
        self.lib.ref_query_ball_point(b, n, m, ctypes.c_float(radius), int(nsample), self._ptr(xyz1), self._ptr(xyz2),
                                      self._ptr(idx), self._ptr(cnt))
        self._sync()
        return idx, cnt

    def group_point(self, points, idx):
        import torch

        b, n, c = points.shape
        _, m, ns = idx.shape
        out = torch.zeros((b, m, ns, c), dtype=torch.float32, device=points.device)
        torch.cuda.synchronize()
        self.lib.ref_group_point(b, n, c, m, ns, self._ptr(points), self._ptr(idx), self._ptr(out))
        self._sync()
        return out

    def select_top_k(self, k, dist):
        import torch

        b, m, n = dist.shape
        outi = torch.zeros((b, m, n), dtype=torch.int32, device=dist.device)
        out = torch.zeros((b, m, n), dtype=torch.float32, device=dist.device)
        torch.cuda.synchronize()
        self.lib.ref_select_top_k(b, n, m, int(k), self._ptr(dist), self._ptr(outi), self._ptr(out))
        self._sync()
        return outi, out


_gridsub = None


def grid_subsample(points, features=None, classes=None, sampleDl=0.1):
    """reference grid_subsampling (grid_subsampling.cpp:4-106); rows in unordered_map iteration order"""
    global _gridsub
    if _gridsub is None:
        _gridsub = _load("libref_gridsub.so")
    points = _f32(points)
    n = points.shape[0]
    feats = _f32(features) if features is not None else np.zeros((n, 0), np.float32)
    cls = np.ascontiguousarray(classes, dtype=np.int32) if classes is not None else np.zeros((n, 0), np.int32)
    fdim, ldim = feats.shape[1], cls.shape[1]
    op, of, oc = np.zeros((n, 3), np.float32), np.zeros((n, fdim), np.float32), np.zeros((n, ldim), np.int32)
    m = _gridsub.ref_grid_subsample(ctypes.c_long(n), fdim, ldim, _p(points), _p(feats), _p(cls), ctypes.c_float(sampleDl),
                                    _p(op), _p(of), _p(oc))
    out = [op[:m]]
    if features is not None:
        out.append(of[:m])
    if classes is not None:
        out.append(oc[:m])
    return out[0] if len(out) == 1 else tuple(out)
