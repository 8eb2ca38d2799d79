"""nearest_neighbors -- drop-in for the reference's Cython module (utils/nearest_neighbors/knn.pyx:33-148),
imported by the models as ``nearest_neighbors.lib.python.nearest_neighbors`` (pointasnl_util.py:19).

The reference builds a nanoflann KD-tree per cloud on the host (OpenMP over the batch) and is reached through
tf.py_func, i.e. a device->host->device round trip per layer.  Here the search is an exact gfx950 kernel (brute force, or
grid-pruned for large clouds) that returns the K nearest in ascending (squared distance, index) order -- nanoflann's list
whenever a query's distances are distinct -- and, by default (tie_order="reference"), a GPU rebuild of nanoflann's own tree
and search for exactly the queries whose list contains or ends on EQUAL distances (there nanoflann's order is its tree's
visit order): the result is cpp_knn_batch's bit for bit, ties included, at the plain search's price on tie-free clouds.

numpy in -> numpy int64 out like the reference (host buffers cross PCIe); torch CUDA tensors in -> torch CUDA
tensors out (no copies), which is what utils/pointasnl_util.py uses.
"""
import numpy as np
import torch

import ctypes

from pointasnl_amd import _hip

GRID = True  # False: brute-force kernels for every size (A/B, and the reference point of the grid kernel's parity test)


# Tree-depth flag of the KD-tree searches (tie_order "reference" / "nanoflann"): ONE sticky int32 per device that the kernels
# only ever set.  It cannot be read while a HIP graph is being captured (and reading it costs a synchronisation), so eager
# tie_order="nanoflann" calls check it at once and everything else leaves it to check_deferred_flags().
_DEPTH_FLAG = {}
_DEPTH_MSG = ("knn_batch: a KD-tree deeper than 96 levels (pathologically clustered cloud); the affected rows hold valid neighbours in "
              "canonical (distance, index) order instead of nanoflann's order among equal distances")


def _depth_flag(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _DEPTH_FLAG:
        _DEPTH_FLAG[key] = torch.zeros((1,), dtype=torch.int32, device=device)
    return _DEPTH_FLAG[key]


def check_deferred_flags(clear=False):
    """Raise PasnlUnsupported if any KD-tree search since the last cleared check (eager or replayed from a HIP graph) met a tree
    deeper than its stack.  Synchronises with the device.  clear=True resets the flags."""
    bad = any(int(f.item()) != 0 for f in _DEPTH_FLAG.values())
    if clear:
        for f in _DEPTH_FLAG.values():
            f.zero_()
    if bad:
        raise _hip.PasnlUnsupported(_DEPTH_MSG)


def _check_args(pts, queries):
    if pts.dim() != 3 or pts.shape[2] != 3 or queries.dim() != 3 or queries.shape[2] != 3:
        raise ValueError("knn_batch expects (B,N,3) pts and (B,M,3) queries")
    if queries.shape[0] != pts.shape[0]:
        raise ValueError("knn_batch expects the same batch size for pts and queries")


def _knn_tree_dev(pts, queries, K, i64, out=None):
    """EVERY query through the GPU rebuild of nanoflann's tree and search (csrc/knn_tree.hip): the checker of the default path.
    K <= 256 (PASNL_KNN_MAX_K); clouds above 10240 points get their tree from the one-lane build (seconds at 1e5 points)."""
    b, n, _ = pts.shape
    m = queries.shape[1]
    if K > n:
        raise ValueError("knn_batch(tie_order='nanoflann') needs K <= number of points")
    out = _out_buffer(out, (b, m, int(K)), torch.int64 if i64 else torch.int32, pts.device)
    nbytes = int(_hip.lib().pasnl_knn_tree_workspace_bytes(b, n, m, int(K)))
    ws = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=pts.device)
    _hip.launch("pasnl_knn_batch_tree", "knn_batch", b, n, m, int(K), _hip.ptr(pts), _hip.ptr(queries), _hip.ptr(out), int(i64),
                _hip.ptr(ws), ctypes.c_size_t(nbytes))
    flag = ws[:4].view(torch.int32)
    if torch.cuda.is_current_stream_capturing():
        _depth_flag(pts.device).bitwise_or_(flag)  # (captured with the search: the sticky flag is read after the replays)
    elif int(flag.item()) != 0:  # eager: checked at once (a synchronisation: this mode is about exactness, not speed)
        raise _hip.PasnlUnsupported(_DEPTH_MSG)
    return out


def _knn_ref_dev(pts, queries, K, i64, out=None, max_workgroups=None, stats=None):
    """The default: canonical search + nanoflann's tree for the queries whose K-list contains or ends on equal distances
    (pasnl_knn_batch_ref).  No synchronisation; the depth flag is sticky (check_deferred_flags)."""
    b, n, _ = pts.shape
    m = queries.shape[1]
    out = _out_buffer(out, (b, m, int(K)), torch.int64 if i64 else torch.int32, pts.device)
    nbytes = int(_hip.lib().pasnl_knn_batch_ref_workspace_bytes(b, n, m, int(K)))
    ws = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=pts.device)
    _hip.launch("pasnl_knn_batch_ref", "knn_batch", b, n, m, int(K), _hip.ptr(pts), _hip.ptr(queries), _hip.ptr(out), int(i64),
                _hip.ptr(_depth_flag(pts.device)), _hip.ptr(ws), ctypes.c_size_t(nbytes), int(max_workgroups or 0))
    if stats is not None:  # (tests, bench) per cloud: the listed queries (the workspace's first b ints); 256-byte aligned behind them what
        # the tie paths left to the builds (clouds of up to 2048 points with K <= 64 resolve inside one kernel and report 0), and what the
        # on-demand tree (clouds of 8193..10240 points) left to the full builds
        stats.append(ws[:4 * b].view(torch.int32))
        off = (4 * b + 255) // 256 * 256
        stats.append(ws[off:off + 4 * b].view(torch.int32))
        stats.append(ws[off + 4 * b:off + 8 * b].view(torch.int32))
    return out


def _out_buffer(t, shape, dtype, device):
    if t is None:
        return torch.empty(shape, dtype=dtype, device=device)
    if tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device or not t.is_contiguous():
        raise ValueError(f"knn_batch: out must be a contiguous {dtype} tensor of shape {tuple(shape)} on {device}")
    return t


TIE_ORDERS = ("reference", "index", "nanoflann")
REF_MAX_K, REF_MAX_N = 256, 2 ** 31 - 1  # limits of the KD-tree kernels (include/pasnl.h: PASNL_KNN_MAX_K; int32 indices)


def _knn_dev(pts, queries, K, i64, tie_order="reference", out=None, max_workgroups=None, stats=None):
    if tie_order not in TIE_ORDERS:
        raise ValueError("tie_order is 'reference' (cpp_knn_batch's result, ties included; the default), 'index' (canonical "
                         "(distance, index) order) or 'nanoflann' (every query through the rebuilt KD-tree: the checker)")
    _check_args(pts, queries)
    if tie_order == "nanoflann":
        return _knn_tree_dev(pts, queries, K, i64, out)
    b, n, _ = pts.shape
    m = queries.shape[1]
    if tie_order == "reference" and b > 0 and m > 0 and 0 < K <= n:
        if K > REF_MAX_K or n > REF_MAX_N:
            raise _hip.PasnlUnsupported(
                f"knn_batch(tie_order='reference') covers K <= {REF_MAX_K} and N <= {REF_MAX_N} (got K={K}, N={n}): pass "
                "tie_order='index' for the canonical (distance, index) order, which differs only among exactly equal distances")
        return _knn_ref_dev(pts, queries, K, i64, out, max_workgroups, stats)
    out = _out_buffer(out, (b, m, int(K)), torch.int64 if i64 else torch.int32, pts.device)
    nbytes = int(_hip.lib().pasnl_knn_workspace_bytes(b, n)) if GRID and K <= 64 else 0
    if nbytes:  # large clouds: grid-pruned search in a scratch workspace (bit-identical results, csrc/knn_grid.hip)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=pts.device)
        if max_workgroups:  # a background search beside other work: a capped grid (pasnl_knn_batch_ws_bg)
            _hip.launch("pasnl_knn_batch_ws_bg", "knn_batch", b, n, m, int(K), _hip.ptr(pts), _hip.ptr(queries), _hip.ptr(out),
                        int(i64), _hip.ptr(None), _hip.ptr(ws), ctypes.c_size_t(nbytes), int(max_workgroups))
        else:
            _hip.launch("pasnl_knn_batch_ws", "knn_batch", b, n, m, int(K), _hip.ptr(pts), _hip.ptr(queries), _hip.ptr(out), int(i64),
                        _hip.ptr(None), _hip.ptr(ws), ctypes.c_size_t(nbytes))
        return out
    _hip.launch("pasnl_knn_batch", "knn_batch", b, n, m, int(K), _hip.ptr(pts), _hip.ptr(queries), _hip.ptr(out), int(i64),
                                          _hip.ptr(None))
    return out


def knn_batch(pts, queries, K, omp=False, dtype=None, tie_order="reference", out=None, max_workgroups=None, stats=None):
    """(B,N,3), (B,M,3) -> (B,M,K) neighbour indices (int64 like the reference; ``dtype=torch.int32`` skips
    the cast the models do at pointasnl_util.py:30).  ``omp`` is accepted and ignored.
    tie_order: "reference" (default, what the models use) = cpp_knn_batch's result bit for bit, its order among EXACTLY equal
    distances (nanoflann's KD-tree visit order) included: the canonical search; for the queries whose list contains or ends on a
    tie (none on clouds with distinct distances: the plain search's price) the runs of equal distances are put in that order -- for
    a few such queries from the tree nodes that separate their tied points alone, else through the rebuilt tree; K <= 256.
    "index" = ascending (distance, index), the canonical order: the same list
    wherever distances are distinct.  "nanoflann" = every query through the rebuilt tree (same result as "reference", slower:
    the checker).  A tree deeper than 96 levels raises a sticky flag: check_deferred_flags().
    out: optional device buffer (B,M,K) of the result's dtype to write into.  stats: a list that receives, per "reference"
    search, three (B,) int32 device tensors: the listed queries of each cloud, what the tie paths left to the builds, what the
    on-demand tree left to the full builds.
    max_workgroups: run the search of a large cloud as a background job on at most that many workgroups (a side stream's
    search beside other work; the same results, see pasnl_knn_batch_ws_bg)."""
    host = not isinstance(pts, torch.Tensor)
    p = _hip.as_dev(pts, torch.float32)
    q = _hip.as_dev(queries, torch.float32)
    i64 = dtype in (None, torch.int64, np.int64)
    if out is not None and host:
        raise ValueError("knn_batch: out= takes a device tensor (device inputs only)")
    out = _knn_dev(p, q, K, i64, tie_order, out, max_workgroups, stats)
    return out.cpu().numpy() if host else out


def _default_order(p, K):
    """single-cloud / legacy entry points: the reference's order where the tree kernels cover the shape, else canonical"""
    return "reference" if K <= REF_MAX_K and p.shape[-2] <= REF_MAX_N else "index"


def knn(pts, queries, K, omp=False):
    """single cloud: (N,3), (M,3) -> (M,K)   (knn.pyx:33-69)"""
    host = not isinstance(pts, torch.Tensor)
    p = _hip.as_dev(pts, torch.float32)[None]
    q = _hip.as_dev(queries, torch.float32)[None]
    out = _knn_dev(p, q, K, True, _default_order(p, K))[0]
    return out.cpu().numpy() if host else out


def knn_batch_distance_pick(pts, nqueries, K, omp=False, seed=None):
    """(B,N,3) -> (indices (B,nqueries,K) int64, queries (B,nqueries,3) float32)   (knn.pyx:111-148 -> knn_.cxx:136-266)
    Coverage-driven selection of `nqueries` query points per cloud (always among the points used least often so far) with
    their K nearest neighbours.  The reference seeds a std::mt19937 with time(0); `seed` makes the draw reproducible
    (None = the reference's behaviour: seconds since the epoch).  The same generator (numpy's MT19937 with legacy seeding ==
    std::mt19937(seed)) produces the stream on the host, the selection itself runs on the GPU.  `omp` is accepted and
    ignored (the reference's OpenMP variant shares one generator between threads without synchronisation)."""
    import time

    host = not isinstance(pts, torch.Tensor)
    p = _hip.as_dev(pts, torch.float32)
    if p.dim() != 3 or p.shape[2] != 3:
        raise ValueError("knn_batch_distance_pick expects (B,N,3) pts")
    b, n, _ = p.shape
    bg = np.random.MT19937()
    bg._legacy_seeding(int(time.time()) if seed is None else int(seed))
    rnd = torch.from_numpy(bg.random_raw(b * int(nqueries)).astype(np.uint32).view(np.int32)).to(p.device)
    idx = torch.empty((b, int(nqueries), int(K)), dtype=torch.int64, device=p.device)
    q = torch.empty((b, int(nqueries), 3), dtype=torch.float32, device=p.device)
    _hip.launch("pasnl_knn_distance_pick", "knn_batch_distance_pick", b, n, int(nqueries), int(K), _hip.ptr(p), _hip.ptr(rnd),
                _hip.ptr(idx), _hip.ptr(q))
    return (idx.cpu().numpy(), q.cpu().numpy()) if host else (idx, q)
