"""nearest_neighbors -- drop-in for the reference's Cython module (utils/nearest_neighbors/knn.pyx:33-148),
imported by the models as ``nearest_neighbors.lib.python.nearest_neighbors`` (pointasnl_util.py:19).

The reference builds a nanoflann KD-tree per cloud on the host (OpenMP over the batch) and is reached through
tf.py_func, i.e. a device->host->device round trip per layer.  Here the search is an exact brute-force gfx950
kernel; results are the K nearest in ascending (squared distance, index) order -- identical to nanoflann
whenever distances are distinct (nanoflann's order among exactly equal distances is traversal dependent).

numpy in -> numpy int64 out like the reference (host buffers cross PCIe); torch CUDA tensors in -> torch CUDA
tensors out (no copies), which is what utils/pointasnl_util.py uses.
"""
import numpy as np
import torch

import ctypes

from pointasnl_amd import _hip

GRID = True  # False: brute-force kernels for every size (A/B, and the reference point of the grid kernel's parity test)


# Tree-depth flags of tie_order="nanoflann" searches that were CAPTURED into a HIP graph: the flag (first word of the search's
# workspace) cannot be read while capturing, so it stays on the device -- the workspace is kept alive here -- and
# check_deferred_flags() reads them after a replay (one synchronisation per step instead of one per search).
_DEFERRED_FLAGS = []
_DEPTH_MSG = "knn_batch(tie_order='nanoflann'): a KD-tree deeper than 96 levels (pathologically clustered cloud)"


def check_deferred_flags(clear=False):
    """Raise PasnlUnsupported if any captured tie_order="nanoflann" search met a tree deeper than its stack (its rows are then
    undefined).  Call it after replaying the graph; synchronises with the device.  clear=True forgets the captured searches
    (their graph is gone)."""
    bad = any(int(f.item()) != 0 for f in _DEFERRED_FLAGS)
    if clear:
        _DEFERRED_FLAGS.clear()
    if bad:
        raise _hip.PasnlUnsupported(_DEPTH_MSG)


def _knn_tree_dev(pts, queries, K, i64, out=None):
    """The reference's own order among equal distances (csrc/knn_tree.hip): nanoflann's tree and search rebuilt on the GPU.
    Limits of the kernels (16-bit arrival / index packing, result sets in LDS): K <= 64, N <= 65535 -- beyond them the launcher
    answers PASNL_EUNSUPPORTED and this raises PasnlUnsupported (use the canonical order there: it differs only inside runs of
    exactly equal distances)."""
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
        _DEFERRED_FLAGS.append(flag)  # stays on the device: check_deferred_flags() after the replay
    elif int(flag.item()) != 0:  # eager: checked at once (a synchronisation: this mode is about exactness, not speed)
        raise _hip.PasnlUnsupported(_DEPTH_MSG)
    return out


def _out_buffer(t, shape, dtype, device):
    if t is None:
        return torch.empty(shape, dtype=dtype, device=device)
    if tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device or not t.is_contiguous():
        raise ValueError(f"knn_batch: out must be a contiguous {dtype} tensor of shape {tuple(shape)} on {device}")
    return t


def _knn_dev(pts, queries, K, i64, tie_order="index", out=None, max_workgroups=None):
    if tie_order == "nanoflann":
        if pts.dim() != 3 or pts.shape[2] != 3 or queries.dim() != 3 or queries.shape[2] != 3 or queries.shape[0] != pts.shape[0]:
            raise ValueError("knn_batch expects (B,N,3) pts and (B,M,3) queries")
        return _knn_tree_dev(pts, queries, K, i64, out)
    if tie_order != "index":
        raise ValueError("tie_order is 'index' (canonical (distance, index) order) or 'nanoflann' (the reference's visit order)")
    if pts.dim() != 3 or pts.shape[2] != 3 or queries.dim() != 3 or queries.shape[2] != 3:
        raise ValueError("knn_batch expects (B,N,3) pts and (B,M,3) queries")
    if queries.shape[0] != pts.shape[0]:
        raise ValueError("knn_batch expects the same batch size for pts and queries")
    b, n, _ = pts.shape
    m = queries.shape[1]
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


def knn_batch(pts, queries, K, omp=False, dtype=None, tie_order="index", out=None, max_workgroups=None):
    """(B,N,3), (B,M,3) -> (B,M,K) neighbour indices (int64 like the reference; ``dtype=torch.int32`` skips
    the cast the models do at pointasnl_util.py:30).  ``omp`` is accepted and ignored.
    tie_order: "index" (default) = ascending (distance, index), the canonical order and what the models use; "nanoflann" =
    the reference's own order among EXACTLY equal distances (its KD-tree's visit order), bit-identical to cpp_knn_batch on
    lattices and duplicated points too -- slower (the tree is rebuilt per call), for exact reproduction only; K <= 64 and
    N <= 65535 (PasnlUnsupported beyond).  Captured into a HIP graph its tree-depth flag stays on the device:
    check_deferred_flags() after the replay.  out: optional device buffer (B,M,K) of the result's dtype to write into.
    max_workgroups: run the search of a large cloud as a background job on at most that many workgroups (a side stream's
    search beside other work; the same results, see pasnl_knn_batch_ws_bg)."""
    host = not isinstance(pts, torch.Tensor)
    p = _hip.as_dev(pts, torch.float32)
    q = _hip.as_dev(queries, torch.float32)
    i64 = dtype in (None, torch.int64, np.int64)
    if out is not None and host:
        raise ValueError("knn_batch: out= takes a device tensor (device inputs only)")
    out = _knn_dev(p, q, K, i64, tie_order, out, max_workgroups)
    return out.cpu().numpy() if host else out


def knn(pts, queries, K, omp=False):
    """single cloud: (N,3), (M,3) -> (M,K)   (knn.pyx:33-69)"""
    host = not isinstance(pts, torch.Tensor)
    p = _hip.as_dev(pts, torch.float32)[None]
    q = _hip.as_dev(queries, torch.float32)[None]
    out = _knn_dev(p, q, K, True)[0]
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
