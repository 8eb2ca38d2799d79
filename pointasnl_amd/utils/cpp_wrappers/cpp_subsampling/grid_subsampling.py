"""grid_subsampling -- drop-in for the reference extension module of the same name
(utils/cpp_wrappers/cpp_subsampling/wrapper.cpp; called as `cpp_subsampling.compute(...)` from
ScanNet/scannet_dataset_grid.py:20-38 and the SemanticKITTI grid loader).

Same call signature and return structure (numpy in, numpy out).  Rows come out in ascending voxel key, not in the
reference's hash-table order, and a label tie goes to the smallest label (include/pasnl.h).
"""
import ctypes

import numpy as np
import torch

from pointasnl_amd import _hip


def compute(points, features=None, classes=None, sampleDl=0.1, verbose=0):
    """points (N,3) float32 [, features (N,fdim) float32] [, classes (N,ldim) int32] -> subsampled arrays:
    points | (points, features) | (points, classes) | (points, features, classes), as the reference returns them."""
    _hip.require_device()
    pts = _hip.as_dev(np.ascontiguousarray(points, dtype=np.float32) if not isinstance(points, torch.Tensor) else points,
                      torch.float32)
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("points.shape is not (N, 3)")  # wrapper.cpp's message
    n = pts.shape[0]
    feats = cls = None
    fdim = ldim = 0
    if features is not None:
        feats = _hip.as_dev(np.ascontiguousarray(features, dtype=np.float32) if not isinstance(features, torch.Tensor)
                            else features, torch.float32)
        if feats.dim() != 2 or feats.shape[0] != n:
            raise ValueError("features.shape is not (N, d)")
        fdim = feats.shape[1]
    if classes is not None:
        cls = _hip.as_dev(np.ascontiguousarray(classes, dtype=np.int32) if not isinstance(classes, torch.Tensor) else classes,
                          torch.int32)
        if cls.dim() == 1:
            cls = cls.reshape(n, 1)
        if cls.dim() != 2 or cls.shape[0] != n:
            raise ValueError("classes.shape is not (N,) or (N, d)")
        ldim = cls.shape[1]
    dev = pts.device
    out_p = torch.empty((n, 3), dtype=torch.float32, device=dev)
    out_f = torch.empty((n, fdim), dtype=torch.float32, device=dev)
    out_c = torch.empty((n, ldim), dtype=torch.int32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    nbytes = int(_hip.lib().pasnl_grid_subsample_workspace_bytes(ctypes.c_long(n)))
    ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
    _hip.launch("pasnl_grid_subsample", "grid_subsampling", ctypes.c_long(n), fdim, ldim, _hip.ptr(pts), _hip.ptr(feats),
                _hip.ptr(cls), ctypes.c_float(float(sampleDl)), _hip.ptr(out_p), _hip.ptr(out_f), _hip.ptr(out_c),
                _hip.ptr(count), _hip.ptr(ws), ctypes.c_size_t(nbytes))
    m = int(count.item())  # the only synchronisation: the caller wants host arrays
    out = [out_p[:m].cpu().numpy()]
    if features is not None:
        out.append(out_f[:m].cpu().numpy())
    if classes is not None:
        out.append(out_c[:m].cpu().numpy())
    return out[0] if len(out) == 1 else tuple(out)
