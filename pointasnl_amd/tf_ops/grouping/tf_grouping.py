"""tf_grouping -- drop-in for the reference module of the same name (tf_ops/grouping/tf_grouping.py:8-73)."""
import ctypes

import torch

from pointasnl_amd import _hip


def query_ball_point(radius, nsample, xyz1, xyz2):
    '''QueryBallPoint (tf_grouping.py:8-21).  For every query xyz2 (B,npoint,3): the first `nsample` indices (ascending) of
    xyz1 (B,ndataset,3) closer than `radius`, padded with the first hit.
    -> idx (B,npoint,nsample) int32, pts_cnt (B,npoint) int32 = hits found (capped at nsample).'''
    if not float(radius) > 0:
        raise ValueError("QueryBallPoint expects positive radius")
    if int(nsample) <= 0:
        raise ValueError("QueryBallPoint expects positive nsample")
    xyz1, xyz2 = _hip.as_dev(xyz1, torch.float32), _hip.as_dev(xyz2, torch.float32)
    if xyz1.dim() != 3 or xyz1.shape[2] != 3:
        raise ValueError("QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape.")
    if xyz2.dim() != 3 or xyz2.shape[2] != 3:
        raise ValueError("QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = torch.empty((b, m, int(nsample)), dtype=torch.int32, device=xyz1.device)
    pts_cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    _hip.launch("pasnl_query_ball_point", "QueryBallPoint", b, n, m, ctypes.c_float(float(radius)), int(nsample), _hip.ptr(xyz1),
                                                 _hip.ptr(xyz2), _hip.ptr(idx), _hip.ptr(pts_cnt))
    return idx, pts_cnt


def select_top_k(k, dist):
    '''SelectionSort (tf_grouping.py:23-34).  dist (b,m,n) f32, one row of n distances per query -> (idx, dist_out), both
    (b,m,n): the first k columns hold the k smallest distances, ascending, and where they came from; the rest of the row is
    whatever the partial selection sort left there (reproduced bit for bit).'''
    if int(k) <= 0:
        raise ValueError("SelectionSort expects positive k")
    dist = _hip.as_dev(dist, torch.float32)
    if dist.dim() != 3:
        raise ValueError("SelectionSort expects (b,m,n) dist shape.")
    b, m, n = dist.shape
    outi = torch.empty((b, m, n), dtype=torch.int32, device=dist.device)
    out = torch.empty((b, m, n), dtype=torch.float32, device=dist.device)
    _hip.launch("pasnl_select_top_k", "SelectionSort", b, n, m, int(k), _hip.ptr(dist), _hip.ptr(outi), _hip.ptr(out))
    return outi, out


class _GroupPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        b, n, c = points.shape
        _, m, ns = idx.shape
        out = torch.empty((b, m, ns, c), dtype=torch.float32, device=points.device)
        _hip.launch("pasnl_group_point", "GroupPoint", b, n, c, m, ns, _hip.ptr(points), _hip.ptr(idx), _hip.ptr(out))
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, grad_out):  # tf_grouping.py:42-46 -> GroupPointGrad
        (idx,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        b, m, ns, c = grad_out.shape
        g = torch.empty((b, ctx.n, c), dtype=torch.float32, device=grad_out.device)
        if _hip.DETERMINISTIC_GRADS:
            ws, nbytes = _hip.grad_workspace(b, ctx.n, m * ns, grad_out.device)
            _hip.launch("pasnl_group_point_grad_det", "GroupPointGrad", b, ctx.n, c, m, ns, _hip.ptr(grad_out), _hip.ptr(idx),
                        _hip.ptr(g), _hip.ptr(ws), nbytes)
        else:
            _hip.launch("pasnl_group_point_grad", "GroupPointGrad", b, ctx.n, c, m, ns, _hip.ptr(grad_out), _hip.ptr(idx), _hip.ptr(g))
        return g, None


def group_point(points, idx):
    '''GroupPoint (tf_grouping.py:36-45): rows idx (B,npoint,nsample) int32 of points (B,ndataset,C) f32
    -> (B,npoint,nsample,C) f32.  Differentiable w.r.t. points (tf_grouping.py:46-50).'''
    points, idx = _hip.as_dev(points, torch.float32), _hip.as_dev(idx, torch.int32)
    if points.dim() != 3:
        raise ValueError("GroupPoint expects (batch_size, num_points, channel) points shape")
    if idx.dim() != 3 or idx.shape[0] != points.shape[0]:
        raise ValueError("GroupPoint expects (batch_size, npoints, nsample) idx shape")
    return _GroupPoint.apply(points, idx)


def knn_point(k, xyz1, xyz2):
    '''tf_grouping.py:52-73.  k nearest rows of xyz1 (B,ndataset,c) for every row of xyz2 (B,npoint,c)
    -> val (B,npoint,k) f32 squared distances, idx (B,npoint,k) int32, nearest first.'''
    # Same composition as the reference (tf_grouping.py:58-71): broadcast squared distances, selection
    # sort, slice.  The (b,m,n) tensor is torch plumbing; the sort is the HIP kernel.
    xyz1, xyz2 = _hip.as_dev(xyz1, torch.float32), _hip.as_dev(xyz2, torch.float32)
    diff = xyz1[:, None, :, :] - xyz2[:, :, None, :]
    sq = diff * diff
    dist = sq[..., 0]
    for c in range(1, sq.shape[-1]):  # left-to-right reduce_sum over the coordinate axis
        dist = dist + sq[..., c]
    outi, out = select_top_k(k, dist.contiguous())
    idx = outi[:, :, :k].contiguous()
    val = out[:, :, :k].contiguous()
    return val, idx
