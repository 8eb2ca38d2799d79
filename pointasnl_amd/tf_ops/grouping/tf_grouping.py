"""tf_grouping -- drop-in for the reference module of the same name (tf_ops/grouping/tf_grouping.py:8-73)."""
import ctypes

import torch

from pointasnl_amd import _hip


def query_ball_point(radius, nsample, xyz1, xyz2):
    '''
    Input:
        radius: float32, ball search radius
        nsample: int32, number of points selected in each ball region
        xyz1: (batch_size, ndataset, 3) float32 array, input points
        xyz2: (batch_size, npoint, 3) float32 array, query points
    Output:
        idx: (batch_size, npoint, nsample) int32 array, indices to input points
        pts_cnt: (batch_size, npoint) int32 array, number of unique points in each local region
    '''
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
    '''
    Input:
        k: int32, number of k SMALLEST elements selected
        dist: (b,m,n) float32 array, distance matrix, m query points, n dataset points
    Output:
        idx: (b,m,n) int32 array, first k in n are indices to the top k
        dist_out: (b,m,n) float32 array, first k in n are the top k
    '''
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
    '''
    Input:
        points: (batch_size, ndataset, channel) float32 array, points to sample from
        idx: (batch_size, npoint, nsample) int32 array, indices to points
    Output:
        out: (batch_size, npoint, nsample, channel) float32 array, values sampled from points
    '''
    points, idx = _hip.as_dev(points, torch.float32), _hip.as_dev(idx, torch.int32)
    if points.dim() != 3:
        raise ValueError("GroupPoint expects (batch_size, num_points, channel) points shape")
    if idx.dim() != 3 or idx.shape[0] != points.shape[0]:
        raise ValueError("GroupPoint expects (batch_size, npoints, nsample) idx shape")
    return _GroupPoint.apply(points, idx)


def knn_point(k, xyz1, xyz2):
    '''
    Input:
        k: int32, number of k in k-nn search
        xyz1: (batch_size, ndataset, c) float32 array, input points
        xyz2: (batch_size, npoint, c) float32 array, query points
    Output:
        val: (batch_size, npoint, k) float32 array, L2 distances
        idx: (batch_size, npoint, k) int32 array, indices to input points
    '''
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
