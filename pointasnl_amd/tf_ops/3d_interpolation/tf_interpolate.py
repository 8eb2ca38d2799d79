"""tf_interpolate -- drop-in for the reference module (tf_ops/3d_interpolation/tf_interpolate.py:8-35).

The reference runs these two ops as single-threaded CPU kernels (tf_interpolate.cpp:187,222); here they are
gfx950 kernels on the caller's stream, so the decoder never leaves the GPU.
"""
import torch

from pointasnl_amd import _hip


def three_nn(xyz1, xyz2):
    '''ThreeNN (tf_interpolate.py:8-18).  For every row of xyz1 (b,n,3) its three nearest rows of xyz2 (b,m,3)
    -> dist (b,n,3) f32 (squared), idx (b,n,3) int32, nearest first.'''
    xyz1, xyz2 = _hip.as_dev(xyz1, torch.float32), _hip.as_dev(xyz2, torch.float32)
    if xyz1.dim() != 3 or xyz1.shape[2] != 3:
        raise ValueError("ThreeNN expects (b,n,3) xyz1 shape.")
    if xyz2.dim() != 3 or xyz2.shape[2] != 3:
        raise ValueError("ThreeNN expects (b,m,3) xyz2 shape.")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    _hip.launch("pasnl_three_nn", "ThreeNN", b, n, m, _hip.ptr(xyz1), _hip.ptr(xyz2), _hip.ptr(dist), _hip.ptr(idx))
    return dist, idx


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx, weight):
        b, m, c = points.shape
        n = idx.shape[1]
        out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
        _hip.launch("pasnl_three_interpolate", "ThreeInterpolate", b, m, c, n, _hip.ptr(points), _hip.ptr(idx), _hip.ptr(weight),
                                                      _hip.ptr(out))
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, grad_out):  # tf_interpolate.py:29-34 -> ThreeInterpolateGrad
        idx, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        b, n, c = grad_out.shape
        g = torch.empty((b, ctx.m, c), dtype=torch.float32, device=grad_out.device)
        if _hip.DETERMINISTIC_GRADS:
            ws, nbytes = _hip.grad_workspace(b, ctx.m, 3 * n, grad_out.device)
            _hip.launch("pasnl_three_interpolate_grad_det", "ThreeInterpolateGrad", b, n, c, ctx.m, _hip.ptr(grad_out),
                        _hip.ptr(idx), _hip.ptr(weight), _hip.ptr(g), _hip.ptr(ws), nbytes)
        else:
            _hip.launch("pasnl_three_interpolate_grad", "ThreeInterpolateGrad", b, n, c, ctx.m, _hip.ptr(grad_out), _hip.ptr(idx),
                        _hip.ptr(weight), _hip.ptr(g))
        return g, None, None


def three_interpolate(points, idx, weight):
    '''ThreeInterpolate (tf_interpolate.py:20-30).  out[b,i,:] = sum_j weight[b,i,j] * points[b, idx[b,i,j], :]
    points (b,m,c), idx (b,n,3) int32, weight (b,n,3) -> (b,n,c).  Differentiable w.r.t. points (tf_interpolate.py:31-35).'''
    points = _hip.as_dev(points, torch.float32)
    idx = _hip.as_dev(idx, torch.int32)
    weight = _hip.as_dev(weight, torch.float32)
    if points.dim() != 3:
        raise ValueError("ThreeInterpolate expects (b,m,c) points shape")
    b = points.shape[0]
    if idx.dim() != 3 or idx.shape[0] != b or idx.shape[2] != 3:
        raise ValueError("ThreeInterpolate expects (b,n,3) idx shape")
    if weight.dim() != 3 or tuple(weight.shape) != (b, idx.shape[1], 3):
        raise ValueError("ThreeInterpolate expects (b,n,3) weight shape")
    return _ThreeInterpolate.apply(points, idx, weight)


def three_weights(dist):
    """The four TF ops at pointasnl_util.py:308-311 / pointnet_util.py:212-215 as one kernel:
    d=max(d,1e-10); w=(1/d)/sum(1/d).  dist (b,n,3) -> weight (b,n,3)."""
    dist = _hip.as_dev(dist, torch.float32)
    w = torch.empty_like(dist)
    _hip.launch("pasnl_three_weights", "ThreeWeights", dist.numel() // 3, _hip.ptr(dist), _hip.ptr(w))
    return w


def fp_interpolate_cat(points2, idx, dist, points1=None):
    """three_weights + three_interpolate + tf.concat([interpolated, points1], axis=2) in one launch (pointnet_util.py:212-219,
    pointasnl_util.py:308-313), bit-identical to the chain; inference only (no autograd node).
    points2 (b,m,c2), idx / dist (b,n,3) from three_nn, points1 (b,n,c1) or None -> (b,n,c2+c1)."""
    points2 = _hip.as_dev(points2, torch.float32)
    idx, dist = _hip.as_dev(idx, torch.int32), _hip.as_dev(dist, torch.float32)
    if points2.dim() != 3 or idx.dim() != 3 or idx.shape[2] != 3 or tuple(dist.shape) != tuple(idx.shape) or idx.shape[0] != points2.shape[0]:
        raise ValueError("fp_interpolate_cat expects (b,m,c2) points2 and (b,n,3) idx / dist")
    b, m, c2 = points2.shape
    n = idx.shape[1]
    c1 = 0
    if points1 is not None:
        points1 = _hip.as_dev(points1, torch.float32)
        if points1.dim() != 3 or points1.shape[0] != b or points1.shape[1] != n:
            raise ValueError("fp_interpolate_cat expects (b,n,c1) points1")
        c1 = points1.shape[2]
    out = torch.empty((b, n, c2 + c1), dtype=torch.float32, device=points2.device)
    _hip.launch("pasnl_fp_interpolate_cat", "FpInterpolateCat", b, m, c2, n, c1, _hip.ptr(points2), _hip.ptr(idx), _hip.ptr(dist),
                _hip.ptr(points1), _hip.ptr(out))
    return out
