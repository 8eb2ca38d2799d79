"""tf_sampling -- drop-in for the reference module of the same name (tf_ops/sampling/tf_sampling.py:13-57).

Same function names, argument order, shapes, dtypes and results; tensors are torch CUDA(ROCm) tensors and the
work is done by hand-written gfx950 kernels behind the C ABI (include/pasnl.h).  Shape errors raise ValueError
with the reference's OP_REQUIRES message text (tf_sampling.cpp).
"""
import torch

from pointasnl_amd import _hip


def prob_sample(inp, inpr):
    '''ProbSample (tf_sampling.py:13-22): inp (B,ncategory) f32 unnormalised weights, inpr (B,npoints) f32 draws in [0,1)
    -> (B,npoints) int32, the category each draw falls into (inverse CDF).'''
    inp, inpr = _hip.as_dev(inp, torch.float32), _hip.as_dev(inpr, torch.float32)
    if inp.dim() != 2:
        raise ValueError("ProbSample expects (batch_size,num_choices) inp shape")
    b, n = inp.shape
    if inpr.dim() != 2 or inpr.shape[0] != b:
        raise ValueError("ProbSample expects (batch_size,num_points) inpr shape")
    m = inpr.shape[1]
    out = torch.empty((b, m), dtype=torch.int32, device=inp.device)
    temp = torch.empty((b, n), dtype=torch.float32, device=inp.device)  # tf_sampling.cpp:85 allocate_temp
    _hip.launch("pasnl_prob_sample", "ProbSample", b, n, m, _hip.ptr(inp), _hip.ptr(inpr), _hip.ptr(temp), _hip.ptr(out))
    return out


class _GatherPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, idx):
        b, n, _ = inp.shape
        m = idx.shape[1]
        out = torch.empty((b, m, 3), dtype=torch.float32, device=inp.device)
        _hip.launch("pasnl_gather_point", "GatherPoint", b, n, m, _hip.ptr(inp), _hip.ptr(idx), _hip.ptr(out))
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, out_g):  # tf_sampling.py:43-47 -> GatherPointGrad
        (idx,) = ctx.saved_tensors
        out_g = out_g.contiguous()
        b, m, _ = out_g.shape
        inp_g = torch.empty((b, ctx.n, 3), dtype=torch.float32, device=out_g.device)
        if _hip.DETERMINISTIC_GRADS:
            ws, nbytes = _hip.grad_workspace(b, ctx.n, m, out_g.device)
            _hip.launch("pasnl_gather_point_grad_det", "GatherPointGrad", b, ctx.n, m, _hip.ptr(out_g), _hip.ptr(idx),
                        _hip.ptr(inp_g), _hip.ptr(ws), nbytes)
        else:
            _hip.launch("pasnl_gather_point_grad", "GatherPointGrad", b, ctx.n, m, _hip.ptr(out_g), _hip.ptr(idx), _hip.ptr(inp_g))
        return inp_g, None


def gather_point(inp, idx):
    '''GatherPoint (tf_sampling.py:26-35): rows idx (B,npoints) int32 of inp (B,ndataset,3) f32 -> (B,npoints,3) f32.
    Differentiable w.r.t. inp (scatter-add, tf_sampling.py:37-43).'''
    inp, idx = _hip.as_dev(inp, torch.float32), _hip.as_dev(idx, torch.int32)
    if inp.dim() != 3 or inp.shape[2] != 3:
        raise ValueError("GatherPoint expects (batch_size,num_points,3) inp shape")
    if idx.dim() != 2 or idx.shape[0] != inp.shape[0]:
        raise ValueError("GatherPoint expects (batch_size,num_result) idx shape")
    return _GatherPoint.apply(inp, idx)


def farthest_point_sample(npoint, inp):
    '''FarthestPointSample (tf_sampling.py:45-54): inp (B,ndataset,3) f32 -> (B,npoint) int32, starting at point 0 and
    adding the point farthest from the chosen set each time (ties: lowest index).'''
    if int(npoint) <= 0:
        raise ValueError("FarthestPointSample expects positive npoint")
    inp = _hip.as_dev(inp, torch.float32)
    if inp.dim() != 3 or inp.shape[2] != 3:
        raise ValueError("FarthestPointSample expects (batch_size,num_points,3) inp shape")
    b, n, _ = inp.shape
    out = torch.empty((b, int(npoint)), dtype=torch.int32, device=inp.device)
    _hip.launch("pasnl_farthest_point_sample", "FarthestPointSample", b, n, int(npoint), _hip.ptr(inp), _hip.ptr(out))
    return out


def _out_buffer(t, shape, dtype, device, what):
    """a caller's output buffer: contiguous, of the result's shape / dtype / device (a serving loop hands its own buffers over
    so that nothing is copied behind the kernel)"""
    if t is None:
        return torch.empty(shape, dtype=dtype, device=device)
    if tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device or not t.is_contiguous():
        raise ValueError(f"{what}: out must be a contiguous {dtype} tensor of shape {tuple(shape)} on {device}")
    return t


def farthest_point_sample_gather(npoint, inp, out=None):
    '''farthest_point_sample + gather_point of its picks in ONE launch (not a symbol of the reference module: its callers run
    the two back to back, pointasnl_util.py:33-49, pointnet_util.py:44).  inp (B,ndataset,3) f32
    -> idx (B,npoint) int32, new_xyz (B,npoint,3) f32 == gather_point(inp, idx) bit for bit.  No gradient path.
    out: optional (idx, new_xyz) buffers to write into (either may be None).'''
    if int(npoint) <= 0:
        raise ValueError("FarthestPointSample expects positive npoint")
    inp = _hip.as_dev(inp, torch.float32)
    if inp.dim() != 3 or inp.shape[2] != 3:
        raise ValueError("FarthestPointSample expects (batch_size,num_points,3) inp shape")
    b, n, _ = inp.shape
    o_idx, o_xyz = out if out is not None else (None, None)
    idx = _out_buffer(o_idx, (b, int(npoint)), torch.int32, inp.device, "farthest_point_sample_gather")
    new_xyz = _out_buffer(o_xyz, (b, int(npoint), 3), torch.float32, inp.device, "farthest_point_sample_gather")
    _hip.launch("pasnl_farthest_point_sample_gather", "FarthestPointSample", b, n, int(npoint), _hip.ptr(inp), _hip.ptr(idx),
                _hip.ptr(new_xyz))
    return idx, new_xyz
