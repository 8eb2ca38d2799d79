"""pointnet_util -- the PointNet++ helpers the PointASNL models call (reference utils/pointnet_util.py:22-229):
sample_and_group, sample_and_group_all, pointnet_sa_module (max pooling) and pointnet_fp_module.
Inference only, torch device tensors; the native ops are the gfx950 kernels.  The MSG / pooling variants the
three models never reach are out of scope.
"""
import ctypes

import torch

from pointasnl_amd.tf_sampling import farthest_point_sample, farthest_point_sample_gather, gather_point
from pointasnl_amd.tf_grouping import query_ball_point, group_point, knn_point
from pointasnl_amd.tf_interpolate import three_nn, three_interpolate, three_weights, fp_interpolate_cat
from pointasnl_amd.utils import tf_util
from pointasnl_amd import _hip


def sample_and_group(npoint, radius, nsample, xyz, points, knn=False, use_xyz=True):
    '''pointnet_util.py:20-57.  FPS of npoint rows of xyz (B,N,3), then nsample neighbours each (kNN if knn, else ball of
    `radius`), coordinates made relative to the sampled row; `points` (B,N,C) rows are joined behind them when use_xyz.
    -> new_xyz (B,npoint,3), new_points (B,npoint,nsample,3+C), idx (B,npoint,nsample), grouped_xyz (B,npoint,nsample,3)'''
    _, new_xyz = farthest_point_sample_gather(npoint, xyz)
    if knn:
        _, idx = knn_point(nsample, xyz, new_xyz)
    else:
        idx, pts_cnt = query_ball_point(radius, nsample, xyz, new_xyz)
    grouped_xyz = group_point(xyz, idx)
    grouped_xyz = grouped_xyz - new_xyz.unsqueeze(2)
    if points is not None:
        grouped_points = group_point(points, idx)
        new_points = torch.cat([grouped_xyz, grouped_points], dim=-1) if use_xyz else grouped_points
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


_GROUP_ALL_CONST = {}


def max_pool_points(new_points, out=None):
    """tf.reduce_max(new_points, axis=[2], keep_dims=True) for (B, npoint, nsample, C) (pointnet_util.py:137).
    out: optional (B*npoint, C) float32 view whose rows may be slices of a wider table (unit column stride): the maxima are
    written there -- the caller's concat of several pooled vectors then costs nothing."""
    b, p, ns, c = new_points.shape
    new_points = new_points.contiguous()
    if out is None:
        out = torch.empty((b, p, 1, c), dtype=torch.float32, device=new_points.device)
        _hip.launch("pasnl_max_pool_rows", "max_pool_rows", b * p, ns, c, _hip.ptr(new_points), _hip.ptr(out))
        return out
    if tuple(out.shape) != (b * p, c) or out.stride(1) != 1 or out.dtype != torch.float32:
        raise ValueError("max_pool_points: out must be a (B*npoint, C) float32 view with unit column stride")
    _hip.launch("pasnl_max_pool_rows_strided", "max_pool_rows", b * p, ns, c, _hip.ptr(new_points), _hip.ptr(out),
                ctypes.c_long(out.stride(0) if b * p > 1 else c))
    return out.unflatten(0, (b, p)).unsqueeze(2)


# first-layer widths for which a group_all module runs on the fused kernel (csrc/mlp_pool.hip); the others run their three
# convolutions on the vendor GEMM + a pooling kernel.  Measured (EXPERIMENTS.md, round 5): the fused kernel streams every
# weight from L2 once per 32 / 64 rows and ties with the vendor chain at best -- see there for what is enabled and why.
FP_HEAD_FUSED = True  # pointnet_fp_module (inference, no autograd): three_weights + three_interpolate + concat as one kernel
GROUP_ALL_FUSED = (128,)
_GROUP_ALL_WS = {}


def group_all_mlp_max(new_points, mlp, bn, pooled_out=None):
    """The body of a group_all module (pointnet_util.py:123-137) in one launch (csrc/mlp_pool.hip): the three [1,1] convolutions
    of `mlp` (BN folded, ReLU) over all n points of every cloud and the maximum over the points.
    new_points (B,1,n,kp) [with `.input_pad` leading alignment columns the reference's tensor does not have] -> (B,1,1,mlp[-1]);
    the variables are the ones conv2d would create (scopes conv0..2).  Raises PasnlUnsupported outside the kernel's shapes."""
    b, one, n, kp = new_points.shape
    if one != 1 or len(mlp) != 3:
        raise _hip.PasnlUnsupported("group_all_mlp_max: one group per cloud, three convolutions")
    pad = getattr(new_points, "input_pad", 0)
    st = tf_util.store()
    cin, ws = kp - pad, []
    for i, c in enumerate(mlp):
        with tf_util.variable_scope('conv%d' % i):
            w, bb = st.layer(cin, c, bn)
            if i == 0 and pad:  # zero rows for the alignment columns (the same cached tensor tf_util._dense would build)
                key = st.path("") + "@pad%d" % pad
                if key not in st._folded:
                    st._folded[key] = (torch.cat([w.new_zeros((pad, w.shape[1])), w], dim=0).contiguous(), bb)
                w, bb = st._folded[key]
            # the kernel takes its weights in the matrix instruction's operand order (pasnl_mlp3_pack_weights), packed once
            key = st.path("") + "@mlp3:%x" % w.data_ptr()
            if key not in st._folded:
                wc = w.contiguous()
                pk = torch.empty(int(_hip.lib().pasnl_mlp3_packed_weights_bytes(wc.shape[0], wc.shape[1])) // 4, dtype=torch.float32,
                                 device=w.device)
                _hip.launch("pasnl_mlp3_pack_weights", "mlp3_pack", int(wc.shape[0]), int(wc.shape[1]), _hip.ptr(wc), _hip.ptr(pk))
                st._folded[key] = (pk, w)  # (keeps `w` alive: the pointer in the key stays unique)
            w = st._folded[key][0]
        ws += [w, bb]
        cin = c
    x = new_points.reshape(b, n, kp)
    if not x.is_contiguous():
        x = x.contiguous()
    c3 = mlp[-1]
    if pooled_out is None:
        out2d = torch.empty((b, c3), dtype=torch.float32, device=x.device)
    else:
        if tuple(pooled_out.shape) != (b, c3) or pooled_out.stride(1) != 1 or pooled_out.dtype != torch.float32:
            raise ValueError("group_all_mlp_max: pooled_out must be a (B, C) float32 view with unit column stride")
        out2d = pooled_out
    nbytes = int(_hip.lib().pasnl_mlp3_max_pool_workspace_bytes(b, n, c3))
    key = (x.device.index, torch.cuda.current_stream().cuda_stream, c3)
    bufs = _GROUP_ALL_WS.setdefault(key, [])  # grow-only, per stream: a captured graph keeps the buffer it was captured with
    if not bufs or bufs[-1].numel() < nbytes:
        bufs.append(torch.empty(max(nbytes, 256), dtype=torch.uint8, device=x.device))
    wsb = bufs[-1]
    _hip.launch("pasnl_mlp3_max_pool", "mlp3_max_pool", b, n, kp, mlp[0], mlp[1], mlp[2], _hip.ptr(x), *[_hip.ptr(t) for t in ws],
                _hip.ptr(out2d), ctypes.c_long(out2d.stride(0) if b > 1 else c3), _hip.ptr(wsb), ctypes.c_size_t(wsb.numel()))
    return out2d.unflatten(0, (b, 1)).unsqueeze(2)


def sample_and_group_all(xyz, points, use_xyz=True):
    '''
    Equivalent to sample_and_group with npoint=1, radius=inf, (0,0,0) as the centroid (pointnet_util.py:59-84).
    '''
    batch_size, nsample = xyz.shape[0], xyz.shape[1]
    # constants of the shape only (pointnet_util.py:72-73): built once per (B, nsample, device), not once per forward
    key = (batch_size, nsample, str(xyz.device))
    if key not in _GROUP_ALL_CONST:
        new_xyz = torch.zeros((batch_size, 1, 3), dtype=torch.float32, device=xyz.device)
        idx = torch.arange(nsample, dtype=torch.int32, device=xyz.device).reshape(1, 1, nsample).repeat(batch_size, 1, 1)
        _GROUP_ALL_CONST[key] = (new_xyz, idx)
    new_xyz, idx = _GROUP_ALL_CONST[key]
    grouped_xyz = xyz.reshape(batch_size, 1, nsample, 3)
    pre = getattr(points, "xyz_concat", None) if (points is not None and use_xyz) else None
    if pre is not None and pre[0] is xyz:
        # the producer of `points` already wrote [0 | xyz | points] rows (PointASNLSetAbstraction(xyz_concat=True)): one zero
        # column in front of the reference's concat -- pointnet_sa_module gives its first convolution a matching zero row
        new_points = pre[1].unsqueeze(1)
        new_points.input_pad = 1
    elif points is not None:
        new_points = torch.cat([xyz, points], dim=2) if use_xyz else points
        new_points = new_points.unsqueeze(1)
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def pointnet_sa_module(xyz, points, npoint, radius, nsample, mlp, mlp2, group_all, is_training, bn_decay, scope,
                       bn=True, pooling='max', knn=False, use_xyz=True, use_nchw=False, pooled_out=None):
    ''' PointNet Set Abstraction (SA) Module (pointnet_util.py:87-153), max pooling
        Return:
            new_xyz: (batch_size, npoint, 3), new_points: (batch_size, npoint, mlp[-1] or mlp2[-1]),
            idx: (batch_size, npoint, nsample)
    '''
    if pooling != 'max' or use_nchw:
        raise NotImplementedError("only pooling='max', NHWC is used by the PointASNL models")
    with tf_util.variable_scope(scope):
        if group_all:
            nsample = xyz.shape[1]
            new_xyz, new_points, idx, grouped_xyz = sample_and_group_all(xyz, points, use_xyz)
        else:
            new_xyz, new_points, idx, grouped_xyz = sample_and_group(npoint, radius, nsample, xyz, points, knn, use_xyz)
        fused = None
        if group_all and len(mlp) == 3 and mlp[0] in GROUP_ALL_FUSED and new_points.is_cuda and not is_training:
            try:
                fused = group_all_mlp_max(new_points, mlp, bn, pooled_out.reshape(-1, mlp[-1]) if pooled_out is not None else None)
            except _hip.PasnlUnsupported:
                fused = None  # other widths: layer by layer below
        if fused is not None:
            new_points = fused
        else:
            for i, num_out_channel in enumerate(mlp):
                new_points = tf_util.conv2d(new_points, num_out_channel, [1, 1], padding='VALID', stride=[1, 1], bn=bn,
                                            is_training=is_training, scope='conv%d' % (i), bn_decay=bn_decay,
                                            input_pad=getattr(new_points, "input_pad", 0) if i == 0 else 0)
            new_points = max_pool_points(new_points, out=pooled_out)
        if mlp2 is not None:
            for i, num_out_channel in enumerate(mlp2):
                new_points = tf_util.conv2d(new_points, num_out_channel, [1, 1], padding='VALID', stride=[1, 1], bn=bn,
                                            is_training=is_training, scope='conv_post_%d' % (i), bn_decay=bn_decay)
        new_points = new_points.squeeze(2)
        return new_xyz, new_points, idx


def pointnet_fp_module(xyz1, xyz2, points1, points2, mlp, is_training, bn_decay, scope, bn=True, nn=None):
    ''' PointNet Feature Propogation (FP) Module (pointnet_util.py:199-229)
        Input:
            xyz1: (batch_size, ndataset1, 3), xyz2: (batch_size, ndataset2, 3) sparser than xyz1
            points1: (batch_size, ndataset1, nchannel1), points2: (batch_size, ndataset2, nchannel2)
        Return:
            new_points: (batch_size, ndataset1, mlp[-1])
        nn: three_nn(xyz1, xyz2) computed ahead by the caller (tuple or pointasnl_util.Forked): it reads coordinates only
    '''
    with tf_util.variable_scope(scope):
        dist, idx = three_nn(xyz1, xyz2) if nn is None else (nn.get() if hasattr(nn, "get") else nn)
        if FP_HEAD_FUSED and not is_training and not torch.is_grad_enabled():
            # weights + interpolation + concat in one launch, the same bits (pasnl_fp_interpolate_cat)
            new_points1 = fp_interpolate_cat(points2, idx, dist, points1)
        else:
            weight = three_weights(dist)  # pointnet_util.py:212-215 as one kernel
            interpolated_points = three_interpolate(points2, idx, weight)
            if points1 is not None:
                new_points1 = torch.cat([interpolated_points, points1], dim=2)
            else:
                new_points1 = interpolated_points
        new_points1 = new_points1.unsqueeze(2)
        for i, num_out_channel in enumerate(mlp):
            new_points1 = tf_util.conv2d(new_points1, num_out_channel, [1, 1], padding='VALID', stride=[1, 1], bn=bn,
                                         is_training=is_training, scope='conv_%d' % (i), bn_decay=bn_decay)
        return new_points1.squeeze(2)
