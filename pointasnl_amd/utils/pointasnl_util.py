"""pointasnl_util -- host-side mirror of the reference's utils/pointasnl_util.py (same function names,
argument order and results), inference only, on torch device tensors.

What changes underneath (and nothing else):
  * farthest_point_sample / kNN / gathers / three_nn / three_interpolate are the gfx950 kernels behind the
    C ABI -- no tf.py_func, no host KD-tree, no CPU ops, so a layer never leaves the GPU
    (reference: GPU -> CPU -> GPU per layer);
  * the attention cores of SampleWeights and PointNonLocalCell are fused kernels: the (B,P,N) attention map
    of pointasnl_util.py:199-212 is never written to HBM;
  * 1x1 convolutions + inference BN are folded GEMMs on the vendor BLAS (tf_util.py here).
"""

import ctypes

import torch

from pointasnl_amd import _hip
from pointasnl_amd import tf_sampling, tf_grouping, nearest_neighbors
from pointasnl_amd.tf_interpolate import three_nn, three_interpolate, three_weights, fp_interpolate_cat
from pointasnl_amd.utils import tf_util

NL_KEY_PARTS = True  # nl_attention: keys over workgroups where the query tiles alone leave CUs empty (pasnl_nl_attention_ws)
NL_VARIANT = 0  # 0 auto / 1 vector-FMA / 2 MFMA  (pasnl_nl_attention); bench.py --ops sweeps it


KNN_TIE_ORDER = "reference"  # nearest_neighbors.knn_batch's tie_order for every search of the models: "reference" = cpp_knn_batch's
#                              result bit for bit, nanoflann's order among exactly equal distances included (the canonical search +
#                              the rebuilt KD-tree for the queries that have such a tie; free on tie-free clouds); "index" = the
#                              canonical (distance, index) order (A/B); "nanoflann" = every query through the tree (the checker).


def knn_query(k, support_pts, query_pts, out=None, max_workgroups=None):
    """Mirror of pointasnl_util.py:22-30.  support_pts (B,N1,3), query_pts (B,N2,3) -> (B,N2,k) int32: for every query the
    indices of its k nearest support points, nearest first.  No host round trip here: the search is a gfx950 kernel.
    out: optional (B,N2,k) int32 buffer to write into; max_workgroups: a background search on a capped grid (knn_batch)."""
    return nearest_neighbors.knn_batch(support_pts, query_pts, k, omp=True, dtype=torch.int32, tie_order=KNN_TIE_ORDER, out=out,
                                       max_workgroups=max_workgroups)


def _gather_rows(points, idx):
    """tf.gather_nd(points, [batch, idx]) for idx (B,M): one C-wide row per index (pointasnl_util.py:43-49)."""
    return tf_grouping.group_point(points, idx.unsqueeze(-1)).squeeze(2)


def sampling(npoint, pts, feature=None):
    '''Mirror of pointasnl_util.py:33-49.  Farthest point sampling of `npoint` rows of pts (B,N,D); returns the sampled rows
    (B,npoint,D), and the same rows of `feature` when it is given.'''
    if pts.shape[-1] == 3:
        fps_idx, sub_pts = tf_sampling.farthest_point_sample_gather(npoint, pts)
    else:
        fps_idx = tf_sampling.farthest_point_sample(npoint, pts)
        sub_pts = _gather_rows(pts, fps_idx)
    if feature is None:
        return sub_pts
    return sub_pts, _gather_rows(feature, fps_idx)


def grouping(feature, K, src_xyz, q_xyz, use_xyz=True, use_knn=True, radius=0.2):
    '''Mirror of pointasnl_util.py:51-76.  For every row of q_xyz (B,P,3) its K neighbours among src_xyz (B,N,3) -- kNN, or
    a ball query of `radius` -- and the gathered coordinates / features (with the coordinates prepended when use_xyz).
    -> (grouped_xyz (B,P,K,3), grouped_feature (B,P,K,[3+]C), indices (B,P,K))'''
    if use_knn:
        point_indices = knn_query(K, src_xyz, q_xyz)
    else:
        # the reference branch never defines `idx` and raises NameError at pointasnl_util.py:71; this is the
        # evident intent (ball query indices used for both gathers)
        point_indices, _ = tf_grouping.query_ball_point(radius, K, src_xyz, q_xyz)
    grouped_xyz = tf_grouping.group_point(src_xyz, point_indices)
    grouped_feature = tf_grouping.group_point(feature, point_indices)
    if use_xyz:
        grouped_feature = torch.cat([grouped_xyz, grouped_feature], dim=-1)
    return grouped_xyz, grouped_feature, point_indices


def sa_group(xyz, feature, idx, new_xyz):
    """Fused grouping of one set-abstraction layer (pointasnl_util.py:63-74, 248-249, 258):
        new_point[b,j,s,:] = [xyz[i]-new_xyz[j] | xyz[i] | feature[i]],  i = idx[b,j,s];   skip = max_s new_point
    One kernel instead of two gathers, two concats, a subtraction and a reduce_max.
    -> new_point (B,P,K,6+C), skip (B,P,6+C)"""
    b, n, c = feature.shape
    _, p, k = idx.shape
    xyz, feature, idx, new_xyz = xyz.contiguous(), feature.contiguous(), idx.contiguous(), new_xyz.contiguous()
    new_point = torch.empty((b, p, k, 6 + c), dtype=torch.float32, device=xyz.device)
    skip = torch.empty((b, p, 6 + c), dtype=torch.float32, device=xyz.device)
    _hip.launch("pasnl_sa_group", "sa_group", b, n, c, p, k, _hip.ptr(xyz), _hip.ptr(feature), _hip.ptr(idx),
                _hip.ptr(new_xyz), _hip.ptr(new_point), _hip.ptr(skip))
    return new_point, skip


SELF_KNN_RATIO = 2       # sa_search: the self-kNN of the level (then a row gather) beside the sampler when at least 1 / RATIO of the points are
#                          sampled.  (4 -- layer 2 of the classifier, whose search sits on the critical path of the --AS model -- puts a
#                          fork inside a forked search: the captured graph then crashed inside hipGraphLaunch, round 6.  Leave it at 2.)
SA_TAIL_MIN_ROWS = 2048
FP_HEAD_FUSED = True     # PointASNLDecodingLayer (inference, no autograd): three_weights + three_interpolate as one kernel
SA_CELL_PACKED = True    # the wide cells (one workgroup per group, weights from L2) get their matrices packed in operand order too
SA_TAIL_PACKED = True    # pasnl_sa_tail with its weights packed in operand order (16-byte weight loads)
SA_TAIL_FUSED = True     # skip conv + back-projection + adds + aggregation in one kernel (pasnl_sa_tail); False = op by op
SA_CELL_GATHER = True    # grouping fused into the local cell (pasnl_sa_cell); False = pasnl_sa_group + pasnl_sa_local_cell
SA_CELL_SINGLE = True     # mlp = [c, c] (one convolution, c = 32 / 64 / 128): no conv1 at all instead of an identity conv1 (False: A/B)
SA_CELL_WIDE = True      # the 256- / 512-channel layers on pasnl_sa_cell too (False: pasnl_sa_group + the vendor chain, A/B)
LOCAL_CELL_FUSED = True  # False = the reference's op-by-op chain on the vendor BLAS (kept for A/B and as fallback)


def _local_cell_supported(w, mlp, nsample):
    """Cheap pre-filter only: the kernel launchers own the exact limits (LDS bytes: cells.hip pasnl_sa_cell) and answer
    PASNL_EUNSUPPORTED beyond them, which PointASNLSetAbstraction catches and answers with the next path down."""
    if not LOCAL_CELL_FUSED or nsample % 32:
        return False
    wide = (256, 512) if SA_CELL_WIDE else ()  # one workgroup per group, weights streamed from L2 (sa_cell_wide_kernel)
    if len(mlp) == 3:  # conv0 -> conv1 -> (weight net, matmul) -> after_conv
        return mlp[0] == mlp[1] and mlp[0] in (16, 32, 64, 128) + wide
    return len(mlp) == 2 and mlp[0] in (32, 64, 128) + wide  # one convolution only (the *_2 layers of pointasnl_sem_seg_res.py)


def sa_local_cell(new_point, mlp, is_training, bn_decay, weight_decay, bn):
    """Fused local cell (pointasnl_util.py:264-274): conv0 -> conv1 on the grouped points, the weight net on the
    centred coordinates, and H2^T . G, in one MFMA kernel.  (B,P,K,6+C) -> (B,P,mlp[1],32) = the input of after_conv.
    The variables are the same ones the op-by-op chain would create (scopes conv0, conv1, weight_net/wconv0)."""
    b, p, k, w = new_point.shape
    c1, c2 = mlp[0], mlp[1]
    st = tf_util.store()
    with tf_util.variable_scope('conv0'):
        w0, b0 = st.layer(w, c1, bn, weight_decay)
    with tf_util.variable_scope('conv1'):
        w1, b1 = st.layer(c1, c2, bn, weight_decay)
    with tf_util.variable_scope('weight_net'), tf_util.variable_scope('wconv0'):
        ww, bw = st.layer(3, 32, True, weight_decay)
    new_point = new_point.contiguous()
    out = torch.empty((b, p, c2, 32), dtype=torch.float32, device=new_point.device)
    _hip.launch("pasnl_sa_local_cell", "sa_local_cell", b * p, k, w, c1, c2, _hip.ptr(new_point), _hip.ptr(w0), _hip.ptr(b0),
                _hip.ptr(w1), _hip.ptr(b1), _hip.ptr(ww), _hip.ptr(bw), _hip.ptr(out))
    return out


SA_CELL16 = True  # the 16-channel first layer on its own 16x16x4 kernel (False: zero-padded on the 32-channel kernel, A/B)


def _sa_cell_weights(w_in, mlp, bn, weight_decay, native16=False, single=False):
    """(w0, b0, w1, b1, ww, bw, c_kernel) for pasnl_sa_cell, whose two convolutions are c x c with c in {32, 64, 128}:
      * mlp = [c, c, out]: the layer's own conv0 / conv1;
      * mlp = [16, 16, out] (pointasnl_sem_seg_res.py layer0): as they are where the 16-channel kernel applies (xyz-only rows,
        32 neighbours, centres from a table: native16); otherwise zero-padded to 32 channels -- the padded channels are
        relu(0 + 0) = 0 and feed zero rows, so channels 0..15 are bit-identical to the unpadded arithmetic;
      * mlp = [c, out] (the *_2 residual layers: ONE convolution, pointasnl_util.py:264-269 with len(mlp) == 2): w1 = None, the
        kernels skip conv1 (SA_CELL_SINGLE = False: conv1 = the identity with zero bias -- relu(h * 1 + 0) = h exactly for
        h = relu(.) >= 0 -- the same bits for 43-47 % more matrix work).
    The folded / padded tensors are cached in the store like every other folded weight."""
    st = tf_util.store()
    key = st.path("sa_cell_weights16" if native16 else ("sa_cell_weights1" if single else "sa_cell_weights"))
    if key not in st._folded:
        c1 = mlp[0]
        with tf_util.variable_scope('conv0'):
            w0, b0 = st.layer(w_in, c1, bn, weight_decay)
        if len(mlp) == 3:
            with tf_util.variable_scope('conv1'):
                w1, b1 = st.layer(c1, mlp[1], bn, weight_decay)
        elif c1 >= 256 or single:
            w1, b1 = None, None  # the kernels take "no conv1" as such (no c x c identity product)
        else:
            w1, b1 = torch.eye(c1, dtype=torch.float32, device=w0.device), torch.zeros(c1, dtype=torch.float32, device=w0.device)
        with tf_util.variable_scope('weight_net'), tf_util.variable_scope('wconv0'):
            ww, bw = st.layer(3, 32, True, weight_decay)
        ck = c1 if native16 else max(c1, 32)
        if ck != c1:
            pad = ck - c1
            w0 = torch.nn.functional.pad(w0, (0, pad))
            b0 = torch.nn.functional.pad(b0, (0, pad))
            w1 = torch.nn.functional.pad(w1, (0, pad, 0, pad))
            b1 = torch.nn.functional.pad(b1, (0, pad))
        st._folded[key] = tuple(None if t is None else t.contiguous() for t in (w0, b0, w1, b1, ww, bw)) + (ck,)
    return st._folded[key]


def _cell_packed(st, w, first_row):
    """rows first_row.. of `w` (k, c) in the operand order of the one-workgroup-per-group cells (pasnl_mlp3_pack_weights' layout),
    packed once per variable"""
    key = "@cellpk%d:%x" % (first_row, w.data_ptr())
    if key not in st._folded:
        wc = w[first_row:].contiguous()
        pk = torch.empty(int(_hip.lib().pasnl_mlp3_packed_weights_bytes(wc.shape[0], wc.shape[1])) // 4, dtype=torch.float32,
                         device=w.device)
        _hip.launch("pasnl_mlp3_pack_weights", "cell_pack", int(wc.shape[0]), int(wc.shape[1]), _hip.ptr(wc), _hip.ptr(pk))
        st._folded[key] = (pk, w)  # (keeps `w` alive: the pointer in the key stays unique)
    return st._folded[key][0]


def sa_cell(xyz, feature, idx, new_xyz, mlp, is_training, bn_decay, weight_decay, bn):
    """Grouping + local cell in ONE kernel (pointasnl_util.py:63-74,248-249,258,264-274): the (B,P,K,6+C) grouped
    tensor is never materialised -- rows are gathered from the L2-resident per-cloud tables inside the MFMA
    kernel, which also takes the skip connection's max over the K neighbours.
    -> (B,P,c,32) = the input of after_conv (c = the width of the last convolution; a view of the kernel's 32-channel
       output when c = 16),  skip (B,P,6+C)"""
    b, n, c = feature.shape
    _, p, k = idx.shape
    native16 = SA_CELL16 and len(mlp) == 3 and mlp[0] == 16 and c == 3 and k == 32 and new_xyz is not None
    # one convolution without an identity conv1: where the kernels have that form (16-byte rows that end with an 8-step chunk)
    xyz, feature, idx = xyz.contiguous(), feature.contiguous(), idx.contiguous()
    # (an unaligned feature view keeps the fused kernel: the identity-conv1 form takes rows that are not 16-byte aligned)
    single = (SA_CELL_SINGLE and len(mlp) == 2 and mlp[0] in (32, 64, 128) and c % 32 == 0 and k == 32
              and feature.data_ptr() % 16 == 0)
    w0, b0, w1, b1, ww, bw, ck = _sa_cell_weights(6 + c, mlp, bn, weight_decay, native16, single)
    out = torch.empty((b, p, ck, 32), dtype=torch.float32, device=xyz.device)
    skip = torch.empty((b, p, 6 + c), dtype=torch.float32, device=xyz.device)
    c_out = mlp[0] if len(mlp) == 2 else mlp[1]
    view = out if c_out == ck else out[:, :, :c_out, :]
    if SA_CELL_PACKED and ck in (128, 256, 512) and c % 16 == 0 and c >= 32 and k == 32:
        # the shapes the one-workgroup-per-group kernels take: the feature rows of w0 and w1 ALSO in the matrix instruction's
        # operand order (packed once, cached); the library uses them where its kernel reads weights from L2, ignores them elsewhere
        st = tf_util.store()
        w0p = _cell_packed(st, w0, 6)
        w1p = _cell_packed(st, w1, 0) if w1 is not None else None
        cen = nf = None
        if new_xyz is None:
            cen = torch.empty((b, p, 3), dtype=torch.float32, device=xyz.device)
            nf = torch.empty((b, p, 3 + c), dtype=torch.float32, device=xyz.device)
        else:
            new_xyz = new_xyz.contiguous()
        _hip.launch("pasnl_sa_cell_packed", "sa_cell", b, n, c, p, k, ck, ck, _hip.ptr(xyz), _hip.ptr(feature), _hip.ptr(idx),
                    _hip.ptr(new_xyz), _hip.ptr(w0), _hip.ptr(b0), _hip.ptr(w1), _hip.ptr(b1), _hip.ptr(ww), _hip.ptr(bw),
                    _hip.ptr(w0p), _hip.ptr(w1p), _hip.ptr(out), _hip.ptr(skip), _hip.ptr(cen), _hip.ptr(nf))
        return (view, skip) if new_xyz is not None else (view, skip, cen, nf)
    if new_xyz is None:
        # the groups' centres are their neighbour 0 (as_neighbor == 0): the kernel takes them from its own tiles and also
        # returns new_xyz (B,P,3) and new_feature (B,P,3+C) = [centre | neighbour 0's feature row]
        cen = torch.empty((b, p, 3), dtype=torch.float32, device=xyz.device)
        nf = torch.empty((b, p, 3 + c), dtype=torch.float32, device=xyz.device)
        _hip.launch("pasnl_sa_cell_centre0", "sa_cell", b, n, c, p, k, ck, ck, _hip.ptr(xyz), _hip.ptr(feature), _hip.ptr(idx),
                    _hip.ptr(w0), _hip.ptr(b0), _hip.ptr(w1), _hip.ptr(b1), _hip.ptr(ww), _hip.ptr(bw), _hip.ptr(out),
                    _hip.ptr(skip), _hip.ptr(cen), _hip.ptr(nf))
        return view, skip, cen, nf
    new_xyz = new_xyz.contiguous()
    _hip.launch("pasnl_sa_cell", "sa_cell", b, n, c, p, k, ck, ck, _hip.ptr(xyz), _hip.ptr(feature), _hip.ptr(idx),
                _hip.ptr(new_xyz), _hip.ptr(w0), _hip.ptr(b0), _hip.ptr(w1), _hip.ptr(b1), _hip.ptr(ww), _hip.ptr(bw),
                _hip.ptr(out), _hip.ptr(skip))
    return view, skip


def weight_net_hidden(xyz, hidden_units, scope, is_training, bn_decay=None, weight_decay=None, activation_fn="relu"):
    with tf_util.variable_scope(scope):
        net = xyz
        for i, num_hidden_units in enumerate(hidden_units):
            net = tf_util.conv2d(net, num_hidden_units, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                                 is_training=is_training, activation_fn=activation_fn, scope='wconv%d' % (i),
                                 bn_decay=bn_decay, weight_decay=weight_decay)
    return net


def as_attention(q, kv):
    """softmax(q k^T / sqrt(cb)) v per group of `as` neighbours (pointasnl_util.py:136-146), fused.
    q (..., as, cb), kv (..., as, 2cb) -> (..., as, cb)"""
    as_, cb = q.shape[-2], q.shape[-1]
    g = q.numel() // (as_ * cb)
    q, kv = q.contiguous(), kv.contiguous()
    out = torch.empty_like(q)
    _hip.launch("pasnl_as_attention", "as_attention", g, as_, cb, _hip.ptr(q), _hip.ptr(kv), _hip.ptr(out))
    return out


def SampleWeights(new_point, grouped_xyz, mlps, is_training, bn_decay, weight_decay, scope, bn=True, scaled=True,
                  return_logits=False):
    """Mirror of pointasnl_util.py:112-173.  new_point (B,P,S,C) and grouped_xyz (B,P,S,3) of every group -> the group's
    re-weighting (B,P,S,mlps[-1]): self-attention among the S neighbours (scaled dot product), the mlps chain, then
    a softmax over the S neighbours (the logits themselves with return_logits)."""
    if not scaled:
        raise NotImplementedError("scaled=False is never used by the reference models")
    with tf_util.variable_scope(scope):
        channel = new_point.shape[-1]
        bottleneck_channel = max(32, channel // 2)
        normalized_xyz = grouped_xyz - grouped_xyz[:, :, :1, :]
        new_point = torch.cat([normalized_xyz, new_point], dim=-1)  # (batch_size, npoint, nsample, channel+3)

        transformed_feature = tf_util.conv2d(new_point, bottleneck_channel * 2, [1, 1], padding='VALID', stride=[1, 1],
                                             bn=bn, is_training=is_training, scope='conv_kv_ds', bn_decay=bn_decay,
                                             weight_decay=weight_decay, activation_fn=None)
        transformed_new_point = tf_util.conv2d(new_point, bottleneck_channel, [1, 1], padding='VALID', stride=[1, 1],
                                               bn=bn, is_training=is_training, scope='conv_query_ds',
                                               bn_decay=bn_decay, weight_decay=weight_decay, activation_fn=None)
        # QK^T / sqrt(cb) -> softmax -> . V  : one fused kernel, K and V read in place from the conv_kv_ds output
        new_group_features = as_attention(transformed_new_point, transformed_feature)
        for i, c in enumerate(mlps):
            activation = "relu" if i < len(mlps) - 1 else None
            new_group_features = tf_util.conv2d(new_group_features, c, [1, 1], padding='VALID', stride=[1, 1], bn=bn,
                                                is_training=is_training, scope='mlp2_%d' % (i), bn_decay=bn_decay,
                                                weight_decay=weight_decay, activation_fn=activation)
        if return_logits:
            return new_group_features
        return torch.softmax(new_group_features, dim=2)  # (batch_size, npoint, nsample, mlp[-1])


def AdaptiveSampling(group_xyz, group_feature, num_neighbor, is_training, bn_decay, weight_decay, scope, bn):
    with tf_util.variable_scope(scope):
        nsample, num_channel = group_feature.shape[-2:]
        if num_neighbor == 0:
            new_xyz = group_xyz[:, :, 0, :].contiguous()
            new_feature = group_feature[:, :, 0, :].contiguous()
            return new_xyz, new_feature
        shift_group_xyz = group_xyz[:, :, :num_neighbor, :]
        shift_group_points = group_feature[:, :, :num_neighbor, :]
        logits = SampleWeights(shift_group_points, shift_group_xyz, [32, 1 + num_channel], is_training, bn_decay,
                               weight_decay, scope, bn, return_logits=True)
        # softmax over the neighbour axis + the two weighted sums (pointasnl_util.py:154,167-171) in one kernel
        # that reads the first `num_neighbor` rows of the full grouped tensors in place
        b, p = group_xyz.shape[:2]
        group_xyz, group_feature, logits = group_xyz.contiguous(), group_feature.contiguous(), logits.contiguous()
        new_xyz = torch.empty((b, p, 3), dtype=torch.float32, device=group_xyz.device)
        new_feature = torch.empty((b, p, num_channel), dtype=torch.float32, device=group_xyz.device)
        _hip.launch("pasnl_as_reweight", "as_reweight", b * p, int(num_neighbor), int(nsample), int(num_channel),
                                                _hip.ptr(logits), _hip.ptr(group_xyz), _hip.ptr(group_feature),
                                                _hip.ptr(new_xyz), _hip.ptr(new_feature))
        return new_xyz, new_feature


AS_FUSED = True  # False = the reference's op-by-op AdaptiveSampling / SampleWeights chain on gathered tensors


AS_CELL_WIDE = True    # wide layers: one kernel after the projection GEMM (pasnl_as_cell_wide)
AS_CELL_NARROW = True  # narrow layers: the whole cell after the gather in one kernel (pasnl_as_cell_narrow)
AS_PROJ_FUSED = True  # False = the projections as one vendor GEMM in front of the attention kernel (any width)


def adaptive_sampling_fused(xyz, feature, idx, num_neighbor, scope, bn, weight_decay=None):
    """AdaptiveSampling + SampleWeights (pointasnl_util.py:112-173) without the grouped tensors: one gather builds the
    (B,P,as,6+C) input of the two projections, ONE GEMM produces [K | V | Q], the micro attention reads it in place,
    and the re-weighting tail reads coordinates and features from the same gathered rows.  Same variables as the
    op-by-op chain (scopes <scope>/<scope>/conv_kv_ds, conv_query_ds, mlp2_0, mlp2_1)."""
    b, n, c = feature.shape
    _, p, k = idx.shape
    as_ = int(num_neighbor)
    channel = 3 + c
    cb = max(32, channel // 2)
    xyz, feature, idx = xyz.contiguous(), feature.contiguous(), idx.contiguous()
    narrow = AS_CELL_NARROW and 6 + c <= 15 and 1 + channel <= 16 and cb in (32, 64)
    x = torch.empty((b, p, as_, 6 + c), dtype=torch.float32, device=xyz.device)
    _hip.launch("pasnl_as_gather", "as_gather", b, n, c, p, k, as_, _hip.ptr(xyz), _hip.ptr(feature), _hip.ptr(idx), _hip.ptr(x))
    st = tf_util.store()
    with tf_util.variable_scope(scope), tf_util.variable_scope(scope):  # AdaptiveSampling's and SampleWeights' scopes
        key = st.path("kvq_fused")
        if key not in st._folded:
            with tf_util.variable_scope('conv_kv_ds'):
                wkv, bkv = st.layer(6 + c, 2 * cb, bn, weight_decay)
            with tf_util.variable_scope('conv_query_ds'):
                wq, bq = st.layer(6 + c, cb, bn, weight_decay)
            st._folded[key] = (torch.cat([wkv, wq], dim=1).contiguous(), torch.cat([bkv, bq]).contiguous())
        wkvq, bkvq = st._folded[key]
        if narrow:
            # narrow rows (the xyz-only first layers): projections, attention, mlp2, the softmax over the neighbours and the
            # re-weighted sums in ONE kernel on the gathered rows -- no (B*P*as, .) intermediate at all.  (Folding the gather
            # in as well was measured: 77 us instead of 18 + 53 -- its two dependent loads per row are exposed there.)
            with tf_util.variable_scope('mlp2_0'):
                wa, ba = st.layer(cb, 32, bn, weight_decay)
            with tf_util.variable_scope('mlp2_1'):
                wb, bb = st.layer(32, 1 + channel, bn, weight_decay)
            new_xyz = torch.empty((b, p, 3), dtype=torch.float32, device=xyz.device)
            new_feature = torch.empty((b, p, channel), dtype=torch.float32, device=xyz.device)
            _hip.launch("pasnl_as_cell_narrow", "as_cell", b * p, as_, cb, 6 + c, channel, _hip.ptr(x), _hip.ptr(wkvq),
                        _hip.ptr(bkvq), _hip.ptr(wa), _hip.ptr(ba), _hip.ptr(wb), _hip.ptr(bb), _hip.ptr(new_xyz),
                        _hip.ptr(new_feature))
            return new_xyz, new_feature
        if AS_CELL_WIDE and not narrow and cb <= 144 and 32 * (1 + channel) * 4 <= 64 * 1024:
            # wide rows: the projections are a GEMM worth running (K = 6 + c); everything after it is one kernel
            with tf_util.variable_scope('mlp2_0'):
                wa, ba = st.layer(cb, 32, bn, weight_decay)
            with tf_util.variable_scope('mlp2_1'):
                wb, bb = st.layer(32, 1 + channel, bn, weight_decay)
            # the projection's width 3 cb is odd-sized (cb = (3 + c) // 2: 195 columns at cls layer2) and the vendor GEMM
            # falls off a cliff there (135 us at N = 195, 74 us at N = 224): zero columns round it up to a multiple of 32
            ld = (3 * cb + 31) // 32 * 32 if AS_PAD_PROJECTION else 3 * cb
            if ld != 3 * cb:
                pkey = key + "@ld%d" % ld
                if pkey not in st._folded:
                    st._folded[pkey] = (torch.cat([wkvq, wkvq.new_zeros((6 + c, ld - 3 * cb))], dim=1).contiguous(),
                                        torch.cat([bkvq, bkvq.new_zeros(ld - 3 * cb)]).contiguous())
                wkvq, bkvq = st._folded[pkey]
            kvq = torch.addmm(bkvq, x.reshape(-1, 6 + c), wkvq)  # (B*P*as, ld) = [K | V | Q | -]
            new_xyz = torch.empty((b, p, 3), dtype=torch.float32, device=xyz.device)
            new_feature = torch.empty((b, p, channel), dtype=torch.float32, device=xyz.device)
            _hip.launch("pasnl_as_cell_wide_ld", "as_cell", b * p, as_, cb, 6 + c, channel, _hip.ptr(kvq), int(ld), _hip.ptr(x),
                        _hip.ptr(wa), _hip.ptr(ba), _hip.ptr(wb), _hip.ptr(bb), _hip.ptr(new_xyz), _hip.ptr(new_feature))
            return new_xyz, new_feature
        att = torch.empty((b, p, as_, cb), dtype=torch.float32, device=xyz.device)
        if AS_PROJ_FUSED and 6 + c <= 15 and cb in (32, 64):
            # narrow rows (the xyz-only first layers): K, V, Q are built inside the attention kernel; the (B*P*as, 3cb)
            # projection -- 151 MB at cls layer1, written by a GEMM with K = 9 and read once -- never exists
            _hip.launch("pasnl_as_attention_proj", "as_attention", b * p, as_, cb, 6 + c, _hip.ptr(x), _hip.ptr(wkvq),
                        _hip.ptr(bkvq), _hip.ptr(att))
        else:
            kvq = torch.addmm(bkvq, x.reshape(-1, 6 + c), wkvq)  # (B*P*as, 3cb) = [K | V | Q]
            _hip.launch("pasnl_as_attention_qkv", "as_attention", b * p, as_, cb, _hip.ptr(kvq), _hip.ptr(att))
        hid = tf_util.conv2d(att, 32, [1, 1], padding='VALID', stride=[1, 1], bn=bn, is_training=False, scope='mlp2_0',
                             weight_decay=weight_decay)
        logits = tf_util.conv2d(hid, 1 + channel, [1, 1], padding='VALID', stride=[1, 1], bn=bn, is_training=False,
                                scope='mlp2_1', activation_fn=None, weight_decay=weight_decay).contiguous()
    new_xyz = torch.empty((b, p, 3), dtype=torch.float32, device=xyz.device)
    new_feature = torch.empty((b, p, channel), dtype=torch.float32, device=xyz.device)
    _hip.launch("pasnl_as_reweight_x", "as_reweight", b * p, as_, channel, _hip.ptr(logits), _hip.ptr(x), _hip.ptr(new_xyz),
                _hip.ptr(new_feature))
    return new_xyz, new_feature


def nl_attention(q, kv, variant=None):
    """softmax(q k^T / sqrt(cb)) v with K,V = halves of kv; q (B,P,cb), kv (B,N,2cb) -> (B,P,cb).  Fused."""
    b, p, cb = q.shape
    n = kv.shape[1]
    q, kv = q.contiguous(), kv.contiguous()
    out = torch.empty_like(q)
    v = int(NL_VARIANT if variant is None else variant)
    nbytes = int(_hip.lib().pasnl_nl_attention_workspace_bytes(b, p, n, cb)) if v == 0 and NL_KEY_PARTS else 0
    if nbytes:  # few query tiles for the chip: the keys split over workgroups too, parts combined in a fixed order
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=q.device)
        _hip.launch("pasnl_nl_attention_ws", "nl_attention", b, p, n, cb, _hip.ptr(q), _hip.ptr(kv), _hip.ptr(out), v, _hip.ptr(ws),
                    ctypes.c_size_t(nbytes))
        return out
    _hip.launch("pasnl_nl_attention", "nl_attention", b, p, n, cb, _hip.ptr(q), _hip.ptr(kv), _hip.ptr(out), v)
    return out


AS_PAD_PROJECTION = True  # AdaptiveSampling, wide layers: [K | V | Q] projection padded to a multiple of 32 columns
XYZ_CONCAT = True  # False = ignore xyz_concat requests (the group_all module concatenates itself)
CENTRE0 = True  # as_neighbor == 0 layers: the fused cell reads its centres from its own tiles (pasnl_sa_cell, new_xyz = NULL)
NL_NARROW_PROJECT = True  # False = conv_kv / conv_query of narrow inputs as two vendor GEMMs


def PointNonLocalCell(feature, new_point, mlp, is_training, bn_decay, weight_decay, scope, bn=True, scaled=True,
                      mode='dot', project=True):
    """Mirror of pointasnl_util.py:175-219.  Every one of the P sampled points (new_point (B,P,1,C)) attends to ALL N points
    of the level (feature (B,N,C)) through a mlp[0]-wide bottleneck, and is projected back to mlp[-1] channels
    -> (B,P,1,mlp[-1]).  project=False stops before the projection (the caller fuses it)."""
    if mode != 'dot' or not scaled:
        raise NotImplementedError("only mode='dot', scaled=True is reachable in the reference models (SURVEY a11)")
    with tf_util.variable_scope(scope):
        bottleneck_channel = mlp[0]
        batch_size, npoint, nsample, channel = new_point.shape
        if NL_NARROW_PROJECT and feature.shape[-1] <= 16 and channel <= 16 and bottleneck_channel in (32, 64, 128):
            # coordinates in, keys|values and queries out: both projections as ONE streaming kernel (two GEMMs with K = 3 / 6
            # are bound by their own start-up); the variables are the ones conv2d creates (scopes conv_kv, conv_query)
            tf_util._require_inference(is_training)
            st = tf_util.store()
            with tf_util.variable_scope('conv_kv'):
                wkv, bkv = st.layer(feature.shape[-1], 2 * bottleneck_channel, bn, weight_decay)
            with tf_util.variable_scope('conv_query'):
                wq, bq = st.layer(channel, bottleneck_channel, bn, weight_decay)
            n_all = feature.shape[1]
            feature, new_point = feature.contiguous(), new_point.contiguous()
            kv = torch.empty((batch_size, n_all, 2 * bottleneck_channel), dtype=torch.float32, device=feature.device)
            q = torch.empty((batch_size, npoint * nsample, bottleneck_channel), dtype=torch.float32, device=feature.device)
            _hip.launch("pasnl_narrow_project2", "narrow_project", ctypes.c_long(batch_size * n_all), int(feature.shape[-1]),
                        2 * bottleneck_channel, _hip.ptr(feature), _hip.ptr(wkv), _hip.ptr(bkv), _hip.ptr(kv),
                        ctypes.c_long(batch_size * npoint * nsample), int(channel), bottleneck_channel, _hip.ptr(new_point), _hip.ptr(wq),
                        _hip.ptr(bq), _hip.ptr(q))
            new_nonlocal_point = nl_attention(q, kv)
            if not project:
                return new_nonlocal_point
            new_nonlocal_point = tf_util.conv2d(
                new_nonlocal_point.reshape(batch_size, npoint, nsample, bottleneck_channel), mlp[-1], [1, 1],
                padding='VALID', stride=[1, 1], bn=bn, is_training=is_training, scope='conv_back_project',
                bn_decay=bn_decay, weight_decay=weight_decay)
            return new_nonlocal_point.squeeze(1)
        feature = feature.unsqueeze(2)  # (batch_size, ndataset, 1, channel)
        transformed_feature = tf_util.conv2d(feature, bottleneck_channel * 2, [1, 1], padding='VALID', stride=[1, 1],
                                             bn=bn, is_training=is_training, scope='conv_kv', bn_decay=bn_decay,
                                             weight_decay=weight_decay, activation_fn=None)
        transformed_new_point = tf_util.conv2d(new_point, bottleneck_channel, [1, 1], padding='VALID', stride=[1, 1],
                                               bn=bn, is_training=is_training, scope='conv_query', bn_decay=bn_decay,
                                               weight_decay=weight_decay, activation_fn=None)
        transformed_new_point = transformed_new_point.reshape(batch_size, npoint * nsample, bottleneck_channel)
        try:
            new_nonlocal_point = nl_attention(transformed_new_point, transformed_feature.squeeze(2))
        except _hip.PasnlUnsupported:
            # bottleneck widths outside {32, 64, 128} (any C with max(32, C//2) not in that set) or operand views that are
            # not 16-byte aligned: the reference's op-by-op chain (pointasnl_util.py:196-212) on the vendor BLAS
            kv = transformed_feature.squeeze(2)
            attention_map = torch.matmul(transformed_new_point, kv[..., :bottleneck_channel].transpose(1, 2))
            attention_map = torch.softmax(attention_map / (float(bottleneck_channel) ** 0.5), dim=-1)
            new_nonlocal_point = torch.matmul(attention_map, kv[..., bottleneck_channel:])
        if not project:  # the caller fuses conv_back_project into the layer's tail (pasnl_sa_tail)
            return new_nonlocal_point
        new_nonlocal_point = tf_util.conv2d(
            new_nonlocal_point.reshape(batch_size, npoint, nsample, bottleneck_channel), mlp[-1], [1, 1],
            padding='VALID', stride=[1, 1], bn=bn, is_training=is_training, scope='conv_back_project',
            bn_decay=bn_decay, weight_decay=weight_decay)
        new_nonlocal_point = new_nonlocal_point.squeeze(1)  # (batch_size, npoints, mlp2[-1])
        return new_nonlocal_point


import os as _os
OVERLAP = _os.environ.get("PASNL_OVERLAP", "1") != "0"  # False = every kernel of a forward on the caller's stream, in program order
_SIDE_STREAMS = {}


def _side_stream(slot=0):
    key = (torch.cuda.current_device(), slot)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream()
    return _SIDE_STREAMS[key]


LAZY_FORK = False  # default of Forked(lazy=): True = a fork's side work is enqueued behind the caller's next kernel (see Forked.__init__)


class Forked:
    """Work enqueued on a side stream of the SAME forward (fork), joined by .get().

    Only hand-written kernels are forked (searches: FPS, kNN, three_nn, gathers) -- never a vendor GEMM: two concurrent
    hipBLASLt Stream-K GEMMs can dead-lock against each other.  The searches are latency-bound (FPS: one
    workgroup per cloud for hundreds of dependent rounds) and depend on coordinates only, so they run beside the MFMA / GEMM
    work of the layer before.  Works under HIP-graph capture (the side stream joins the capture through the fork event and
    rejoins at .get()) and eagerly.  Allocator safety: every fork starts by waiting for the caller's stream, so a block a
    side-stream tensor gave back is never rewritten before its last consumer, which was enqueued earlier, has run."""

    def __init__(self, fn, slot=0, lazy=None):
        self.stream = None
        self._fn = None
        if not OVERLAP:
            self.value = fn()
            return
        lazy = LAZY_FORK if lazy is None else lazy
        main = torch.cuda.current_stream()
        self._side = _side_stream(slot)
        if lazy:
            # The fork POINT is here (an event on the caller's stream), the side work is enqueued behind the caller's NEXT
            # hand-written kernel (or at .get()).  In a captured graph the successor of a node that is captured first keeps
            # that node's hardware queue and every other successor starts on another one, behind a cross-queue signal
            # (~10 us measured between a cell's end and the next kernel of the forward's own chain when the fork was
            # captured first): the forward's chain should be the one that stays.
            self._fork_ev = torch.cuda.Event()
            self._fork_ev.record(main)
            self._main = main
            self._fn = fn
            # (a weak callback: a fork that is dropped before .get() -- an exception in the forward -- must not leave its start
            # behind the next launch of a LATER forward)
            import weakref
            ref = weakref.ref(self)

            def _cb():
                me = ref()
                if me is None or me._fn is None:
                    if _cb in _hip.AFTER_LAUNCH:
                        _hip.AFTER_LAUNCH.remove(_cb)
                    return
                me._start_if_main()
            self._cb = _cb
            _hip.AFTER_LAUNCH.append(_cb)
            return
        self._side.wait_stream(main)
        self._run(fn)

    def _run(self, fn):
        side = self._side
        with torch.cuda.stream(side):
            self.value = fn()
            # the join point is THIS work's end, not whatever else is queued on the side stream by the time .get() runs
            # (a set-abstraction layer queues the next level's search behind its own small gather)
            self.done = torch.cuda.Event()
            self.done.record(side)
        self.stream = side

    def _start_if_main(self):
        if self._fn is not None and torch.cuda.current_stream() == self._main:
            self._start()

    def _start(self):
        fn, self._fn = self._fn, None
        if fn is None:
            return
        if getattr(self, "_cb", None) in _hip.AFTER_LAUNCH:
            _hip.AFTER_LAUNCH.remove(self._cb)
        # allocator safety (class docstring): the side stream must not reuse a block before its last consumer on the caller's stream
        # has run -- consumers enqueued between the fork point and this deferred start included, so the side work waits for the
        # caller's stream AS IT IS NOW, not only for the fork event.  Eagerly that is the whole ordering; under capture
        # it adds one edge from the caller's latest node, which is the node this start was deferred behind anyway.
        self._side.wait_event(self._fork_ev)
        if torch.cuda.current_stream() == self._main:
            self._side.wait_stream(self._main)
        self._run(fn)

    def get(self):
        if self._fn is not None:
            self._start()
        # every consumer stream joins (a result may be consumed by the caller's stream AND by another fork)
        if self.stream is not None and torch.cuda.current_stream() != self.stream:
            torch.cuda.current_stream().wait_event(self.done)
        return self.value


class Deferred:
    """.get() like Forked: a value put together from forked pieces on the FIRST consumer's stream, once; an event recorded
    behind the joining work makes any later consumer on ANOTHER stream wait for it (like Forked.get() does for every consumer)."""

    def __init__(self, fn):
        self._fn, self._have, self.value, self._stream, self._event = fn, False, None, None, None

    def get(self):
        if not self._have:
            self.value, self._have, self._fn = self._fn(), True, None
            if torch.cuda.is_available():
                self._stream = torch.cuda.current_stream()
                self._event = torch.cuda.Event()
                self._event.record(self._stream)
        elif self._event is not None and torch.cuda.current_stream() != self._stream:
            torch.cuda.current_stream().wait_event(self._event)
        return self.value


def _resolved(x):
    return x.get() if isinstance(x, (Forked, Deferred)) else x


def neighbor0_xyz(xyz, idx):
    """new_xyz of a layer WITHOUT adaptive sampling (as_neighbor == 0): the coordinates of neighbour 0 of every group
    (pointasnl_util.py:161-163) -- a function of coordinates and neighbour lists only, so the next level's searches can be
    started from it before the layer's features exist (PointASNLSetAbstraction computes the same values again, with the
    features, in pasnl_take_neighbor0)."""
    return _gather_rows(xyz, idx[:, :, 0].contiguous())


def _gather_index_rows(table, idx, out=None):
    """rows of an int32 table (B,N,K) at idx (B,M) -> (B,M,K): the row gather on the bit pattern.
    out: optional contiguous (B,M,K) int32 buffer to write into (inference plumbing: no gradient path)."""
    if out is None:
        return _gather_rows(table.view(torch.float32), idx).view(torch.int32)
    b, n, k = table.shape
    m = idx.shape[1]
    if tuple(out.shape) != (b, m, k) or out.dtype != torch.int32 or not out.is_contiguous() or out.device != table.device:
        raise ValueError("_gather_index_rows: out must be a contiguous (B,M,K) int32 tensor on the table's device")
    table, idx = table.contiguous(), idx.contiguous()
    _hip.launch("pasnl_group_point", "GroupPoint", b, n, k, m, 1, _hip.ptr(table), _hip.ptr(idx), _hip.ptr(out))
    return out


def sa_search(xyz, feature, npoint, nsample, use_knn=True, radius=None, knn_all=None):
    """The search prefix of a set-abstraction layer (pointasnl_util.py:236-242): farthest point sampling + the gather of the
    sampled coordinates (skipped when npoint == ndataset) and the neighbour search.  It reads coordinates only -- the
    features sampling() also gathers are dead in the reference graph: AdaptiveSampling overwrites them whenever sampling
    happened (:246-247) -- so a caller can run it ahead of / beside the dense layers.
    knn_all: the self-kNN of xyz, (B,N,K>=nsample) int32 (tensor or Forked), when the caller already has it (the seg models
    share one self-kNN per level between the encoder layer and the decoder layer of that level): the neighbour lists of
    the sampled points are then ROWS of it -- the queries are support points, distances and the (distance, index) order do
    not depend on which other queries run.  When more than half of the points are sampled the self-kNN is started here, on
    a side stream, so that it runs beside the latency-bound FPS instead of after it.
    -> (new_xyz (B,npoint,3), None, idx (B,npoint,nsample))"""
    num_points = xyz.shape[1]
    if num_points == npoint:
        if knn_all is not None:
            k_all = _resolved(knn_all)
            idx = k_all if k_all.shape[2] == nsample else k_all[:, :, :nsample].contiguous()
        elif use_knn:
            idx = knn_query(nsample, xyz, xyz)
        else:
            idx, _ = tf_grouping.query_ball_point(radius, nsample, xyz, xyz)
        return xyz, None, idx
    if use_knn and knn_all is None and OVERLAP and SELF_KNN_RATIO * npoint >= num_points:
        knn_all = Forked(lambda: knn_query(nsample, xyz, xyz), slot=1)
    fps_idx, new_xyz = tf_sampling.farthest_point_sample_gather(npoint, xyz)  # the sampler writes the sampled rows itself
    if use_knn and knn_all is not None:
        k_all = _resolved(knn_all)
        idx = _gather_index_rows(k_all, fps_idx)
        if k_all.shape[2] != nsample:
            idx = idx[:, :, :nsample].contiguous()
    elif use_knn:
        idx = knn_query(nsample, xyz, new_xyz)
    else:
        idx, _ = tf_grouping.query_ball_point(radius, nsample, xyz, new_xyz)
    return new_xyz, None, idx


def sa_search_split(xyz, npoint, nsample, knn_all, slot=0):
    """sa_search(xyz, None, npoint, nsample, knn_all=knn_all) with ONLY the sampler on a side stream: the neighbour lists (rows
    of the self-kNN `knn_all`, tensor or Forked) are gathered where the result is consumed.  A fork that waits for another
    fork inside it -- Forked(lambda: sa_search(.., knn_all=<Forked>)): the sampler's branch joins the kNN's branch before it
    returns to the forward -- made the HIP-graph executor place the sampler BEHIND the kernels of the layer it was meant to
    run beside (sem_seg_res: layer 1's 1-ms sampler started after layer 0's cell instead of at t = 0; serial forward 3.38 ->
    2.96 ms with the branches kept apart, profiles/r04_q_fork_variants.txt).  -> Deferred of (new_xyz, None, idx)"""
    fp = Forked(lambda: tf_sampling.farthest_point_sample_gather(npoint, xyz), slot=slot)

    def join():
        fps_idx, new_xyz = fp.get()
        k_all = _resolved(knn_all)
        idx = _gather_index_rows(k_all, fps_idx)
        if k_all.shape[2] != nsample:
            idx = idx[:, :, :nsample].contiguous()
        return new_xyz, None, idx
    return Deferred(join)


def _tail_packed(st, w):
    """`w` (k, c) in pasnl_sa_tail's operand order, packed once per variable (cached next to the folded weights)."""
    key = "@tailpk:%x" % w.data_ptr()
    if key not in st._folded:
        wc = w.contiguous()
        pk = torch.empty(int(_hip.lib().pasnl_sa_tail_packed_weights_bytes(wc.shape[0], wc.shape[1])) // 4, dtype=torch.float32,
                         device=w.device)
        _hip.launch("pasnl_sa_tail_pack_weights", "sa_tail_pack", int(wc.shape[0]), int(wc.shape[1]), _hip.ptr(wc), _hip.ptr(pk))
        st._folded[key] = (pk, w)  # (keeps `w` alive: the pointer in the key stays unique)
    return st._folded[key][0]


def PointASNLSetAbstraction(xyz, feature, npoint, nsample, mlp, is_training, bn_decay, weight_decay, scope, bn=True,
                            use_knn=True, radius=None, as_neighbor=8, NL=True, search=None, after_sampling=None,
                            xyz_concat=False, after_cell=None, before_after_conv=None, residual=None):
    '''Mirror of pointasnl_util.py:221-292: one PointASNL set-abstraction layer.
        xyz (B,N,3), feature (B,N,C)  ->  new_xyz (B,npoint,3), new_points (B,npoint,mlp[-1])
    npoint points are sampled (FPS) and moved by AdaptiveSampling over their first `as_neighbor` neighbours; each keeps
    `nsample` neighbours, which feed the local cell (mlp[:-1], weight net, after_conv); skip connection, the optional
    Point-NonLocal cell over the whole level, and the aggregation layer follow.
    search: (new_xyz, None, idx) of sa_search() on the same coordinates (tuple or Forked), computed ahead by the caller;
    after_sampling: callback(new_xyz) the moment the level's coordinates are final (the caller forks the next searches).
    xyz_concat: the caller will feed tf.concat([new_xyz, new_points]) to a group_all module (pointasnl_cls layer3_x): the
    layer's last kernel then writes those rows as well, and the returned new_points carries them as `.xyz_concat`
    = (new_xyz, (B,npoint,4+C) table [0 | xyz | points]) for pointnet_util.sample_and_group_all to pick up.
    residual: (B,npoint,mlp[-1]) added to the layer's output -- the `_res` model's "l1_2_points + l1_1_points"
    (pointasnl_sem_seg_res.py:37,42,47,52) in the epilogue of the layer's last kernel instead of a pass of its own.'''
    with tf_util.variable_scope(scope):
        batch_size, num_points, num_channel = feature.shape
        # Farthest point sampling + neighbour search (the reference's sampling() / grouping(): pointasnl_util.py:236-242);
        # `search` = the result of sa_search() on the same coordinates, computed ahead / on a side stream by the caller
        new_xyz, _, idx = _resolved(search) if search is not None else sa_search(xyz, feature, npoint, nsample, use_knn, radius)
        new_feature = feature  # npoint == ndataset (:237-239); otherwise AdaptiveSampling defines it below (:246-247)
        nl_channel = mlp[-1]

        # ---- adaptive sampling (:244-247)
        fused = _local_cell_supported(6 + num_channel, mlp, nsample)
        centre0 = False  # True: the layer's centres are its groups' neighbour 0 and the fused cell produces them
        if num_points != npoint and as_neighbor == 0:
            # AdaptiveSampling with num_neighbor == 0 takes neighbour 0 of every group (:161-164): one gather kernel
            xyz, feature, idx = xyz.contiguous(), feature.contiguous(), idx.contiguous()

            def take0():
                nx = torch.empty((batch_size, npoint, 3), dtype=torch.float32, device=xyz.device)
                nf = torch.empty((batch_size, npoint, 3 + num_channel), dtype=torch.float32, device=xyz.device)
                _hip.launch("pasnl_take_neighbor0", "take_neighbor0", batch_size, num_points, num_channel, npoint, nsample,
                            _hip.ptr(xyz), _hip.ptr(feature), _hip.ptr(idx), _hip.ptr(nx), _hip.ptr(nf))
                return nx, nf
            # the fused cell finds its centres in its own tiles and writes new_xyz / new_feature itself: no gather launch
            # between the search and the cell (pasnl_sa_cell_centre0)
            # (rows wider than 128 channels: only the wide kernels -- mlp[0] = 256 / 512 -- write the neighbour-0 row themselves)
            centre0 = (CENTRE0 and fused and SA_CELL_GATHER and npoint <= num_points
                       and (num_channel <= 128 or (SA_CELL_WIDE and mlp[0] in (256, 512) and nsample == 32)))
            if not centre0:
                new_xyz, new_feature = take0()
        elif num_points != npoint and AS_FUSED and as_neighbor <= 16:
            tf_util._require_inference(is_training)
            new_xyz, new_feature = adaptive_sampling_fused(xyz, feature, idx, as_neighbor, scope, bn, weight_decay)
        elif num_points != npoint:
            # AdaptiveSampling only ever reads the first `as_neighbor` neighbours (:165-166): gather just those
            # instead of slicing the full grouped tensors
            k_as = max(1, as_neighbor)
            idx_as = idx[:, :, :k_as].contiguous()
            g_xyz = tf_grouping.group_point(xyz, idx_as)
            g_pts = torch.cat([g_xyz, tf_grouping.group_point(feature, idx_as)], dim=-1)
            new_xyz, new_feature = AdaptiveSampling(g_xyz, g_pts, as_neighbor, is_training, bn_decay, weight_decay,
                                                    scope, bn)
        if after_sampling is not None and not centre0:
            after_sampling(new_xyz)  # the sampled (and shifted) coordinates are final: the caller may start the next search
        new_point = None
        if fused and SA_CELL_GATHER:
            # grouping + skip max + local cell: one MFMA kernel reading the tables in place
            tf_util._require_inference(is_training)
            try:
                if centre0:
                    new_point, skip_spatial, new_xyz, new_feature = sa_cell(xyz, feature, idx, None, mlp, is_training,
                                                                            bn_decay, weight_decay, bn)
                else:
                    new_point, skip_spatial = sa_cell(xyz, feature, idx, new_xyz, mlp, is_training, bn_decay, weight_decay, bn)
            except _hip.PasnlUnsupported:  # e.g. a row too wide for the LDS-resident weights: two-kernel / op-by-op path
                new_point = None
        if centre0:
            if new_point is None:
                new_xyz, new_feature = take0()
            if after_sampling is not None:
                after_sampling(new_xyz)
        if new_point is None:
            # gather + translation normalisation + both concats + the skip connection's reduce_max: one kernel
            new_point, skip_spatial = sa_group(xyz, feature, idx, new_xyz)
            grouped_xyz = new_point[..., 0:3]
            if fused and not (len(mlp) == 3 and mlp[0] >= 32):
                fused = False  # the two-kernel form exists for the plain [c, c, out] shape only
            if fused:
                tf_util._require_inference(is_training)
                try:
                    new_point = sa_local_cell(new_point, mlp, is_training, bn_decay, weight_decay, bn)
                except _hip.PasnlUnsupported:
                    fused = False
            if not fused:
                for i, num_out_channel in enumerate(mlp):
                    if i != len(mlp) - 1:
                        new_point = tf_util.conv2d(new_point, num_out_channel, [1, 1], padding='VALID', stride=[1, 1],
                                                   bn=bn, is_training=is_training, scope='conv%d' % (i),
                                                   bn_decay=bn_decay, weight_decay=weight_decay)
                weight = weight_net_hidden(grouped_xyz, [32], scope='weight_net', is_training=is_training,
                                           bn_decay=bn_decay, weight_decay=weight_decay)
                new_point = new_point.transpose(2, 3)
                new_point = torch.matmul(new_point, weight)

        if after_cell is not None:
            after_cell()  # (hooks for a serving loop: where the next batch's search prefix is forked, bench.py --pipeline prefetch)
        c_out = mlp[-1]
        # (layers with few groups -- the deep, wide ones: 512 rows x 512 channels -- are 16 workgroups of dependent MFMA
        # chains fed from L2; there the three small vendor GEMMs are faster: measured 101 vs ~45 us)
        if SA_TAIL_FUSED and c_out % 32 == 0 and c_out <= 512 and batch_size * npoint >= SA_TAIL_MIN_ROWS:
            # skip convolution + back-projection of the non-local cell + both adds + aggregation: ONE kernel behind the
            # after_conv GEMM (pasnl_sa_tail) instead of three small GEMMs and two element-wise passes
            tf_util._require_inference(is_training)
            cb = max(32, num_channel // 2)
            att = None
            if NL:
                att = PointNonLocalCell(feature, new_feature.unsqueeze(1), [cb, nl_channel], is_training, bn_decay,
                                        weight_decay, scope, bn, project=False)  # (B, P, cb)
            if before_after_conv is not None:
                before_after_conv()
            after = tf_util.conv2d(new_point, c_out, [1, new_point.shape[2]], padding='VALID', stride=[1, 1], bn=bn,
                                   is_training=is_training, scope='after_conv', bn_decay=bn_decay, weight_decay=weight_decay)
            st = tf_util.store()
            w_in = skip_spatial.shape[-1]
            with tf_util.variable_scope('skip'):
                ws, bs = st.layer(w_in, c_out, bn, weight_decay)
            if NL:
                with tf_util.variable_scope(scope), tf_util.variable_scope('conv_back_project'):
                    wb, bb = st.layer(cb, c_out, bn, weight_decay)
            with tf_util.variable_scope('aggregation'):
                wagg, bagg = st.layer(c_out, c_out, bn, weight_decay)
            rows = batch_size * npoint
            after, skip_spatial = after.contiguous(), skip_spatial.contiguous()
            att = att.contiguous() if NL else None  # bound to a name: the buffer has to outlive the launch
            out = torch.empty((batch_size, npoint, c_out), dtype=torch.float32, device=xyz.device)
            args = (rows, int(w_in), int(cb if NL else 0), int(c_out), _hip.ptr(after), _hip.ptr(skip_spatial),
                    _hip.ptr(att), _hip.ptr(ws), _hip.ptr(bs), _hip.ptr(wb if NL else None),
                    _hip.ptr(bb if NL else None), _hip.ptr(wagg), _hip.ptr(bagg), _hip.ptr(out))
            try:
                if SA_TAIL_PACKED:
                    # the three matrices in the matrix instruction's operand order, packed once per variable
                    pk = [_tail_packed(st, m) if m is not None else None for m in (ws, wb if NL else None, wagg)]
                    cat = None
                    if xyz_concat and XYZ_CONCAT:
                        new_xyz = new_xyz.contiguous()
                        cat = torch.empty((batch_size, npoint, 4 + c_out), dtype=torch.float32, device=xyz.device)
                    res_c = residual.contiguous() if residual is not None else None
                    _hip.launch("pasnl_sa_tail_packed", "sa_tail", *args[:7], _hip.ptr(pk[0]), args[8], _hip.ptr(pk[1]), args[10],
                                _hip.ptr(pk[2]), args[12], _hip.ptr(res_c), _hip.ptr(new_xyz if cat is not None else None),
                                _hip.ptr(cat), args[13])
                    residual = None  # (added)
                    if cat is not None:
                        out.xyz_concat = (new_xyz, cat)
                elif residual is not None:
                    residual = residual.contiguous()
                    _hip.launch("pasnl_sa_tail_res", "sa_tail", *args[:-1], _hip.ptr(residual), args[-1])
                    residual = None  # (added)
                elif xyz_concat and XYZ_CONCAT:
                    new_xyz = new_xyz.contiguous()
                    cat = torch.empty((batch_size, npoint, 4 + c_out), dtype=torch.float32, device=xyz.device)
                    _hip.launch("pasnl_sa_tail_cat", "sa_tail", *args, _hip.ptr(new_xyz), _hip.ptr(cat))
                    out.xyz_concat = (new_xyz, cat)
                else:
                    _hip.launch("pasnl_sa_tail", "sa_tail", *args)
            except _hip.PasnlUnsupported:
                # weights too wide for the kernel's LDS tile (e.g. c_out = 512 with ~480 input channels): the same tail op by
                # op on the vendor BLAS, from the pieces above -- out = relu((after + relu(skip) + relu(back_project)) . Wagg)
                tail = after.reshape(rows, c_out) + torch.relu_(torch.addmm(bs, skip_spatial.reshape(rows, w_in), ws))
                if NL:
                    tail = tail + torch.relu_(torch.addmm(bb, att.reshape(rows, cb), wb))
                out = torch.relu_(torch.addmm(bagg, tail, wagg)).reshape(batch_size, npoint, c_out)
            return new_xyz, (out if residual is None else out + residual)

        # ---- non-local cell (:251-255)
        if NL:
            new_nonlocal_point = PointNonLocalCell(feature, new_feature.unsqueeze(1),
                                                   [max(32, num_channel // 2), nl_channel], is_training, bn_decay,
                                                   weight_decay, scope, bn)

        # ---- skip connection (:257-261)
        skip_spatial = tf_util.conv1d(skip_spatial, mlp[-1], 1, padding='VALID', stride=1, bn=bn,
                                      is_training=is_training, scope='skip', bn_decay=bn_decay,
                                      weight_decay=weight_decay)

        new_point = tf_util.conv2d(new_point, mlp[-1], [1, new_point.shape[2]], padding='VALID', stride=[1, 1], bn=bn,
                                   is_training=is_training, scope='after_conv', bn_decay=bn_decay,
                                   weight_decay=weight_decay)
        new_point = new_point.squeeze(2)  # (batch_size, npoints, mlp2[-1])
        new_point = new_point + skip_spatial
        if NL:
            new_point = new_point + new_nonlocal_point

        # ---- aggregation (:287-290)
        new_point = tf_util.conv1d(new_point, mlp[-1], 1, padding='VALID', stride=1, bn=bn, is_training=is_training,
                                   scope='aggregation', bn_decay=bn_decay, weight_decay=weight_decay)
        return new_xyz, (new_point if residual is None else new_point + residual)


DECODE_CELL_FUSED = True  # False = the reference's op-by-op chain (gathers, concat, conv2d, transpose, batched matmul)


DECODE_CELL_TILED = True  # the decoder cell writes its output in the tiled order (pasnl_decode_cell_tiled) when it can
_DECODE_PERM = {}


def decode_tiled_order(c, v4, device):
    """(3+c)*32 int64: entry q = the reference position ch*32 + j of what pasnl_decode_cell_tiled stores at position q of a
    point (include/pasnl.h); cached per (c, v4, device).  decode_after_conv's weight rows are gathered with it."""
    key = (c, bool(v4), str(device))
    if key not in _DECODE_PERM:
        V = 4 if v4 else 1
        T = torch.arange(c // 32).view(-1, 1, 1, 1, 1)
        g = torch.arange(4).view(1, -1, 1, 1, 1)
        h = torch.arange(2).view(1, 1, -1, 1, 1)
        m = torch.arange(32).view(1, 1, 1, -1, 1)
        i = torch.arange(4).view(1, 1, 1, 1, -1)
        ch = 3 + 32 * V * (T // V) + V * m + T % V
        j = 8 * g + 4 * h + i
        tiles = (ch * 32 + j).reshape(-1)  # order (T, g, h, m, i) == 96 + T*1024 + (2g+h)*128 + 4m + i
        _DECODE_PERM[key] = torch.cat([torch.arange(96), tiles]).to(device)
    return _DECODE_PERM[key]


def decode_cell(xyz, feature, idx, weight_decay=None):
    """Decoder local cell (pointasnl_util.py:323-331) fused: (B,N,3), (B,N,C), (B,N,K) -> (B,N,3+C,32).
    The variables are the ones the op-by-op chain creates (scope decode_weight_net/wconv0)."""
    b, n, c = feature.shape
    k = idx.shape[2]
    with tf_util.variable_scope('decode_weight_net'), tf_util.variable_scope('wconv0'):
        ww, bw = tf_util.store().layer(3, 32, True, weight_decay)
    xyz, feature, idx = xyz.contiguous(), feature.contiguous(), idx.contiguous()
    out = torch.empty((b, n, 3 + c, 32), dtype=torch.float32, device=xyz.device)
    if DECODE_CELL_TILED and k == 16 and c % 32 == 0:
        # the same values in an order that is cheap to write (4 x 1 KiB per tile) -- its only consumer, decode_after_conv,
        # contracts all of them and gets its weight rows in the same order (`row_order` rides on the tensor)
        v4 = int(_hip.lib().pasnl_decode_cell_tiled_v4(c, _hip.ptr(feature)))
        _hip.launch("pasnl_decode_cell_tiled", "decode_cell", b, n, c, k, _hip.ptr(xyz), _hip.ptr(feature), _hip.ptr(idx),
                    _hip.ptr(ww), _hip.ptr(bw), _hip.ptr(out))
        out.row_order = ("decode_tiled_c%d_v%d" % (c, v4), decode_tiled_order(c, v4, xyz.device))
        return out
    _hip.launch("pasnl_decode_cell", "decode_cell", b, n, c, k, _hip.ptr(xyz), _hip.ptr(feature), _hip.ptr(idx), _hip.ptr(ww),
                _hip.ptr(bw), _hip.ptr(out))
    return out


def PointASNLDecodingLayer(xyz1, xyz2, points1, points2, nsample, mlp, is_training, bn_decay, weight_decay, scope,
                           bn=True, use_xyz=True, use_knn=True, radius=None, dilate_rate=1, mode='concat', NL=False,
                           nn=None, knn_all=None):
    '''Mirror of pointasnl_util.py:294-345: one PointASNL up-sampling layer.
        dense level xyz1 (B,N1,3) / points1 (B,N1,C1), coarse level xyz2 (B,N2,3) / points2 (B,N2,C2) -> (B,N1,mlp[-1])
    points2 is carried to the dense level by inverse-distance interpolation over 3 neighbours, the dense level's own
    local cell over `nsample` neighbours is added, and the mlp chain runs on the sum joined with points1.'''
    if NL:
        raise NotImplementedError("NL=True in the decoder selects mode='concat', never used by the models (SURVEY a11)")
    with tf_util.variable_scope(scope):
        # nn = three_nn(xyz1, xyz2), knn_all = the self-kNN of xyz1 (B,N1,K>=nsample): both read coordinates only and may
        # have been computed ahead by the caller (tuple / tensor / Forked); the encoder layer of this level shares knn_all
        dist, idx = three_nn(xyz1, xyz2) if nn is None else _resolved(nn)
        if FP_HEAD_FUSED and not is_training and not torch.is_grad_enabled():
            interpolated_points = fp_interpolate_cat(points2, idx, dist)  # weights + interpolation in one launch, the same bits
        else:
            weight = three_weights(dist)  # pointasnl_util.py:308-311 as one kernel
            interpolated_points = three_interpolate(points2, idx, weight)

        # ---- local cell (:322-331)
        if DECODE_CELL_FUSED and use_xyz and use_knn and nsample in (16, 32):
            # self-kNN, both gathers, the centring, the weight net and F^T.G in one kernel (no grouped tensors)
            tf_util._require_inference(is_training)
            if knn_all is None:
                kidx = knn_query(nsample, xyz1, xyz1)
            else:  # ascending (distance, index): the first nsample columns of a wider self-kNN ARE the nsample-NN
                kidx = _resolved(knn_all)
                kidx = kidx if kidx.shape[2] == nsample else kidx[:, :, :nsample].contiguous()
            new_points = decode_cell(xyz1, interpolated_points, kidx, weight_decay)
        else:
            grouped_xyz, grouped_feature, idx = grouping(interpolated_points, nsample, xyz1, xyz1, use_xyz=use_xyz,
                                                         use_knn=use_knn, radius=radius)
            grouped_xyz = grouped_xyz - xyz1.unsqueeze(2)  # translation normalization
            weight = weight_net_hidden(grouped_xyz, [32], scope='decode_weight_net', is_training=is_training,
                                       bn_decay=bn_decay, weight_decay=weight_decay)
            new_points = grouped_feature.transpose(2, 3)
            new_points = torch.matmul(new_points, weight)
        new_points = tf_util.conv2d(new_points, mlp[0], [1, new_points.shape[2]], padding='VALID', stride=[1, 1], bn=bn,
                                    is_training=is_training, scope='decode_after_conv', bn_decay=bn_decay,
                                    weight_decay=weight_decay, row_order=getattr(new_points, "row_order", None))
        if points1 is not None:
            new_points1 = torch.cat([new_points, points1.unsqueeze(2)], dim=-1)
        else:
            new_points1 = new_points
        for i, num_out_channel in enumerate(mlp):
            if i != 0:
                new_points1 = tf_util.conv2d(new_points1, num_out_channel, [1, 1], padding='VALID', stride=[1, 1],
                                             bn=bn, is_training=is_training, scope='conv_%d' % (i), bn_decay=bn_decay,
                                             weight_decay=weight_decay)
        new_points = new_points1.squeeze(2)  # B,ndataset1,mlp[-1]
        return new_points


def get_repulsion_loss(pred, nsample=20, radius=0.07):
    """pointasnl_util.py:361-378; the only in-model consumer of query_ball_point / group_point."""
    idx, pts_cnt = tf_grouping.query_ball_point(radius, nsample, pred, pred)
    grouped_pred = tf_grouping.group_point(pred, idx)  # (batch_size, npoint, nsample, 3)
    grouped_pred = grouped_pred - pred.unsqueeze(2)
    h = 0.03
    dist_square = (grouped_pred ** 2).sum(dim=-1)
    dist_square, _ = torch.topk(-dist_square, 5)
    dist_square = -dist_square[:, :, 1:]  # remove the first one
    dist_square = torch.clamp(dist_square, min=1e-12)
    dist = torch.sqrt(dist_square)
    weight = torch.exp(-dist_square / h ** 2)
    uniform_loss = torch.mean(radius - dist * weight)
    return uniform_loss
