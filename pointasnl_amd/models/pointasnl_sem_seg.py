"""pointasnl_sem_seg -- inference graph of the reference's ScanNet segmentation model
(models/pointasnl_sem_seg.py:18-50): 4 PointASNL set-abstraction layers + 4 PointASNL decoding layers
(three_nn / three_interpolate + self-kNN local cell).  Same get_model signature; torch device tensors."""
import torch
from pointasnl_amd.utils import tf_util
from pointasnl_amd.utils.pointasnl_util import (PointASNLSetAbstraction, PointASNLDecodingLayer, get_repulsion_loss, Forked,
                                                sa_search, sa_search_split, knn_query, neighbor0_xyz)
from pointasnl_amd.tf_interpolate import three_nn


def first_layer(num_point):
    """sa_search() arguments of layer1 (for callers that run the search ahead of the rest of the forward)"""
    return dict(npoint=num_point // 8, nsample=32)


def get_model(point_cloud, is_training, num_class, bn_decay=None, weight_decay=None, feature_channel=0, search=None, before_head=None):
    """ Semantic segmentation PointNet, input is B x N x (3+feature_channel), output B x N x num_class """
    end_points = {}
    num_point = point_cloud.shape[1]
    if feature_channel > 0:
        l0_xyz = point_cloud[:, :, 0:3].contiguous()
        l0_points = point_cloud[:, :, 3:3 + feature_channel].contiguous()
    else:
        l0_xyz = point_cloud
        l0_points = point_cloud
    end_points['l0_xyz'] = l0_xyz
    num_points = [num_point // 8, num_point // 32, num_point // 128, num_point // 256]
    kw = dict(is_training=is_training, bn_decay=bn_decay, weight_decay=weight_decay)
    # ---- searches.  Every FPS / kNN / three_nn of the graph reads coordinates only, and the coordinates of level k are
    # final right after layer k's AdaptiveSampling -- long before its features are.  So each level's searches are forked
    # onto side streams at that moment (pointasnl_util.Forked: hand-written kernels only) and run beside the MFMA / GEMM work:
    #   * ONE self-kNN (K = 32) per level serves the encoder layer that samples from the level (its neighbour lists are the
    #     rows of the sampled points) AND the decoder layer of the level (K = 16: the first 16 columns);
    #   * level 0's self-kNN runs beside layer1's FPS (num_point/8 dependent rounds on one CU per cloud);
    #   * three_nn of decoder k starts as soon as both of its levels exist; levels 3 and 4 (as_neighbor = 0: the sampled
    #     coordinates are neighbour 0's) are searched ahead of layer2's dense part.
    knn, nn, srch = {}, {}, {}
    knn[0] = Forked(lambda: knn_query(32, l0_xyz, l0_xyz), slot=1)
    srch[1] = search if search is not None else sa_search(l0_xyz, None, num_points[0], 32, knn_all=knn[0])

    def level1(xyz1):  # l1_xyz final
        knn[1] = Forked(lambda: knn_query(32, xyz1, xyz1), slot=1)
        srch[2] = sa_search_split(xyz1, num_points[1], 32, knn[1], slot=0)  # sampler alone on its side stream; rows joined at use

    def level2(xyz2):  # l2_xyz final: everything below it depends on coordinates only
        def chain():
            k2 = knn_query(32, xyz2, xyz2)
            s3 = sa_search(xyz2, None, num_points[2], 32, knn_all=k2)
            xyz3 = neighbor0_xyz(xyz2, s3[2])
            k3 = knn_query(32, xyz3, xyz3)
            s4 = sa_search(xyz3, None, num_points[3], 32, knn_all=k3)
            xyz4 = neighbor0_xyz(xyz3, s4[2])
            return dict(k2=k2, s3=s3, k3=k3, s4=s4, n1=three_nn(xyz3, xyz4), n2=three_nn(xyz2, xyz3))
        srch["deep"] = Forked(chain, slot=0)
        # needed by the decoders at the very end: queued BEHIND the urgent searches (side streams share hardware queues)
        nn[3] = Forked(lambda: three_nn(l1_xyz_box[0], xyz2), slot=0)
        nn[4] = Forked(lambda: three_nn(l0_xyz, l1_xyz_box[0]), slot=0)

    l1_xyz_box = []
    # Feature encoding layers
    l1_xyz, l1_points = PointASNLSetAbstraction(l0_xyz, l0_points, npoint=num_points[0], nsample=32, mlp=[32, 32, 64],
                                                scope='layer1', as_neighbor=8, search=srch[1],
                                                after_sampling=lambda x: (l1_xyz_box.append(x), level1(x)), **kw)
    l2_xyz, l2_points = PointASNLSetAbstraction(l1_xyz, l1_points, npoint=num_points[1], nsample=32, mlp=[64, 64, 128],
                                                scope='layer2', as_neighbor=4, search=srch[2], after_sampling=level2, **kw)
    if before_head is not None:  # a serving loop forks the next batch's search prefix here, beside the deep layers and the
        before_head()            # decoder (long chains of small kernels that leave most of the GPU idle); bench.py --pipeline prefetch
    deep = srch["deep"].get()
    l3_xyz, l3_points = PointASNLSetAbstraction(l2_xyz, l2_points, npoint=num_points[2], nsample=32, mlp=[128, 128, 256],
                                                scope='layer3', as_neighbor=0, search=deep["s3"], **kw)
    l4_xyz, l4_points = PointASNLSetAbstraction(l3_xyz, l3_points, npoint=num_points[3], nsample=32, mlp=[256, 256, 512],
                                                scope='layer4', as_neighbor=0, search=deep["s4"], **kw)
    end_points['l1_xyz'] = l1_xyz
    # Feature decoding layers
    l3_points = PointASNLDecodingLayer(l3_xyz, l4_xyz, l3_points, l4_points, 16, [512, 512], is_training, bn_decay,
                                       weight_decay, scope='fa_layer1', nn=deep["n1"], knn_all=deep["k3"])
    l2_points = PointASNLDecodingLayer(l2_xyz, l3_xyz, l2_points, l3_points, 16, [256, 256], is_training, bn_decay,
                                       weight_decay, scope='fa_layer2', nn=deep["n2"], knn_all=deep["k2"])
    l1_points = PointASNLDecodingLayer(l1_xyz, l2_xyz, l1_points, l2_points, 16, [256, 128], is_training, bn_decay,
                                       weight_decay, scope='fa_layer3', nn=nn[3], knn_all=knn[1])
    l0_points = PointASNLDecodingLayer(l0_xyz, l1_xyz, l0_points, l1_points, 16, [128, 128, 128], is_training, bn_decay,
                                       weight_decay, scope='fa_layer4', nn=nn[4], knn_all=knn[0])
    # FC layers
    net = tf_util.conv1d(l0_points, 128, 1, padding='VALID', bn=True, is_training=is_training, scope='fc1',
                         bn_decay=bn_decay, weight_decay=weight_decay)
    end_points['feats'] = net
    net = tf_util.dropout(net, keep_prob=0.5, is_training=is_training, scope='dp1')
    net = tf_util.conv1d(net, num_class, 1, padding='VALID', activation_fn=None, weight_decay=weight_decay, scope='fc2')
    return net, end_points


def get_loss(pred, label, end_points, smpw=1.0, uniform_weight=0.01, weights_decay=1e-4, radius=0.07):
    """ pred: BxNxC, label: BxN, smpw: BxN  (pointasnl_sem_seg.py get_loss) """
    regularization_loss = tf_util.regularization_loss(weights_decay)
    # tf.losses.sparse_softmax_cross_entropy(weights=smpw): sum(w * ce) / number of non-zero weights -- and the result
    # joins the 'losses' collection (tf.GraphKeys.LOSSES), the one tf.add_n(tf.get_collection('losses')) sums below: the
    # reference's total therefore holds the classify loss TWICE (checked against the reference's Python executed under
    # oracle/tf_shim: tests/golden/ref_losses.npz)
    ce = torch.nn.functional.cross_entropy(pred.reshape(-1, pred.shape[-1]), label.reshape(-1).long(), reduction='none')
    w = torch.as_tensor(smpw, dtype=ce.dtype, device=ce.device).expand(label.shape).reshape(-1)
    classify_loss = (ce * w).sum() / torch.count_nonzero(w).clamp(min=1)
    uniform_loss = get_repulsion_loss(end_points['l1_xyz'], nsample=20, radius=radius)
    weight_reg = tf_util.collection_losses(extra=[classify_loss])
    return classify_loss + weight_reg + uniform_weight * uniform_loss + regularization_loss
