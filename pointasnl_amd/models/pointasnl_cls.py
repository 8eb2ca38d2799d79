"""pointasnl_cls -- inference graph of the reference's ModelNet40 classifier (models/pointasnl_cls.py:17-52)
on the MI355X hot path.  ``get_model`` keeps the reference signature; tensors are torch CUDA tensors and the
weights live in the active tf_util.VariableStore (seeded; there are no checkpoints offline).
"""
import torch

from pointasnl_amd.utils import tf_util
from pointasnl_amd.utils.pointnet_util import pointnet_sa_module
from pointasnl_amd.utils.pointasnl_util import PointASNLSetAbstraction, get_repulsion_loss, sa_search, Forked


def first_layer(num_point=None):
    """sa_search() arguments of layer1 (for callers that run the search ahead of the rest of the forward)"""
    return dict(npoint=512, nsample=32)


def get_model(point_cloud, is_training=False, use_normal=False, bn_decay=None, weight_decay=None, num_class=40,
              adaptive_sample=False, search=None, before_head=None, fork_at="head", lazy_fork=None):
    """ Classification PointNet, input is BxNx3 (BxNx6 with normals), output Bx40
    search / before_head (not in the reference signature; both optional): the search prefix of layer1 computed ahead by the
    caller, and a callback invoked once the set-abstraction layers are enqueued -- a serving loop forks the NEXT batch's
    search prefix there, beside the classifier head, which leaves most of the GPU idle (bench.py --pipeline prefetch).
    fork_at: where that callback is invoked -- "head" (behind layer 2), "conv2" (before layer 2's after_conv GEMM) or "cell2"
    (behind layer 2's cell): the prefix is ~0.3 ms of dependent rounds and has to END with the head.
    lazy_fork: layer 2's search is enqueued behind the forward's next kernel instead of at the fork point (Forked(lazy=True):
    the forward's chain keeps its hardware queue in a captured graph); default: without adaptive sampling. """
    if lazy_fork is None:
        lazy_fork = not adaptive_sample
    batch_size = point_cloud.shape[0]
    end_points = {}
    if use_normal:
        l0_xyz = point_cloud[:, :, 0:3].contiguous()
        l0_points = point_cloud[:, :, 3:6].contiguous()
    else:
        l0_xyz = point_cloud
        l0_points = point_cloud
    end_points['l0_xyz'] = l0_xyz
    as_neighbor = [12, 12] if adaptive_sample else [0, 0]

    # Set abstraction layers.  Layer 2's search (FPS + kNN) reads layer 1's sampled coordinates only, so it is forked onto
    # a side stream the moment those are final and runs beside layer 1's MFMA / GEMM work (pointasnl_util.Forked)
    search2 = []
    if isinstance(search, dict):  # {1: layer1's search, 2: layer2's}: without adaptive sampling layer 2's coordinates are layer 1's sampled
        search2.append(search[2])  # input points -- its search, too, reads the input cloud alone and can be computed ahead
        search = search[1]
    l1_xyz, l1_points = PointASNLSetAbstraction(l0_xyz, l0_points, npoint=512, nsample=32, mlp=[64, 64, 128],
                                                is_training=is_training, bn_decay=bn_decay, weight_decay=weight_decay,
                                                scope='layer1', as_neighbor=as_neighbor[0], search=search, xyz_concat=True,
                                                after_sampling=None if search2 else lambda xyz1: search2.append(
                                                    Forked(lambda: sa_search(xyz1, None, 128, 64), lazy=lazy_fork)))
    end_points['l1_xyz'] = l1_xyz
    l2_xyz, l2_points = PointASNLSetAbstraction(l1_xyz, l1_points, npoint=128, nsample=64, mlp=[128, 128, 256],
                                                is_training=is_training, bn_decay=bn_decay, weight_decay=weight_decay,
                                                scope='layer2', as_neighbor=as_neighbor[1], search=search2[0], xyz_concat=True,
                                                after_cell=before_head if fork_at == "cell2" else None,
                                                before_after_conv=before_head if fork_at == "conv2" else None)
    end_points['l2_xyz'] = l1_xyz  # sic: the reference stores l1_xyz here (pointasnl_cls.py:38)
    if before_head is not None and fork_at == "head":
        before_head()
    # the two pooled vectors are written side by side into fc1's input: tf.concat([l3_points, l3_points_res]) for free
    net = torch.empty((batch_size, 1024 + 512), dtype=torch.float32, device=point_cloud.device)
    _, l3_points_res, _ = pointnet_sa_module(l1_xyz, l1_points, npoint=None, radius=None, nsample=None,
                                             mlp=[128, 256, 512], mlp2=None, group_all=True, is_training=is_training,
                                             bn_decay=bn_decay, scope='layer3_1', pooled_out=net[:, 1024:])
    _, l3_points, _ = pointnet_sa_module(l2_xyz, l2_points, npoint=None, radius=None, nsample=None,
                                         mlp=[256, 512, 1024], mlp2=None, group_all=True, is_training=is_training,
                                         bn_decay=bn_decay, scope='layer3_2', pooled_out=net[:, :1024])

    # Fully connected layers
    net = tf_util.fully_connected(net, 512, bn=True, is_training=is_training, scope='fc1', bn_decay=bn_decay)
    net = tf_util.dropout(net, keep_prob=0.4, is_training=is_training, scope='dp1')
    net = tf_util.fully_connected(net, 256, bn=True, is_training=is_training, scope='fc2', bn_decay=bn_decay)
    net = tf_util.dropout(net, keep_prob=0.4, is_training=is_training, scope='dp2')
    net = tf_util.fully_connected(net, num_class, activation_fn=None, scope='fc3')
    end_points['l1_points'] = l1_points
    end_points['l2_points'] = l2_points
    end_points['l2_xyz_true'] = l2_xyz
    return net, end_points


def get_loss(pred, label, end_points, uniform_weight=0, weights_decay=1e-4):
    """ pred: B*NUM_CLASSES, label: B  (pointasnl_cls.py:55-70) """
    regularization_loss = tf_util.regularization_loss(weights_decay)
    loss = torch.nn.functional.cross_entropy(pred, label.long(), reduction='none')
    classify_loss = loss.mean()
    if uniform_weight > 0:
        uniform_loss = get_repulsion_loss(end_points['l1_xyz'], nsample=20, radius=0.07)
    else:
        uniform_loss = classify_loss
    return classify_loss + uniform_weight * uniform_loss + regularization_loss
