"""pointasnl_amd -- MI355X (gfx950) native set-abstraction hot path of PointASNL.

Layout mirrors the reference so its ``sys.path`` idiom keeps working (utils/pointasnl_util.py:11-19):

    tf_ops/sampling/tf_sampling.py            farthest_point_sample, gather_point, prob_sample
    tf_ops/grouping/tf_grouping.py            query_ball_point, group_point, select_top_k, knn_point
    tf_ops/3d_interpolation/tf_interpolate.py three_nn, three_interpolate
    utils/nearest_neighbors/lib/python/nearest_neighbors.py   knn_batch, knn
    utils/pointasnl_util.py, utils/pointnet_util.py, utils/tf_util.py, models/*.py   (torch host mirror)
    csrc/                                     hand-written HIP kernels + the C ABI (include/pasnl.h)

``install_paths()`` appends those directories to ``sys.path`` so that ``import tf_sampling`` etc. resolve
exactly as in the reference tree.  The modules are also reachable as attributes of this package.
"""
import importlib.util
import os
import sys

__version__ = "0.1.0"

_ROOT = os.path.dirname(os.path.abspath(__file__))

OP_DIRS = [
    os.path.join(_ROOT, "tf_ops", "sampling"),
    os.path.join(_ROOT, "tf_ops", "grouping"),
    os.path.join(_ROOT, "tf_ops", "3d_interpolation"),
    os.path.join(_ROOT, "utils"),
    os.path.join(_ROOT, "models"),
]


def install_paths():
    for d in OP_DIRS:
        if d not in sys.path:
            sys.path.append(d)


def _load(name, *parts):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(_ROOT, *parts))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    sys.modules[__name__ + "." + name] = mod  # also importable as pointasnl_amd.tf_sampling ...
    spec.loader.exec_module(mod)
    return mod


# the reference's module names; loaded under exactly those names so both import styles share one module
tf_sampling = _load("tf_sampling", "tf_ops", "sampling", "tf_sampling.py")
tf_grouping = _load("tf_grouping", "tf_ops", "grouping", "tf_grouping.py")
tf_interpolate = _load("tf_interpolate", "tf_ops", "3d_interpolation", "tf_interpolate.py")
from .utils.nearest_neighbors.lib.python import nearest_neighbors  # noqa: E402
