"""Generates the committed known-answer fixtures.  The reference has no tests or golden vectors of its own
(SURVEY 4), so these are produced by RUNNING THE REFERENCE'S OWN CODE:

  python tests/golden/make_golden.py cpu      (this container: needs /root/reference -> oracle/_ref)
      ref_knn.npz, ref_interp.npz   <- nanoflann kNN (knn_.cxx) and threenn/threeinterpolate (tf_interpolate.cpp)
  python tests/golden/make_golden.py gpu      (GPU box, through gpurun; writes gpurun_out/ref_tfops_hip.npz,
                                               copied into tests/golden/ afterwards)
      ref_tfops_hip.npz             <- the reference .cu kernels compiled unchanged by hipcc -ffp-contract=off

  python tests/golden/make_golden.py cells    (this container: imports and RUNS the reference's own Python --
  python tests/golden/make_golden.py models    utils/pointasnl_util.py, utils/pointnet_util.py, utils/tf_util.py,
  python tests/golden/make_golden.py losses    tf_ops/*/tf_*.py, models/pointasnl_*.py -- under oracle/tf_shim, a numpy
                                               stand-in for the TF symbols they use; custom ops = oracle/_ref + C oracle)
      ref_cells.npz                 <- SampleWeights/AdaptiveSampling, PointNonLocalCell, PointASNLSetAbstraction,
                                       PointASNLDecodingLayer, pointnet_fp_module, pointnet_sa_module, get_repulsion_loss
      ref_models.npz                <- get_model of the three models, small sizes and BASELINE.json configs[1..4]
      ref_losses.npz                <- get_loss of the three models

Inputs are regenerated from seeds (tests/conftest.clouds, tests/golden/ref_cases.py, bench.synth_*) and weights from
(seed, variable name, shape) by oracle/weights.py, so the files hold only outputs + the variable name/shape lists.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import clouds  # noqa: E402

KNN_CASES = [  # (seed, b, n, m, k, kind)
    (301, 4, 1024, 512, 32, "ball"), (302, 4, 512, 128, 64, "ball"), (303, 2, 2048, 256, 16, "cube"),
    (304, 2, 40, 40, 32, "ball"), (305, 2, 300, 64, 8, "cube"),
]
PICK_CASES = [(311, 3, 700, 90, 16, "ball"), (312, 2, 1500, 400, 8, "cube"), (313, 1, 64, 200, 5, "ball")]  # nq > n: counts wrap
NN_CASES = [(401, 4, 512, 128, "cube"), (402, 2, 2048, 256, "ball"), (403, 2, 320, 80, "lattice"), (404, 2, 10, 2, "cube")]
FPS_CASES = [(501, 4, 1024, 512, "ball"), (502, 3, 1024, 512, "lattice"), (503, 2, 2500, 300, "lattice"), (504, 6, 512, 128, "cube")]
GRIDSUB_CASES = [(701, 6000, 0.1, 3, 1), (702, 2500, 0.03, 0, 0), (703, 1500, 0.4, 5, 2)]  # seed, n, sampleDl, fdim, ldim
BALL_CASES = [(601, 8, 512, 128, 64, 0.1, "cube"), (602, 2, 1024, 512, 32, 0.2, "ball"), (603, 2, 700, 90, 16, 0.25, "lattice")]


def gridsub_inputs(seed, n, dl, fdim, ldim):
    """points in an anisotropic box, random features, labels that are a function of the voxel (no vote ties)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    p = (rng.random((n, 3)) * np.array([3.0, 2.0, 1.0]) - 0.7).astype(np.float32)
    f = rng.random((n, fdim)).astype(np.float32) if fdim else None
    vox = np.floor(p / np.float32(dl)).astype(np.int64)
    c = np.stack([(vox[:, 0] * 7 + vox[:, 1] * 3 + vox[:, 2] + l) % 5 for l in range(ldim)], 1).astype(np.int32) if ldim else None
    return p, f, c


def make_cpu():
    from oracle import ref

    out = {}
    for seed, b, n, m, k, kind in KNN_CASES:
        sup = clouds(seed, b, n, kind)
        out[f"knn_{seed}"] = ref.knn_batch(sup, sup[:, :m].copy(), k, omp=False).astype(np.int32)
    for seed, b, n, nq, k, kind in PICK_CASES:  # cpp_knn_batch_distance_pick with its time(0) seed pinned to `seed`
        i, q = ref.knn_batch_distance_pick(clouds(seed, b, n, kind), nq, k, seed)
        out[f"pick_idx_{seed}"], out[f"pick_q_{seed}"] = i.astype(np.int32), q
    from golden import ref_cases as RC
    for seed, b, n, m, k, kind in RC.KNN_TIE_CASES:  # exact ties: nanoflann's visit order
        sup, qry = RC.knn_tie_cloud(seed, b, n, m, kind)
        out[f"knn_tie_{seed}"] = ref.knn_batch(sup, qry, k, omp=False).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "ref_knn.npz"), **out)
    out = {}
    for seed, b, n, m, kind in NN_CASES:
        x1, x2 = clouds(seed, b, n, kind), clouds(seed + 50, b, m, kind)
        d, i = ref.three_nn(x1, x2)
        pts = np.random.Generator(np.random.PCG64(seed)).random((b, m, 16), dtype=np.float32)
        w = np.maximum(d, 1e-10)
        w = (1.0 / w) / (1.0 / w).sum(-1, keepdims=True)
        out[f"nn_dist_{seed}"], out[f"nn_idx_{seed}"] = d, i
        out[f"interp_{seed}"] = ref.three_interpolate(pts, i, w.astype(np.float32))
        g = np.random.Generator(np.random.PCG64(seed + 1)).random((b, n, 16), dtype=np.float32)
        out[f"interp_grad_{seed}"] = ref.three_interpolate_grad(pts, i, w.astype(np.float32), g)
    np.savez_compressed(os.path.join(HERE, "ref_interp.npz"), **out)
    out = {}
    for seed, n, dl, fdim, ldim in GRIDSUB_CASES:
        p, f, c = gridsub_inputs(seed, n, dl, fdim, ldim)
        res = ref.grid_subsample(p, f, c, dl)
        res = res if isinstance(res, tuple) else (res,)
        order = np.lexsort(res[0].T[::-1])  # the reference emits hash-table order: store the rows sorted by (x, y, z)
        for name, arr in zip(["pts"] + (["feat"] if fdim else []) + (["cls"] if ldim else []), res):
            out[f"{name}_{seed}"] = arr[order]
    np.savez_compressed(os.path.join(HERE, "ref_gridsub.npz"), **out)
    print("wrote ref_knn.npz, ref_interp.npz, ref_gridsub.npz")


def make_gpu():
    import torch

    from oracle import ref

    R = ref.HipRef(nofma=True)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    out = {}
    for seed, b, n, m, kind in FPS_CASES:
        out[f"fps_{seed}"] = R.farthest_point_sample(m, dev(clouds(seed, b, n, kind))).cpu().numpy()
    for seed, b, n, m, ns, r, kind in BALL_CASES:
        x1 = clouds(seed, b, n, kind)
        idx, cnt = R.query_ball_point(r, ns, dev(x1), dev(x1[:, :m].copy()))
        out[f"ball_idx_{seed}"], out[f"ball_cnt_{seed}"] = idx.cpu().numpy(), cnt.cpu().numpy()
    rng = np.random.Generator(np.random.PCG64(701))
    dist = rng.random((4, 32, 128), dtype=np.float32)
    dist[:, :, ::5] = np.round(dist[:, :, ::5] * 4) / 4
    oi, oo = R.select_top_k(16, dev(dist))
    out["topk_idx_701"], out["topk_val_701"] = oi.cpu().numpy()[:, :, :16], oo.cpu().numpy()[:, :, :16]
    p = np.random.Generator(np.random.PCG64(801)).random((3, 9000), dtype=np.float32)
    r = np.random.Generator(np.random.PCG64(802)).random((3, 256), dtype=np.float32)
    ps, cdf = R.prob_sample(dev(p), dev(r))
    out["prob_sample_801"] = ps.cpu().numpy()
    out["cdf_tail_801"] = cdf.cpu().numpy()[:, -8:]
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez_compressed("gpurun_out/ref_tfops_hip.npz", **out)
    print("wrote gpurun_out/ref_tfops_hip.npz")


# ---------------------------------------------------------------------------------------------------------------------
# cells / models / losses: the reference's Python, executed

def _ref_call_cell(tfs, case, x):
    """Call the reference function of `case` on shim tensors -> dict of outputs.  All keyword names are the reference's."""
    import pointasnl_util as U   # /root/reference/utils/pointasnl_util.py
    import pointnet_util as PU   # /root/reference/utils/pointnet_util.py

    T = lambda a: None if a is None else tfs.constant(a)  # noqa: E731
    fn = case["fn"]
    common = dict(is_training=tfs.constant(False), bn_decay=None)
    if fn == "AdaptiveSampling":
        nx, nf = U.AdaptiveSampling(T(x["group_xyz"]), T(x["group_feature"]), case["as_"], weight_decay=None, scope="layer1",
                                    bn=True, **common)
        return dict(new_xyz=nx, new_feature=nf)
    if fn == "PointNonLocalCell":
        c = case["c"]
        out = U.PointNonLocalCell(T(x["feature"]), T(x["new_point"]), [max(32, c // 2), case["out"]], weight_decay=None,
                                  scope="layerX", bn=True, **common)
        return dict(out=out)
    if fn == "PointASNLSetAbstraction":
        nx, npts = U.PointASNLSetAbstraction(T(x["xyz"]), T(x["feature"]), npoint=case["npoint"], nsample=case["nsample"],
                                             mlp=case["mlp"], weight_decay=None, scope="layerS", as_neighbor=case["as_"],
                                             NL=case["NL"], **common)
        return dict(new_xyz=nx, new_points=npts)
    if fn == "PointASNLDecodingLayer":
        out = U.PointASNLDecodingLayer(T(x["xyz1"]), T(x["xyz2"]), T(x["points1"]), T(x["points2"]), case["nsample"], case["mlp"],
                                       weight_decay=None, scope="fa", **common)
        return dict(out=out)
    if fn == "pointnet_fp_module":
        out = PU.pointnet_fp_module(T(x["xyz1"]), T(x["xyz2"]), T(x["points1"]), T(x["points2"]), case["mlp"], scope="fp", bn=True,
                                    **common)
        return dict(out=out)
    if fn == "pointnet_sa_module":
        _, out, _ = PU.pointnet_sa_module(T(x["xyz"]), T(x["points"]), npoint=None, radius=None, nsample=None, mlp=case["mlp"],
                                          mlp2=None, group_all=True, scope="sa_all", **common)
        return dict(out=out)
    if fn == "get_repulsion_loss":
        return dict(loss=U.get_repulsion_loss(T(x["pred"]), nsample=case["nsample"], radius=case["radius"]))
    raise KeyError(fn)


def _vars_json(tfs):
    import json

    return json.dumps([[k, list(v.shape)] for k, v in tfs.variables.items()])


def make_cells():
    from golden import ref_cases as R
    from oracle import tf_shim

    out = {}
    for case in R.CELL_CASES:
        x = R.cell_inputs(case)
        for dt, tag in ((np.float64, "f64"),):  # the fp32 evaluation is kept for the model graphs only (ref_models.npz)
            xs = {k: (None if v is None else v.astype(dt)) for k, v in x.items()}
            with tf_shim.session(seed=R.cell_seed(case), dtype=dt) as tfs:
                res = _ref_call_cell(tfs, case, xs)
                out[f"{case['name']}/vars"] = np.array(_vars_json(tfs))
            for k, v in res.items():
                out[f"{case['name']}/{k}_{tag}"] = np.asarray(v)
        if case.get("dup"):
            # clouds with exactly equidistant neighbours: nanoflann's order among them is traversal order, the product's is
            # ascending index.  Keep the reference's own neighbour lists so that the tests can (a) show that the
            # deviation is confined to runs of equal distance and (b) check everything downstream on the reference's lists.
            from oracle import ops, ref
            fps = ops.farthest_point_sample(case["npoint"], x["xyz"])
            q = np.take_along_axis(x["xyz"], fps[..., None].astype(np.int64), 1)
            out[f"{case['name']}/knn_idx"] = ref.knn_batch(x["xyz"], q, case["nsample"], omp=True).astype(np.int32)
        print(case["name"], {k: np.asarray(v).shape for k, v in res.items()})
    np.savez_compressed(os.path.join(HERE, "ref_cells.npz"), **out)
    print("wrote ref_cells.npz", os.path.getsize(os.path.join(HERE, "ref_cells.npz")))


def _ref_model(name):
    import importlib

    return importlib.import_module({"cls": "pointasnl_cls", "sem_seg": "pointasnl_sem_seg", "sem_seg_res": "pointasnl_sem_seg_res"}[name])


def make_models(only=None):
    """get_model of the reference, one cloud at a time for the big ones (every op is per cloud at inference)."""
    import time

    from golden import ref_cases as R
    from oracle import tf_shim

    path = os.path.join(HERE, "ref_models.npz")
    out = dict(np.load(path)) if os.path.exists(path) and only else {}
    for case in R.MODEL_CASES:
        if only and case["name"] not in only:
            continue
        pc = R.model_input(case)
        stride = case.get("stride", 1)
        t0 = time.time()
        for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
            logits, l1 = [], []
            chunk = 8 if case["model"] == "cls" else 1
            for s in range(0, case["b"], chunk):
                with tf_shim.session(seed=R.model_seed(case), dtype=dt) as tfs:
                    M = _ref_model(case["model"])
                    net, ep = M.get_model(tfs.constant(pc[s:s + chunk].astype(dt)), tfs.constant(False), **case["kw"])
                    out[f"{case['name']}/vars"] = np.array(_vars_json(tfs))
                logits.append(np.asarray(net))
                l1.append(np.asarray(ep["l1_xyz"]))
            logits, l1 = np.concatenate(logits), np.concatenate(l1)
            out[f"{case['name']}/logits_{tag}"] = logits if case["model"] == "cls" else logits[:, ::stride]
            if tag == "f64":
                out[f"{case['name']}/l1_xyz_{tag}"] = l1[:, ::max(1, l1.shape[1] // 64)].astype(np.float32)
        print(case["name"], logits.shape, f"{time.time() - t0:.1f}s", flush=True)
    np.savez_compressed(path, **out)
    print("wrote ref_models.npz", os.path.getsize(path))


def make_losses():
    from golden import ref_cases as R
    from oracle import tf_shim

    out = {}
    for case in R.LOSS_CASES:
        x = R.loss_inputs(case)
        with tf_shim.session(seed=R.loss_seed(case), dtype=np.float64) as tfs:
            M = _ref_model(case["model"])
            net, ep = M.get_model(tfs.constant(x["pc"].astype(np.float64)), tfs.constant(False), **case["kw"])
            args = [net, tfs.constant(x["label"]), ep]
            kw = dict(case["loss_kw"])
            if "smpw" in x:
                kw["smpw"] = tfs.constant(x["smpw"].astype(np.float64))
            loss = M.get_loss(*args, **kw)
            out[f"{case['name']}/vars"] = np.array(_vars_json(tfs))
        out[f"{case['name']}/loss_f64"] = np.asarray(loss)
        out[f"{case['name']}/logits_f64"] = np.asarray(net) if case["model"] == "cls" else np.asarray(net)[:, ::8]
        print(case["name"], float(loss))
    np.savez_compressed(os.path.join(HERE, "ref_losses.npz"), **out)


if __name__ == "__main__":
    mode = sys.argv[1]
    {"cpu": make_cpu, "gpu": make_gpu, "cells": make_cells, "losses": make_losses,
     "models": lambda: make_models(sys.argv[2:] or None)}[mode]()
