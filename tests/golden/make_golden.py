"""Generates the committed known-answer fixtures.  The reference has no tests or golden vectors of its own
(SURVEY 4), so these are produced by RUNNING THE REFERENCE'S OWN CODE:

  python tests/golden/make_golden.py cpu      (this container: needs /root/reference -> oracle/_ref)
      ref_knn.npz, ref_interp.npz   <- nanoflann kNN (knn_.cxx) and threenn/threeinterpolate (tf_interpolate.cpp)
  python tests/golden/make_golden.py gpu      (GPU box, through gpurun; writes gpurun_out/ref_tfops_hip.npz,
                                               copied into tests/golden/ afterwards)
      ref_tfops_hip.npz             <- the reference .cu kernels compiled unchanged by hipcc -ffp-contract=off

Inputs are regenerated from seeds by tests/conftest.clouds, so the files hold only seeds + outputs.
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


if __name__ == "__main__":
    (make_cpu if sys.argv[1:] == ["cpu"] else make_gpu)()
