"""GPU: the grid-pruned kNN (csrc/knn_grid.hip, clouds of >= 4096 points) returns exactly what the brute-force kernels
return -- same indices in the same (distance, index) order, same distance bits -- and both equal the C oracle, which is
pinned to the reference's nanoflann (tests/test_oracle_golden.py).  Inputs aim at the acceptance test of the ring search:
density gradients, flat clouds (2-D grid), duplicates and lattices (ties beyond the sort network), queries far outside the
support's bounding box, more neighbours than a ring holds."""
import numpy as np
import pytest
import torch

import bench as B
from conftest import clouds
from oracle import ops as O

pytestmark = pytest.mark.gpu


def both(sup, qry, k, dist=False):
    import pointasnl_amd as P
    from pointasnl_amd.utils.nearest_neighbors.lib.python import nearest_neighbors as NN

    s, q = torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda()
    NN.GRID = True
    g = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32).cpu().numpy()
    NN.GRID = False
    try:
        bf = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32).cpu().numpy()
    finally:
        NN.GRID = True
    return g, bf


def _cases():
    rng = np.random.default_rng(1)
    out = {}
    out["ball_8192_self"] = (B.synth_clouds(1, 3, 8192), None, 32)
    out["ball_8192_k16"] = (B.synth_clouds(2, 2, 8192), None, 16)
    out["scannet_8192"] = (B.synth_scannet(3, 2, 8192)[..., :3].copy(), None, 32)
    out["kitti_10240"] = (B.synth_kitti(4, 2, 10240), None, 32)
    flat = B.synth_clouds(5, 2, 6000)
    flat[..., 2] = 0.0
    out["flat_plane"] = (flat, None, 32)
    line = np.zeros((1, 4500, 3), np.float32)
    line[0, :, 0] = rng.random(4500)
    out["line"] = (line, None, 20)
    out["lattice"] = (clouds(6, 2, 5000, "lattice"), None, 32)        # ~730 distinct positions: massive ties + duplicates
    dup = B.synth_clouds(7, 2, 4096)
    dup[:, 1::2] = dup[:, 0::2]
    out["duplicates"] = (dup, None, 32)
    same = np.ones((1, 4200, 3), np.float32)
    out["all_identical"] = (same, None, 40)
    clus = B.synth_clouds(8, 2, 9000) * 0.01
    clus[:, :100] = B.synth_clouds(9, 2, 100) * 50.0                   # a dense blob inside a huge, nearly empty box
    out["density_gradient"] = (clus, None, 32)
    sup = B.synth_clouds(10, 2, 8192)
    out["queries_outside"] = (sup, (B.synth_clouds(11, 2, 777) * 3.0 + 2.0).astype(np.float32), 32)
    out["k64"] = (B.synth_clouds(12, 2, 4100), None, 64)
    out["k1"] = (B.synth_clouds(13, 2, 4096), B.synth_clouds(14, 2, 333), 1)
    out["n16384"] = (B.synth_clouds(15, 1, 16384), B.synth_clouds(15, 1, 16384)[:, :2000].copy(), 32)
    # clouds far from the origin (un-normalised / UTM-like coordinates): the cell faces x0 + c*h and the distances to them are
    # rounded relative to |x0| >> h, which the acceptance bound has to allow for
    far = (B.synth_clouds(16, 2, 6000) * 4.0 + np.float32(1.0e4)).astype(np.float32)
    out["translated_1e4"] = (far, None, 32)
    out["translated_3e5"] = ((B.synth_clouds(17, 1, 5000) * 50.0 + np.array([3.0e5, -2.0e5, 1.0e3])).astype(np.float32), None, 16)
    return out


CASES = _cases()


@pytest.mark.parametrize("name", list(CASES))
def test_grid_knn_is_bit_identical_to_brute_force_and_oracle(name):
    sup, qry, k = CASES[name]
    qry = sup if qry is None else qry
    g, bf = both(sup, qry, k)
    np.testing.assert_array_equal(g, bf)
    sub = slice(0, min(qry.shape[1], 600))
    want = O.knn_batch(sup[:1], qry[:1, sub], k).astype(np.int32)
    np.testing.assert_array_equal(g[:1, sub], want)


@pytest.mark.parametrize("name", ["kitti_10240", "k64", "n16384", "queries_outside", "k1", "duplicates"])
@pytest.mark.parametrize("cap", [1, 7, 512])
def test_grid_knn_background_form_gives_the_same_lists(name, cap):
    """pasnl_knn_batch_ws_bg (a capped grid that walks the queries: the form a serving loop uses for a search beside other
    work): the lists of the one-wave-per-query form, bit for bit, for caps below the batch size, odd caps and a usual one."""
    import pointasnl_amd as P
    sup, qry, k = CASES[name]
    qry = sup if qry is None else qry
    s, q = torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda()
    usual = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32)
    capped = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, max_workgroups=cap)
    assert torch.equal(usual, capped)


def test_grid_knn_distances_and_int64():
    from pointasnl_amd import _hip
    import ctypes

    sup = B.synth_clouds(21, 2, 8192)
    s = torch.from_numpy(sup).cuda()
    b, n, k = 2, 8192, 32
    nbytes = int(_hip.lib().pasnl_knn_workspace_bytes(b, n))
    assert nbytes > 0 and int(_hip.lib().pasnl_knn_workspace_bytes(b, 2048)) == 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    idx = torch.empty((b, n, k), dtype=torch.int64, device="cuda")
    d = torch.empty((b, n, k), dtype=torch.float32, device="cuda")
    _hip.launch("pasnl_knn_batch_ws", "knn", b, n, n, k, _hip.ptr(s), _hip.ptr(s), _hip.ptr(idx), 1, _hip.ptr(d), _hip.ptr(ws),
                ctypes.c_size_t(nbytes))
    i2 = torch.empty((b, n, k), dtype=torch.int64, device="cuda")
    d2 = torch.empty((b, n, k), dtype=torch.float32, device="cuda")
    _hip.launch("pasnl_knn_batch", "knn", b, n, n, k, _hip.ptr(s), _hip.ptr(s), _hip.ptr(i2), 1, _hip.ptr(d2))
    assert torch.equal(idx, i2) and torch.equal(d, d2)
    with pytest.raises(_hip.PasnlError):  # a workspace that is too small is refused before anything is launched
        _hip.launch("pasnl_knn_batch_ws", "knn", b, n, n, k, _hip.ptr(s), _hip.ptr(s), _hip.ptr(idx), 1, _hip.ptr(d), _hip.ptr(ws),
                    ctypes.c_size_t(nbytes - 1))


# ---- the reference's own order among exactly equal distances (csrc/knn_tree.hip): nanoflann's tree and search on the GPU
import os  # noqa: E402
import sys  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from golden import ref_cases as RC  # noqa: E402
import pointasnl_amd as P  # noqa: E402

GOLD_KNN = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_knn.npz"))


@pytest.mark.parametrize("case", RC.KNN_TIE_CASES, ids=lambda c: f"{c[0]}_{c[5]}")
def test_knn_nanoflann_tie_order_matches_the_reference(case):
    """knn_batch(..., tie_order="nanoflann") equals the reference's cpp_knn_batch (knn_.cxx + nanoflann compiled where it lies:
    tests/golden/ref_knn.npz, made by tests/golden/make_golden.py) INDEX FOR INDEX on lattices, duplicated points, identical
    points, collinear points and queries equidistant from several points -- where the canonical (distance, index) order of
    the default kernels differs inside the runs of equal distance."""
    seed, b, n, m, k, kind = case
    sup, qry = RC.knn_tie_cloud(seed, b, n, m, kind)
    want = GOLD_KNN[f"knn_tie_{seed}"]
    s, q = torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda()
    got = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, tie_order="nanoflann").cpu().numpy()
    np.testing.assert_array_equal(got, want)
    got64 = P.nearest_neighbors.knn_batch(s, q, k, tie_order="nanoflann")
    assert got64.dtype == torch.int64 and np.array_equal(got64.cpu().numpy(), want)
    # the default order: the same distances position by position, a different order only inside runs of equal distance
    canon = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32).cpu().numpy()
    d = lambda idx: ((qry[:, :, None, :] - np.take_along_axis(sup[:, None, :, :], idx[..., None].astype(np.int64), axis=2)) ** 2).sum(-1)
    np.testing.assert_array_equal(d(canon), d(want))
    if kind in ("lattice", "lattice16", "dup", "same", "line", "lattice_q_off"):
        assert (canon != want).any(), "these clouds are built to have ties the two orders resolve differently"


def test_knn_nanoflann_matches_live_reference_when_built():
    from oracle import ref
    if not ref.available("libref_knn.so"):
        pytest.skip("oracle/_ref/libref_knn.so not built here")
    sup = (np.round(np.random.default_rng(5).random((2, 1500, 3)) * 12) / 12).astype(np.float32)
    qry = np.random.default_rng(6).random((2, 200, 3)).astype(np.float32)
    want = ref.knn_batch(sup, qry, 24)
    got = P.nearest_neighbors.knn_batch(torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda(), 24, tie_order="nanoflann")
    np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("n,m,k,kind", [(8192, 300, 32, "lattice"), (10240, 200, 16, "dup"), (37, 20, 8, "lattice"), (11, 5, 3, "lattice"),
                                        (10, 4, 3, "lattice"), (5000, 256, 32, "plane"), (12000, 64, 8, "lattice")])
def test_knn_nanoflann_parallel_build_matches_live_reference(n, m, k, kind):
    """The workgroup-per-cloud build (n <= 10240; csrc/knn_tree.hip knn_tree_build_par_kernel) and the one-lane build behind it
    (n = 12000) against the reference library itself, on clouds made of ties: lattices, duplicated points, a flat cloud."""
    from oracle import ref
    if not ref.available("libref_knn.so"):
        pytest.skip("oracle/_ref/libref_knn.so not built here")
    rng = np.random.default_rng(n + m)
    sup = rng.random((3, n, 3))
    if kind == "lattice":
        sup = np.round(sup * 10) / 10
    elif kind == "dup":
        sup[:, n // 2:] = sup[:, : n - n // 2]  # every point twice
        sup = np.round(sup * 64) / 64
    elif kind == "plane":
        sup[..., 2] = 0.5
        sup = np.round(sup * 40) / 40
    sup = sup.astype(np.float32)
    qry = np.concatenate([sup[:, : m // 2], rng.random((3, m - m // 2, 3)).astype(np.float32)], axis=1)
    want = ref.knn_batch(sup, qry, k)
    got = P.nearest_neighbors.knn_batch(torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda(), k, tie_order="nanoflann")
    np.testing.assert_array_equal(got.cpu().numpy(), want)
