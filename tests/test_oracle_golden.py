"""CPU: the oracle (oracle/pasnl_oracle.c) against the committed known-answer vectors, which are outputs of the
reference's OWN code (tests/golden/make_golden.py): nanoflann kNN and threenn/threeinterpolate compiled from
/root/reference, and the reference CUDA kernels compiled unchanged by hipcc and run on an MI355X."""
import os

import numpy as np
import pytest

from conftest import clouds
from golden import make_golden as G
from oracle import ops as O

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def knn_gold():
    return np.load(os.path.join(HERE, "ref_knn.npz"))


@pytest.fixture(scope="module")
def interp_gold():
    return np.load(os.path.join(HERE, "ref_interp.npz"))


@pytest.fixture(scope="module")
def hip_gold():
    return np.load(os.path.join(HERE, "ref_tfops_hip.npz"))


@pytest.mark.parametrize("case", G.KNN_CASES)
def test_knn_matches_reference_nanoflann(knn_gold, case):
    seed, b, n, m, k, kind = case
    sup = clouds(seed, b, n, kind)
    got = O.knn_batch(sup, sup[:, :m].copy(), k)
    np.testing.assert_array_equal(got.astype(np.int32), knn_gold[f"knn_{seed}"])  # tie-free clouds: identical


@pytest.mark.parametrize("case", G.PICK_CASES)
def test_knn_distance_pick_matches_reference(knn_gold, case):
    """oracle_knn_distance_pick == the reference's cpp_knn_batch_distance_pick (knn_.cxx:136-200) run with its time(0) seed
    pinned (oracle/shims/ref_knn_shim.cpp overrides time() inside libref_knn.so); the random stream is numpy's MT19937 with
    legacy seeding == std::mt19937(seed)."""
    seed, b, n, nq, k, kind = case
    idx, q = O.knn_batch_distance_pick(clouds(seed, b, n, kind), nq, k, seed)
    np.testing.assert_array_equal(idx.astype(np.int32), knn_gold[f"pick_idx_{seed}"])
    np.testing.assert_array_equal(q, knn_gold[f"pick_q_{seed}"])


@pytest.mark.parametrize("case", G.NN_CASES)
def test_three_nn_interpolate_match_reference(interp_gold, case):
    seed, b, n, m, kind = case
    x1, x2 = clouds(seed, b, n, kind), clouds(seed + 50, b, m, kind)
    d, i = O.three_nn(x1, x2)
    np.testing.assert_array_equal(i, interp_gold[f"nn_idx_{seed}"])
    np.testing.assert_array_equal(d, interp_gold[f"nn_dist_{seed}"])
    pts = np.random.Generator(np.random.PCG64(seed)).random((b, m, 16), dtype=np.float32)
    w = np.maximum(d, 1e-10)
    w = ((1.0 / w) / (1.0 / w).sum(-1, keepdims=True)).astype(np.float32)
    np.testing.assert_array_equal(O.three_interpolate(pts, i, w), interp_gold[f"interp_{seed}"])
    g = np.random.Generator(np.random.PCG64(seed + 1)).random((b, n, 16), dtype=np.float32)
    np.testing.assert_array_equal(O.three_interpolate_grad(pts, i, w, g), interp_gold[f"interp_grad_{seed}"])


@pytest.mark.parametrize("case", G.GRIDSUB_CASES)
def test_grid_subsample_matches_reference(case):
    """oracle_grid_subsample vs outputs of the reference's own grid_subsampling.cpp (fixture rows sorted by x,y,z)"""
    seed, n, dl, fdim, ldim = case
    gold = np.load(os.path.join(HERE, "ref_gridsub.npz"))
    p, f, c = G.gridsub_inputs(seed, n, dl, fdim, ldim)
    res = O.grid_subsample(p, f, c, dl)
    res = res if isinstance(res, tuple) else (res,)
    order = np.lexsort(res[0].T[::-1])
    for name, arr in zip(["pts"] + (["feat"] if fdim else []) + (["cls"] if ldim else []), res):
        np.testing.assert_array_equal(arr[order], gold[f"{name}_{seed}"])


@pytest.mark.parametrize("case", G.FPS_CASES)
def test_fps_matches_reference_kernel(hip_gold, case):
    seed, b, n, m, kind = case
    np.testing.assert_array_equal(O.farthest_point_sample(m, clouds(seed, b, n, kind)), hip_gold[f"fps_{seed}"])


@pytest.mark.parametrize("case", G.BALL_CASES)
def test_ball_query_matches_reference_kernel(hip_gold, case):
    seed, b, n, m, ns, r, kind = case
    x1 = clouds(seed, b, n, kind)
    idx, cnt = O.query_ball_point(r, ns, x1, x1[:, :m].copy())
    np.testing.assert_array_equal(cnt, hip_gold[f"ball_cnt_{seed}"])
    np.testing.assert_array_equal(idx, hip_gold[f"ball_idx_{seed}"])


def test_selection_sort_and_prob_sample_match_reference_kernels(hip_gold):
    rng = np.random.Generator(np.random.PCG64(701))
    dist = rng.random((4, 32, 128), dtype=np.float32)
    dist[:, :, ::5] = np.round(dist[:, :, ::5] * 4) / 4
    oi, oo = O.select_top_k(16, dist)
    np.testing.assert_array_equal(oi[:, :, :16], hip_gold["topk_idx_701"])
    np.testing.assert_array_equal(oo[:, :, :16], hip_gold["topk_val_701"])
    p = np.random.Generator(np.random.PCG64(801)).random((3, 9000), dtype=np.float32)
    r = np.random.Generator(np.random.PCG64(802)).random((3, 256), dtype=np.float32)
    np.testing.assert_array_equal(O.cumsum(p)[:, -8:], hip_gold["cdf_tail_801"])
    np.testing.assert_array_equal(O.prob_sample(p, r), hip_gold["prob_sample_801"])


# ---- properties that do not need a second implementation
def test_fps_tie_rule_differs_from_plain_argmax_on_lattice():
    xyz = clouds(502, 1, 1024, "lattice")
    got = O.farthest_point_sample(64, xyz)[0]
    # plain lowest-index argmax FPS
    temp = np.full(1024, 1e38, np.float32)
    old, naive = 0, [0]
    for _ in range(63):
        d = ((xyz[0] - xyz[0, old]) ** 2).astype(np.float32)
        d = (d[:, 0] + d[:, 1]) + d[:, 2]
        temp = np.minimum(temp, d)
        old = int(np.argmax(temp))
        naive.append(old)
    assert (got != np.array(naive)).any(), "lattice input should exercise the (k mod 512, k) tie rule"
    assert len(set(got.tolist())) == 64


def test_knn_is_sorted_and_selfless_ties_by_index():
    sup = clouds(9, 2, 400, "lattice")
    idx, d = O.knn_batch(sup, sup[:, :50].copy(), 20, return_dist=True)
    assert (np.diff(d, axis=-1) >= 0).all()
    same = np.diff(d, axis=-1) == 0
    assert (np.diff(idx, axis=-1)[same] > 0).all()  # equal distances -> ascending index


def test_ball_query_properties():
    x = clouds(3, 2, 500, "cube")
    idx, cnt = O.query_ball_point(0.15, 8, x, x[:, :60].copy())
    assert ((cnt >= 1) & (cnt <= 8)).all()  # a query that is itself a dataset point always hits itself
    for b in range(2):
        for j in range(60):
            c = cnt[b, j]
            assert (np.diff(idx[b, j, :c]) > 0).all()
            assert (idx[b, j, c:] == idx[b, j, 0]).all()


def test_three_weights_sum_to_one():
    d = np.abs(clouds(1, 3, 100, "cube"))
    w = O.three_weights(d)
    np.testing.assert_allclose(w.sum(-1), 1.0, rtol=1e-6)


def test_the_prune_rounding_fixture_is_what_it_says():
    """tests/golden/knn_prune_rounding_case.npz: the reference library's row misses point 6152, which the exact (distance, index) list
    holds -- nanoflann skips its subtree on a pruning bound rounded one ulp above its distance (tests/test_gpu_tie_paths.py)."""
    d = np.load(os.path.join(HERE, "knn_prune_rounding_case.npz"))
    sup, qry, k, want = d["sup"], d["qry"], int(d["k"]), d["reference"]
    idx, dist = O.knn_batch(sup, qry, k, return_dist=True)
    assert 6152 in idx[0, 0] and 6152 not in want[0, 0]
    dref = np.array([((np.float32(sup[0, i, 0] - qry[0, 0, 0]) ** 2 + np.float32(sup[0, i, 1] - qry[0, 0, 1]) ** 2) + np.float32(sup[0, i, 2] - qry[0, 0, 2]) ** 2)
                     for i in want[0, 0]], np.float32)
    assert (np.diff(dref) >= 0).all() and dref[-1] > dist[0, 0][list(idx[0, 0]).index(6152)]
