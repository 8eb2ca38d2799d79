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
    g = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, tie_order="index").cpu().numpy()
    NN.GRID = False
    try:
        bf = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, tie_order="index").cpu().numpy()
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
    for order in ("index", "reference"):
        usual = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, tie_order=order)
        capped = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, max_workgroups=cap, tie_order=order)
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
    # THE DEFAULT PATH (canonical search + the tree for the flagged queries only): the reference's lists, index for index
    stats = []
    dflt = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, stats=stats)
    np.testing.assert_array_equal(dflt.cpu().numpy(), want)
    assert int(stats[0].max()) <= m and (int(stats[0].sum()) > 0 or kind not in ("lattice", "lattice16", "dup", "same", "line", "lattice_q_off"))
    d64 = P.nearest_neighbors.knn_batch(s, q, k)
    assert d64.dtype == torch.int64 and np.array_equal(d64.cpu().numpy(), want)
    # the canonical order: the same distances position by position, a different order only inside runs of equal distance
    canon = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, tie_order="index").cpu().numpy()
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
    dflt = P.nearest_neighbors.knn_batch(torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda(), 24)
    np.testing.assert_array_equal(dflt.cpu().numpy(), want)


@pytest.mark.parametrize("n,m,k,kind", [(8192, 300, 32, "lattice"), (10240, 200, 16, "dup"), (37, 20, 8, "lattice"), (11, 5, 3, "lattice"),
                                        (10, 4, 3, "lattice"), (5000, 256, 32, "plane"), (12000, 64, 8, "lattice"),
                                        (3000, 100, 16, "lattice"), (6500, 100, 16, "clustered"), (20000, 128, 16, "lattice"),
                                        (30000, 100, 8, "clustered"), (70000, 64, 16, "dup")])
def test_knn_nanoflann_parallel_build_matches_live_reference(n, m, k, kind):
    """The parallel builds (csrc/knn_tree.hip: records in LDS up to 8192 points, in the workspace above; nodes above 2048 points split
    by the whole workgroup, nodes up to 64 points in registers) against the reference library itself, on clouds made of ties:
    lattices, duplicated points, a flat cloud, clouds with 90 % of their points in one corner (lopsided trees)."""
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
    elif kind == "clustered":  # lopsided trees: 90 % of the points in a corner of the box (the large-node splits several levels deep)
        sup[:, : n * 9 // 10] *= 0.03
        sup = np.round(sup * 512) / 512
    sup = sup.astype(np.float32)
    qry = np.concatenate([sup[:, : m // 2], rng.random((3, m - m // 2, 3)).astype(np.float32)], axis=1)
    want = ref.knn_batch(sup, qry, k)
    got = P.nearest_neighbors.knn_batch(torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda(), k, tie_order="nanoflann")
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    stats = []
    dflt = P.nearest_neighbors.knn_batch(torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda(), k, stats=stats)
    np.testing.assert_array_equal(dflt.cpu().numpy(), want)   # the default path: only the flagged queries went through the tree
    assert int(stats[0].sum()) <= 3 * m and (int(stats[0].sum()) > 0 or n < 100)


# ---- the default path (pasnl_knn_batch_ref): which queries it sends through the tree
@pytest.mark.parametrize("name", ["ball_8192_self", "scannet_8192", "kitti_10240", "k64", "queries_outside", "translated_1e4"])
def test_default_order_on_tie_free_clouds_flags_nothing_and_is_the_canonical_list(name):
    """Clouds whose distances are distinct: no query is flagged (the kernels behind the search return at once), the result is the
    canonical list bit for bit -- and the reference library's where it is here."""
    from oracle import ref
    sup, qry, k = CASES[name]
    qry = sup if qry is None else qry
    s, q = torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda()
    stats = []
    dflt = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, stats=stats)
    canon = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, tie_order="index")
    nflag = int(stats[0].sum())
    # (a pair of exactly equal fp32 distances occurs by chance: about one query in 30 000 at K = 32 on unit-ball clouds, more where
    # the coordinates are large against the distances -- translated clouds -- and few of the mantissa's bits are left to differ)
    assert nflag <= max(3, qry.shape[0] * qry.shape[1] // 300), nflag
    if nflag == 0:
        assert torch.equal(dflt, canon)
    if ref.available("libref_knn.so"):
        sub = slice(0, min(qry.shape[1], 1500))
        np.testing.assert_array_equal(dflt[:, sub].cpu().numpy(), ref.knn_batch(sup, qry[:, sub].copy(), k))


@pytest.mark.parametrize("name", ["lattice", "duplicates", "all_identical", "flat_plane", "line"])
def test_default_order_equals_the_full_tree_search_on_clouds_made_of_ties(name):
    """default (flagged queries only) == tie_order="nanoflann" (every query through the tree) == the reference library, on the
    grid-pruned path's tie cases (n >= 4096: knn_grid's flags) -- and on a brute-force-sized slice of them (knn2's flags)."""
    from oracle import ref
    sup, qry, k = CASES[name]
    for n in (sup.shape[1], 1500):
        sp = np.ascontiguousarray(sup[:, :n])
        qr = np.ascontiguousarray(sp[:, :700])
        s, q = torch.from_numpy(sp).cuda(), torch.from_numpy(qr).cuda()
        stats = []
        dflt = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, stats=stats)
        full = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, tie_order="nanoflann")
        assert torch.equal(dflt, full), (name, n)
        if ref.available("libref_knn.so"):
            np.testing.assert_array_equal(dflt.cpu().numpy(), ref.knn_batch(sp, qr, k))


def test_default_order_mixed_batch_only_builds_the_trees_it_needs():
    """A batch where ONE cloud has ties: its flagged queries go through the tree, the other clouds' counts stay zero."""
    from oracle import ref
    sup = B.synth_clouds(31, 4, 1024)
    sup[2] = (np.round(sup[2] * 8) / 8).astype(np.float32)
    qry = np.ascontiguousarray(sup[:, :512])
    stats = []
    dflt = P.nearest_neighbors.knn_batch(torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda(), 32, dtype=torch.int32, stats=stats)
    counts = stats[0].cpu().numpy()
    assert counts[2] > 400 and counts[[0, 1, 3]].sum() <= 2, counts
    if ref.available("libref_knn.so"):
        np.testing.assert_array_equal(dflt.cpu().numpy(), ref.knn_batch(sup, qry, 32))


@pytest.mark.parametrize("k,n", [(1, 300), (5, 2500), (16, 3000), (16, 9000), (33, 600), (64, 64), (64, 5000), (7, 7)])
def test_default_order_random_lattices_vs_live_reference(k, n):
    """sweeps over K (both selection widths, the insertion kernel for K <= 16 on clouds above 2048 points, K = n) on half-snapped
    clouds: some queries tie inside the list, some only at its end, some not at all."""
    from oracle import ref
    if not ref.available("libref_knn.so"):
        pytest.skip("oracle/_ref/libref_knn.so not built here")
    rng = np.random.default_rng(k * 1000 + n)
    sup = rng.random((3, n, 3))
    snap = rng.random((3, n)) < 0.5
    sup[snap] = np.round(sup[snap] * 6) / 6
    sup = sup.astype(np.float32)
    m = min(n, 400)
    qry = np.concatenate([sup[:, : m // 2], rng.random((3, m - m // 2, 3)).astype(np.float32)], axis=1)
    stats = []
    dflt = P.nearest_neighbors.knn_batch(torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda(), k, stats=stats)
    np.testing.assert_array_equal(dflt.cpu().numpy(), ref.knn_batch(sup, qry, k))
    if k > 1 and n > 64:
        assert 0 < int(stats[0].sum()) <= 3 * m


@pytest.mark.parametrize("k,n,m", [(100, 700, 200), (200, 3000, 150), (256, 256, 64), (65, 5000, 100), (129, 1000, 80)])
def test_wide_lists_default_and_every_query_tree_vs_live_reference(k, n, m):
    """K > 64 (VERDICT r05 missing 4: the tree kernels refused it): the wave-per-query search keeps the sorted list in 2 / 4
    registers per lane; half-snapped clouds so that the lists contain ties."""
    from oracle import ref
    if not ref.available("libref_knn.so"):
        pytest.skip("oracle/_ref/libref_knn.so not built here")
    rng = np.random.default_rng(k + n)
    sup = rng.random((2, n, 3))
    snap = rng.random((2, n)) < 0.5
    sup[snap] = np.round(sup[snap] * 5) / 5
    sup = sup.astype(np.float32)
    qry = np.concatenate([sup[:, : m // 2], rng.random((2, m - m // 2, 3)).astype(np.float32)], axis=1)
    want = ref.knn_batch(sup, qry, k)
    s, q = torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda()
    np.testing.assert_array_equal(P.nearest_neighbors.knn_batch(s, q, k).cpu().numpy(), want)
    np.testing.assert_array_equal(P.nearest_neighbors.knn_batch(s, q, k, tie_order="nanoflann").cpu().numpy(), want)


def test_the_references_own_shape_81920_points():
    """utils/nearest_neighbors/test.py:5-8 of the reference: knn_batch on (16, 81920, 3) random points, K = 16 (queries = the
    points).  Here: the default path at that cloud size (n > 65535: beyond the 16-bit packing of the lane-per-query tree
    search; the canonical search is brute force above 16384 points) on 2 clouds x 4096 of their points as queries, against the
    reference library; ties exist by chance only, so the tree is built (one-lane build, ~1 s) for the clouds that have one."""
    import time
    from oracle import ref
    rng = np.random.default_rng(0)
    sup = rng.random((2, 81920, 3)).astype(np.float32)       # np.random.rand(batch_size, num_points, 3).astype(np.float32)
    qry = np.ascontiguousarray(sup[:, :4096])
    s, q = torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda()
    stats = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = P.nearest_neighbors.knn_batch(s, q, 16, stats=stats)
    torch.cuda.synchronize()
    print(f"default path, 2 x 81920 points, 4096 queries each, K=16: {1e3 * (time.perf_counter() - t0):.1f} ms, flagged {stats[0].tolist()}")
    canon = P.nearest_neighbors.knn_batch(s, q, 16, tie_order="index")
    d = lambda idx: ((qry[:, :, None, :] - np.take_along_axis(sup[:, None, :, :], idx[..., None], axis=2)) ** 2).sum(-1)
    np.testing.assert_array_equal(d(got.cpu().numpy()), d(canon.cpu().numpy()))    # the same distances whatever the order
    if ref.available("libref_knn.so"):
        np.testing.assert_array_equal(got.cpu().numpy(), ref.knn_batch(sup, qry, 16))
    # and a snapped cloud of that size: every query ties, every cloud gets its tree
    sup2 = (np.round(sup[:1] * 64) / 64).astype(np.float32)
    q2 = np.ascontiguousarray(sup2[:, :512])
    got2 = P.nearest_neighbors.knn_batch(torch.from_numpy(sup2).cuda(), torch.from_numpy(q2).cuda(), 16)
    if ref.available("libref_knn.so"):
        np.testing.assert_array_equal(got2.cpu().numpy(), ref.knn_batch(sup2, q2, 16))


@pytest.mark.parametrize("many", [False, True])
@pytest.mark.parametrize("n,k,kind", [(10240, 32, "kitti"), (9000, 16, "ball"), (10240, 64, "ball"), (10240, 8, "plane"), (8193, 32, "ball"),
                                      (8192, 32, "ball")])
def test_few_listed_queries_take_the_on_demand_tree(n, k, kind, many):
    """Clouds of thousands of points with a HANDFUL of tied queries (what the segmentation models' input levels meet: chance ties):
    knn_tree_lazy_kernel builds the reference's KD-tree only along each search's path, a workgroup per listed query.  Ties are
    planted -- a few duplicated points -- so that a handful of queries in two clouds of four are listed; `many`: a dozen more in a
    third cloud, past the form's limit (the batch then takes the full builds).  Equal to the reference library, to
    every-query-through-the-tree, and deterministic; `stats[1]`, `stats[2]` say which form ran."""
    from oracle import ref
    b = 4
    if kind == "kitti":
        sup = B.synth_kitti(n, b, n)
    else:
        sup = B.synth_clouds(n, b, n)
        if kind == "plane":
            sup[..., 2] = 0.0
    sup = sup.copy()
    sup[0, 2000] = sup[0, 7]                  # one duplicated support point next to a query: its neighbours see a tie
    sup[1, 2000] = sup[1, 47]
    sup[1, 2001] = sup[1, 130]
    if many:
        for t in range(40):
            sup[2, 2100 + t] = sup[2, 5 + 7 * t]
    qry = np.ascontiguousarray(sup[:, :300])
    s, q = torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda()
    stats = []
    got = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, stats=stats)
    # stats: listed queries per cloud; what the tie paths left (duplicates share a leaf: its reading order needs positions -- clouds of
    # more than 8192 points hand those on); what the on-demand tree left to the full builds
    counts, left, left2 = stats[0].cpu().numpy(), stats[1].cpu().numpy(), stats[2].cpu().numpy()
    assert counts[0] >= 1 and counts[1] >= 2 and counts[3] <= 1, counts
    assert ((left == 0) | (left == counts)).all(), (counts, left)
    if n <= 8192:
        pass                                             # (records fit the LDS: the full build is the faster form there)
    elif left.sum() > 32:
        assert many and (left2 == left).all(), (counts, left, left2)            # past the limit: the full builds took what was left
    else:
        assert not many and left.sum() >= 3 and (left2 == 0).all(), (counts, left, left2)   # every query left got its own on-demand tree
    again = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32)
    assert torch.equal(got, again)
    full = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, tie_order="nanoflann")
    assert torch.equal(got, full)
    if ref.available("libref_knn.so"):
        np.testing.assert_array_equal(got.cpu().numpy(), ref.knn_batch(sup, qry, k))


@pytest.mark.parametrize("seed", range(10))
def test_on_demand_tree_random_sweep(seed):
    """The on-demand tree (clouds of 8193..10240 points, at most 32 listed queries in the batch) on varied geometry: clustered
    (lopsided trees), flat, anisotropic and lattice-with-jitter clouds, queries outside the cloud's box, K from 1 to 64, batches
    of 1-6 clouds; ties planted by duplicating a few points / snapping a few coordinates.  Against the reference library and
    against every-query-through-the-tree."""
    from oracle import ref
    rng = np.random.default_rng(1000 + seed)
    b = int(rng.integers(1, 7))
    n = int(rng.choice([8193, 8500, 9000, 9999, 10240]))
    k = int(rng.choice([1, 2, 8, 16, 32, 33, 64]))
    m = int(rng.integers(50, 400))
    sup = rng.random((b, n, 3))
    kind = seed % 5
    if kind == 0:
        sup[:, : n * 9 // 10] *= 0.05                      # clustered: very lopsided trees
    elif kind == 1:
        sup[..., 2] *= 1e-3                                # nearly flat
    elif kind == 2:
        sup *= np.array([50.0, 1.0, 0.02])                 # anisotropic
    elif kind == 3:
        sup = np.round(sup * 40) / 40 + rng.random((b, n, 3)) * 1e-4   # lattice with jitter: near-ties everywhere, few exact ones
    sup = sup.astype(np.float32)
    qry = np.concatenate([sup[:, : m // 2], (rng.random((b, m - m // 2, 3)) * 1.5 - 0.25).astype(np.float32) * sup.max((0, 1))], axis=1)
    for t in range(int(rng.integers(1, 4))):               # a few exact ties next to queries
        c = int(rng.integers(0, b))
        sup[c, n - 1 - t] = sup[c, int(rng.integers(0, m // 2))]
    qry = np.ascontiguousarray(qry)
    s, q = torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda()
    stats = []
    got = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, stats=stats)
    full = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, tie_order="nanoflann")
    assert torch.equal(got, full), (b, n, k, kind, stats[0].tolist(), stats[1].tolist())
    if ref.available("libref_knn.so"):
        np.testing.assert_array_equal(got.cpu().numpy(), ref.knn_batch(sup, qry, k))
