"""Parity of the HIP path (through the C ABI, via the reference-named Python modules) against the oracle.
Bit-exact for indices; exact for gathers; 1e-5 (in fact exact) for interpolation."""
import numpy as np
import pytest
import torch

from conftest import clouds
from oracle import ops as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import pointasnl_amd

    return pointasnl_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("b,n,m,kind", [
    (3, 1024, 512, "ball"), (2, 512, 128, "ball"), (2, 1024, 512, "lattice"), (2, 600, 77, "cube"),
    (1, 64, 64, "ball"), (1, 37, 5, "lattice"), (2, 2048, 300, "lattice"), (1, 8192, 1024, "cube"),
    (1, 10240, 1280, "ball"), (2, 1, 1, "ball"), (1, 300, 1, "ball"),
])
def test_fps(P, b, n, m, kind):
    xyz = clouds(11 + n, b, n, kind)
    want = O.farthest_point_sample(m, xyz)
    got = P.tf_sampling.farthest_point_sample(m, dev(xyz)).cpu().numpy()
    assert got.dtype == np.int32 and got.shape == (b, m)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("case", ["ragged_2050", "ragged_5003", "ragged_9999", "lattice_4100", "lattice_8192", "lattice16_10240",
                                  "duplicates_6000", "exhausted_3000", "kitti_10240", "scannet_8192", "plane_5000", "line_3000"])
def test_fps_large_clouds_exact(P, case):
    """The kernels of the large clouds (n > 2048: Morton-sorted blobs, box-pruned updates, several picks per round) against the
    oracle, bit for bit, on what their shortcuts could get wrong: cloud sizes that are not multiples of 4 (LDS layout),
    lattices (distance ties: the (distance, k mod 512, k) rule across lanes, waves and picks of one round), duplicated
    points, more samples than distinct points (the field of zeros), metre-scale lidar-like coordinates, degenerate extents."""
    import bench as B
    rng = np.random.default_rng(sum(map(ord, case)))
    if case.startswith("ragged"):
        n = int(case.split("_")[1]); xyz, m = clouds(41, 2, n, "ball"), n // 8
    elif case == "lattice_4100":
        xyz, m = clouds(42, 2, 4100, "lattice"), 600        # 729 distinct positions: ties everywhere, then exhaustion
    elif case == "lattice_8192":
        xyz, m = clouds(43, 1, 8192, "lattice"), 1024
    elif case == "lattice16_10240":
        xyz, m = (np.round(rng.random((1, 10240, 3)) * 16) / 16).astype(np.float32), 1280   # 4913 positions
    elif case == "duplicates_6000":
        base = clouds(44, 2, 1500, "cube"); xyz, m = np.tile(base, (1, 4, 1)), 800
        xyz = xyz[:, rng.permutation(6000)].copy()
    elif case == "exhausted_3000":
        base = clouds(45, 1, 100, "cube"); xyz, m = np.tile(base, (1, 30, 1)).copy(), 300   # 100 distinct points, 300 samples
    elif case == "kitti_10240":
        xyz, m = B.synth_kitti(46, 2, 10240), 1280
    elif case == "scannet_8192":
        xyz, m = B.synth_scannet(47, 2, 8192)[..., :3].copy(), 1024
    elif case == "plane_5000":
        xyz, m = clouds(48, 1, 5000, "cube"), 700; xyz[..., 2] = 0.5
    else:
        xyz, m = clouds(49, 1, 3000, "cube"), 400; xyz[..., 1:] = 0.25
    want = O.farthest_point_sample(m, xyz)
    got = P.tf_sampling.farthest_point_sample(m, dev(xyz)).cpu().numpy()
    np.testing.assert_array_equal(got, want)


def test_fps_exhausted_cloud(P):
    # more samples than distinct points: once every distance is 0 the reference keeps returning index 0
    xyz = np.repeat(clouds(5, 1, 4, "cube"), 8, axis=1)
    want = O.farthest_point_sample(20, xyz)
    got = P.tf_sampling.farthest_point_sample(20, dev(xyz)).cpu().numpy()
    np.testing.assert_array_equal(got, want)


def test_gather_point_and_grad(P):
    xyz = clouds(3, 4, 777)
    idx = np.random.default_rng(0).integers(0, 777, (4, 333)).astype(np.int32)
    got = P.tf_sampling.gather_point(dev(xyz), dev(idx)).cpu().numpy()
    np.testing.assert_array_equal(got, O.gather_point(xyz, idx))
    x = dev(xyz).requires_grad_(True)
    out = P.tf_sampling.gather_point(x, dev(idx))
    g = np.random.default_rng(1).random((4, 333, 3), dtype=np.float32)
    out.backward(dev(g))
    # deterministic backward: contributions summed in ascending output order == the sequential loop, bit for bit
    np.testing.assert_array_equal(x.grad.cpu().numpy(), O.gather_point_grad(xyz, idx, g))


@pytest.mark.parametrize("kind", ["ball", "lattice"])
def test_fps_many_clouds_take_the_one_wave_kernel(P, kind):
    """More than 640 clouds of 513..1024 points are sampled by one wave per cloud (16 points per lane): the same picks, ties
    included (the lattice clouds are full of equal distances)."""
    xyz = clouds(77, 704, 1000, kind)
    want = O.farthest_point_sample(300, xyz)
    got = P.tf_sampling.farthest_point_sample(300, dev(xyz)).cpu().numpy()
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("b,n,m,ns,r,kind", [
    (32, 512, 128, 64, 0.1, "cube"),      # the reference's own micro-bench shape (tf_grouping.py:78-88)
    (4, 1024, 512, 32, 0.2, "ball"),
    (2, 1024, 1024, 20, 0.07, "ball"),    # repulsion-loss shape (pointasnl_util.py:361)
    (2, 700, 90, 16, 0.25, "lattice"),    # distances exactly on / next to the radius
    (2, 5000, 64, 8, 0.5, "cube"),        # early exit, multi-tile cloud
    (1, 3000, 50, 200, 0.4, "cube"),      # nsample > 64
    (2, 100, 10, 4, 1e-3, "cube"),        # zero-hit rows
])
def test_query_ball_point(P, b, n, m, ns, r, kind):
    xyz1 = clouds(21, b, n, kind)
    xyz2 = clouds(22, b, m, kind) if kind != "lattice" else xyz1[:, :m].copy()
    want_idx, want_cnt = O.query_ball_point(r, ns, xyz1, xyz2)
    idx, cnt = P.tf_grouping.query_ball_point(r, ns, dev(xyz1), dev(xyz2))
    np.testing.assert_array_equal(cnt.cpu().numpy(), want_cnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), want_idx)


@pytest.mark.parametrize("case", ["outside", "huge_radius", "degenerate", "plane", "many_queries", "nmax", "tiny", "dense",
                                  "far_outlier", "odd_nsample", "inf_point", "inf_radius", "unaligned", "clustered",
                                  "crowded_lanes"])
def test_query_ball_point_grid_edges(P, case):
    """Geometry the grid-pruned kernel must survive with bit-identical results: queries outside the cloud's bounding
    box, a radius larger than the cloud (single cell), coincident points, flat clouds, more queries than one
    workgroup handles, the largest LDS-resident cloud, clouds with an outlier that stretches the grid."""
    rng = np.random.default_rng(sum(map(ord, case)))
    b, n, m, ns, r = 2, 700, 300, 16, 0.15
    xyz1 = clouds(31, b, n, "cube")
    xyz2 = clouds(32, b, m, "cube")
    if case == "outside":
        xyz2 = (xyz2 * 3 - 1).astype(np.float32)  # two thirds of the queries lie outside [0,1)^3
    elif case == "huge_radius":
        r, ns = 5.0, 64
    elif case == "degenerate":
        xyz1 = np.repeat(xyz1[:, :1], n, axis=1).copy()
        xyz2[:, ::2] = xyz1[:, :m:2] if m <= n else xyz2[:, ::2]
    elif case == "plane":
        xyz1[..., 2] = 0.25
        xyz2[..., 2] = np.float32(0.25) + (rng.random((b, m)).astype(np.float32) - 0.5) * 0.2
    elif case == "many_queries":
        m = 1500
        xyz2 = clouds(33, b, m, "cube")
    elif case == "nmax":
        n, ns, r = 2048, 40, 0.08
        xyz1 = clouds(34, b, n, "cube")
    elif case == "tiny":
        n, m, ns = 3, 5, 4
        xyz1, xyz2 = clouds(35, b, n, "cube"), clouds(36, b, m, "cube")
        r = 0.6
    elif case == "dense":
        r, ns = 0.4, 8  # hundreds of hits per query, early exit in the brute-force kernel
    elif case == "far_outlier":
        xyz1[:, 5] = 1e6  # one point stretches the bounding box: everything else collapses into one cell
        xyz2[:, 3] = 1e6
    elif case == "odd_nsample":
        ns = 37
    elif case == "inf_point":
        xyz1[:, 7, 1] = np.inf  # the bounding box is not finite: one cell, every query walks every point
    elif case == "inf_radius":
        r, ns = float("inf"), 24  # everything is a hit: the first nsample indices
    elif case == "unaligned":
        n = 701  # clouds of 701 * 12 bytes: no 16-byte pieces
        xyz1 = clouds(37, b, n, "cube")
    elif case == "clustered":
        xyz1[:, 100:400] = xyz1[:, 100:101] + (rng.random((b, 300, 3)).astype(np.float32) - 0.5) * 0.02  # runs of > 31 records
        xyz2[:, :50] = xyz1[:, 100:150]
    elif case == "crowded_lanes":
        r, ns = 0.17, 32  # ~14 expected hits: some lanes exceed the in-register list and take the bit rows, most do not
    want_idx, want_cnt = O.query_ball_point(r, ns, xyz1, xyz2)
    idx, cnt = P.tf_grouping.query_ball_point(r, ns, dev(xyz1), dev(xyz2))
    np.testing.assert_array_equal(cnt.cpu().numpy(), want_cnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), want_idx)


@pytest.mark.parametrize("n", [2048, 2047, 1500])
@pytest.mark.parametrize("r", [0.24, 0.3, 0.12])
def test_query_ball_point_wedge_empty_tail(P, n, r):
    """ADVICE r05 (ball_grid.hip:397): at n == 2048 an EMPTY run whose cell start equals n used to become the table entry 0x800 =
    "one record at position 0"; as a lane's trailing entry it counted record 0 twice when record 0 lay inside the ball.  A wedge that
    is long in x and thin in y / z, tapering so that the last rows of cells are empty, with the queries next to record 0."""
    rng = np.random.default_rng(n * 7 + int(r * 100))
    b, m, ns = 3, 400, 32
    t = rng.random((b, n, 1)) ** 0.5                       # dense at the thick end
    xyz1 = np.concatenate([10 * t, 1.5 * t * rng.random((b, n, 1)), 1.5 * t * rng.random((b, n, 1))], -1).astype(np.float32)
    # record 0 in the grid's LAST cell (max x, max y, max z corner is empty for most draws: the wedge's cross-section is a square
    # whose far corner few points reach), queries scattered around record 0 and around the far corner
    xyz1[:, 0] = xyz1.max(1) - np.float32(0.05)
    xyz2 = (xyz1[:, :1] + (rng.random((b, m, 3)).astype(np.float32) - 0.5) * np.float32(2 * r)).astype(np.float32)
    xyz2[:, m // 2:] = xyz1[:, rng.integers(0, n, m - m // 2)][0][None]
    want_idx, want_cnt = O.query_ball_point(r, ns, xyz1, xyz2)
    idx, cnt = P.tf_grouping.query_ball_point(r, ns, dev(xyz1), dev(xyz2))
    np.testing.assert_array_equal(cnt.cpu().numpy(), want_cnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), want_idx)


def test_ball_radius_boundary(P):
    # points at distances straddling sqrt: radius 0.25 exactly representable, lattice of 1/8
    xyz1 = clouds(9, 1, 2000, "lattice")
    for r in (0.125, 0.25, 0.375, np.float32(0.21650635), np.float32(0.2165064)):
        want_idx, want_cnt = O.query_ball_point(float(r), 32, xyz1, xyz1[:, :200])
        idx, cnt = P.tf_grouping.query_ball_point(float(r), 32, dev(xyz1), dev(xyz1[:, :200]))
        np.testing.assert_array_equal(cnt.cpu().numpy(), want_cnt)
        np.testing.assert_array_equal(idx.cpu().numpy(), want_idx)


@pytest.mark.parametrize("b,n,c,m,ns", [(4, 512, 64, 128, 64), (2, 1024, 3, 512, 32), (2, 512, 131, 128, 64),
                                        (3, 100, 6, 7, 1), (2, 300, 8, 11, 5)])
def test_group_point_and_grad(P, b, n, c, m, ns):
    rng = np.random.default_rng(c)
    pts = rng.random((b, n, c), dtype=np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    got = P.tf_grouping.group_point(dev(pts), dev(idx)).cpu().numpy()
    np.testing.assert_array_equal(got, O.group_point(pts, idx))
    x = dev(pts).requires_grad_(True)
    g = rng.random((b, m, ns, c), dtype=np.float32)
    P.tf_grouping.group_point(x, dev(idx)).backward(dev(g))
    np.testing.assert_array_equal(x.grad.cpu().numpy(), O.group_point_grad(pts, idx, g))


@pytest.mark.parametrize("b,n,m,k,kind", [
    (4, 1024, 512, 32, "ball"), (4, 512, 128, 64, "ball"), (2, 1024, 512, 32, "lattice"),
    (1, 8192, 1024, 32, "cube"), (1, 10240, 700, 32, "ball"), (2, 40, 40, 32, "ball"), (2, 64, 32, 64, "cube"),
    (2, 300, 33, 1, "cube"), (1, 500, 20, 100, "cube"), (1, 700, 9, 200, "lattice"), (2, 1024, 1024, 16, "ball"),
])
def test_knn_batch(P, b, n, m, k, kind):
    sup = clouds(31, b, n, kind)
    qry = sup[:, :m].copy() if m <= n else clouds(32, b, m, kind)
    # the canonical (distance, index) order against the C oracle
    want, wd = O.knn_batch(sup, qry, k, return_dist=True)
    got = P.nearest_neighbors.knn_batch(dev(sup), dev(qry), k, omp=True, tie_order="index")
    assert got.dtype == torch.int64
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    got32 = P.nearest_neighbors.knn_batch(dev(sup), dev(qry), k, dtype=torch.int32, tie_order="index")
    assert got32.dtype == torch.int32
    np.testing.assert_array_equal(got32.cpu().numpy(), want.astype(np.int32))
    # the DEFAULT: the reference's result, ties included -- against the reference library itself where it travelled with the tree
    # (oracle/_ref/libref_knn.so = knn_.cxx + nanoflann), and against the canonical list wherever no two distances are equal
    from oracle import ref
    if k <= 64:
        stats = []
        dflt = P.nearest_neighbors.knn_batch(dev(sup), dev(qry), k, omp=True, stats=stats)
        assert dflt.dtype == torch.int64
        if ref.available("libref_knn.so"):
            np.testing.assert_array_equal(dflt.cpu().numpy(), ref.knn_batch(sup, qry, k))
        if kind != "lattice":
            np.testing.assert_array_equal(dflt.cpu().numpy(), want)
        # numpy in -> numpy int64 out, like the reference binding
        got_np = P.nearest_neighbors.knn_batch(sup, qry, k)
        assert isinstance(got_np, np.ndarray) and got_np.dtype == np.int64
        np.testing.assert_array_equal(got_np, dflt.cpu().numpy())
    else:  # K > 64: lists of several registers per lane (insertion kernel's flags, wave-per-query tree search)
        dflt = P.nearest_neighbors.knn_batch(dev(sup), dev(qry), k, omp=True)
        if ref.available("libref_knn.so"):
            np.testing.assert_array_equal(dflt.cpu().numpy(), ref.knn_batch(sup, qry, k))
        if kind != "lattice":
            np.testing.assert_array_equal(dflt.cpu().numpy(), want)


def test_knn_self_first(P):
    # AdaptiveSampling relies on neighbour 0 being the query itself (pointasnl_util.py:162-163)
    sup = clouds(41, 2, 1024)
    got = P.nearest_neighbors.knn_batch(dev(sup), dev(sup[:, :512]), 32).cpu().numpy()
    np.testing.assert_array_equal(got[:, :, 0], np.broadcast_to(np.arange(512), (2, 512)))


@pytest.mark.parametrize("b,m,n,k", [(32, 128, 512, 64), (2, 10, 100, 100), (2, 7, 33, 5), (1, 4, 700, 16)])
def test_select_top_k(P, b, m, n, k):
    rng = np.random.default_rng(n)
    dist = rng.random((b, m, n), dtype=np.float32)
    dist[:, :, ::7] = np.round(dist[:, :, ::7] * 4) / 4  # ties
    wi, wo = O.select_top_k(k, dist)
    gi, go = P.tf_grouping.select_top_k(k, dev(dist))
    np.testing.assert_array_equal(go.cpu().numpy(), wo)
    np.testing.assert_array_equal(gi.cpu().numpy(), wi)


def test_knn_point(P):
    xyz1, xyz2 = clouds(51, 4, 512, "cube"), clouds(52, 4, 128, "cube")
    wv, wi = O.knn_point(16, xyz1, xyz2)
    val, idx = P.tf_grouping.knn_point(16, dev(xyz1), dev(xyz2))
    np.testing.assert_array_equal(idx.cpu().numpy(), wi)
    np.testing.assert_array_equal(val.cpu().numpy(), wv)


@pytest.mark.parametrize("b,n,m,kind", [(32, 512, 128, "cube"), (2, 8192, 1024, "ball"), (2, 1280, 320, "lattice"),
                                        (2, 64, 32, "ball"), (2, 10, 2, "cube"), (1, 5000, 3000, "cube")])
def test_three_nn(P, b, n, m, kind):
    x1, x2 = clouds(61, b, n, kind), clouds(62, b, m, kind)
    wd, wi = O.three_nn(x1, x2)
    d, i = P.tf_interpolate.three_nn(dev(x1), dev(x2))
    np.testing.assert_array_equal(i.cpu().numpy(), wi)
    np.testing.assert_array_equal(d.cpu().numpy(), wd)


@pytest.mark.parametrize("n,m", [(70, 1), (70, 2), (70, 3), (70, 5), (130, 17), (64, 18), (200, 16), (33, 1281), (257, 4095)])
def test_three_nn_ragged_quarters_and_ties(P, n, m):
    """The four waves of a workgroup scan quarters of the known cloud (rounded up to 4 points) and merge: every split raggedness,
    fewer known points than neighbours asked for (+inf / index 0 like the reference), and equal distances on either side of a
    quarter boundary (every known point duplicated 4x at shuffled positions: the lower index has to win), plus inf / nan
    coordinates (a nan distance fails every compare: skipped)."""
    rng = np.random.default_rng(n * 7 + m)
    base = rng.random((2, max(1, (m + 3) // 4), 3), dtype=np.float32)
    x2 = np.concatenate([base] * 4, axis=1)[:, :m]
    x2 = np.stack([x2[c][rng.permutation(m)] for c in range(2)])
    x1 = rng.random((2, n, 3), dtype=np.float32)
    x1[:, : min(n, m)] = x2[:, : min(n, m)]  # zero distances too
    for a, b in ((x1, x2),):
        wd, wi = O.three_nn(a, b)
        d, i = P.tf_interpolate.three_nn(dev(a), dev(b))
        np.testing.assert_array_equal(i.cpu().numpy(), wi)
        np.testing.assert_array_equal(d.cpu().numpy(), wd)
    if m >= 5:
        x2b = x2.copy()
        x2b[0, 1, 0] = np.inf
        x2b[1, m - 1, 2] = np.nan
        x2b[1, 0, 1] = -np.inf
        wd, wi = O.three_nn(x1, x2b)
        d, i = P.tf_interpolate.three_nn(dev(x1), dev(x2b))
        np.testing.assert_array_equal(i.cpu().numpy(), wi)
        np.testing.assert_array_equal(d.cpu().numpy(), wd)


@pytest.mark.parametrize("b,m,c,n", [(32, 128, 64, 512), (2, 1024, 128, 8192), (2, 64, 512, 256), (2, 50, 7, 33)])
def test_three_interpolate_and_grad(P, b, m, c, n):
    rng = np.random.default_rng(m)
    pts = rng.random((b, m, c), dtype=np.float32)
    x1, x2 = clouds(71, b, n, "cube"), clouds(72, b, m, "cube")
    d, i = O.three_nn(x1, x2)
    w = O.three_weights(d)
    gw = P.tf_interpolate.three_weights(dev(d)).cpu().numpy()
    np.testing.assert_array_equal(gw, w)
    got = P.tf_interpolate.three_interpolate(dev(pts), dev(i), dev(w)).cpu().numpy()
    np.testing.assert_allclose(got, O.three_interpolate(pts, i, w), rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(got, O.three_interpolate(pts, i, w))  # same op order, no contraction -> identical
    x = dev(pts).requires_grad_(True)
    g = rng.random((b, n, c), dtype=np.float32)
    P.tf_interpolate.three_interpolate(x, dev(i), dev(w)).backward(dev(g))
    np.testing.assert_array_equal(x.grad.cpu().numpy(), O.three_interpolate_grad(pts, i, w, g))


@pytest.mark.parametrize("b,m,c2,n,c1", [(8, 1280, 128, 10240, 32), (2, 80, 512, 320, 256), (2, 50, 7, 33, 5), (3, 40, 64, 17, 0),
                                          (1, 64, 12, 100, 6), (2, 128, 256, 512, 128)])
def test_fp_interpolate_cat_is_weights_interpolate_concat(P, b, m, c2, n, c1):
    """pasnl_fp_interpolate_cat = three_weights -> three_interpolate -> tf.concat (pointnet_util.py:212-219) in one launch:
    the oracle's values bit for bit (same operations, same order), rows that do not fill a tile of 16, channel counts that
    are not multiples of 4 (scalar path), no points1 (the decoding layer's use)."""
    rng = np.random.default_rng(m + c1)
    p2 = rng.standard_normal((b, m, c2)).astype(np.float32)
    p1 = rng.standard_normal((b, n, c1)).astype(np.float32) if c1 else None
    x1, x2 = clouds(73, b, n, "cube"), clouds(74, b, m, "cube")
    x1[0, :3] = x2[0, :3]  # zero distances: the 1e-10 floor of the weights
    d, i = O.three_nn(x1, x2)
    want = O.three_interpolate(p2, i, O.three_weights(d))
    if c1:
        want = np.concatenate([want, p1], axis=2)
    got = P.tf_interpolate.fp_interpolate_cat(dev(p2), dev(i), dev(d), dev(p1) if c1 else None).cpu().numpy()
    assert got.shape == (b, n, c2 + c1)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("b,n,m", [(2, 100, 10), (3, 8192 + 1000, 2048), (2, 5, 100), (1, 20000, 64)])
def test_prob_sample(P, b, n, m):
    rng = np.random.default_rng(n)
    p = rng.random((b, n), dtype=np.float32)
    r = rng.random((b, m), dtype=np.float32)
    got = P.tf_sampling.prob_sample(dev(p), dev(r)).cpu().numpy()
    np.testing.assert_array_equal(got, O.prob_sample(p, r))


def test_errors(P):
    x = dev(clouds(1, 2, 16))
    with pytest.raises(ValueError, match="positive npoint"):
        P.tf_sampling.farthest_point_sample(0, x)
    with pytest.raises(ValueError, match="FarthestPointSample expects"):
        P.tf_sampling.farthest_point_sample(4, x[:, :, :2])
    with pytest.raises(ValueError, match="positive radius"):
        P.tf_grouping.query_ball_point(0.0, 4, x, x)
    with pytest.raises(ValueError, match="positive nsample"):
        P.tf_grouping.query_ball_point(0.1, 0, x, x)
    with pytest.raises(ValueError, match="positive k"):
        P.tf_grouping.select_top_k(0, torch.zeros(1, 2, 3).cuda())
    with pytest.raises(ValueError):
        P.nearest_neighbors.knn_batch(x, x, 17)  # k > n
    with pytest.raises(Exception, match="no CPU fallback|CPU torch tensor"):
        P.tf_sampling.farthest_point_sample(4, x.cpu())


def test_empty(P):
    x = dev(clouds(1, 2, 16))
    e = torch.zeros((2, 0, 3), device="cuda")
    idx, cnt = P.tf_grouping.query_ball_point(0.1, 4, x, e)
    assert idx.shape == (2, 0, 4) and cnt.shape == (2, 0)
    assert P.nearest_neighbors.knn_batch(x, e, 3).shape == (2, 0, 3)
    d, i = P.tf_interpolate.three_nn(e, x)
    assert d.shape == (2, 0, 3)
    assert P.tf_grouping.group_point(x, torch.zeros((2, 0, 5), dtype=torch.int32, device="cuda")).shape == (2, 0, 5, 3)


@pytest.mark.parametrize("b,n,c,m,k", [(4, 1024, 3, 512, 32), (2, 512, 128, 128, 64), (2, 64, 256, 32, 32), (1, 50, 7, 50, 5),
                                       (2, 300, 300, 9, 3), (1, 40, 512, 40, 32), (1, 33, 700, 5, 4)])
def test_sa_group(P, b, n, c, m, k):
    # fused gather + centre + concat + max-over-K == the reference's op-by-op composition
    # (pointasnl_util.py:63-74, 248-249, 258), bit for bit (gathers, one subtraction, max)
    from pointasnl_amd.utils import pointasnl_util as U

    rng = np.random.default_rng(c)
    xyz = clouds(5, b, n)
    feat = rng.standard_normal((b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, k)).astype(np.int32)
    new_xyz = clouds(6, b, m)
    gx = O.group_point(xyz, idx)
    want = np.concatenate([gx - new_xyz[:, :, None, :], gx, O.group_point(feat, idx)], axis=-1)
    new_point, skip = U.sa_group(dev(xyz), dev(feat), dev(idx), dev(new_xyz))
    np.testing.assert_array_equal(new_point.cpu().numpy(), want)
    np.testing.assert_array_equal(skip.cpu().numpy(), want.max(axis=2))


@pytest.mark.parametrize("b,p,ns,c", [(64, 1, 512, 512), (3, 1, 128, 1024), (2, 5, 7, 33), (1, 1, 1, 1), (2, 3, 1000, 70)])
def test_max_pool_points(P, b, p, ns, c):
    # pointnet_util.py:137: tf.reduce_max(new_points, axis=[2], keep_dims=True) -- bit-exact (a maximum has no rounding)
    from pointasnl_amd.utils import pointnet_util as PU

    x = np.random.default_rng(c).standard_normal((b, p, ns, c)).astype(np.float32)
    got = PU.max_pool_points(dev(x)).cpu().numpy()
    np.testing.assert_array_equal(got, x.max(axis=2, keepdims=True))


def test_deterministic_grads_long_lists_and_atomic_variants(P):
    """Heavily repeated targets (lists far longer than one wave: the 64-smallest-at-a-time path), a cloud with an
    untouched row (gradient must be an exact 0), and the atomic variants (same value up to summation order)."""
    from pointasnl_amd import _hip

    rng = np.random.default_rng(3)
    b, n, c, m, ns = 2, 50, 70, 40, 33
    pts = rng.standard_normal((b, n, c)).astype(np.float32)
    idx = rng.integers(0, 4, (b, m, ns)).astype(np.int32)  # 1320 contributions onto 4 rows
    idx[1, :5] = 49
    g = rng.standard_normal((b, m, ns, c)).astype(np.float32)
    want = O.group_point_grad(pts, idx, g)
    for det in (True, False):
        _hip.DETERMINISTIC_GRADS = det
        try:
            x = dev(pts).requires_grad_(True)
            P.tf_grouping.group_point(x, dev(idx)).backward(dev(g))
        finally:
            _hip.DETERMINISTIC_GRADS = True
        got = x.grad.cpu().numpy()
        if det:
            np.testing.assert_array_equal(got, want)
            x2 = dev(pts).requires_grad_(True)
            P.tf_grouping.group_point(x2, dev(idx)).backward(dev(g))
            np.testing.assert_array_equal(x2.grad.cpu().numpy(), got)  # run-to-run identical
        else:
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-3)
    assert (want[0, 10] == 0).all()


@pytest.mark.parametrize("n,dl,fdim,ldim,kind", [(200000, 0.04, 3, 1, "room"), (5000, 0.1, 0, 0, "cube"), (3000, 0.5, 6, 2, "cube"),
                                                 (1, 0.1, 2, 1, "cube"), (4096, 10.0, 1, 1, "cube"), (10000, 0.06, 0, 1, "neg")])
def test_grid_subsampling(P, n, dl, fdim, ldim, kind):
    """pasnl_grid_subsample through the reference's module interface (cpp_subsampling.compute) vs the C restatement of
    grid_subsampling.cpp:4-106: same voxels in the same (ascending-key) order, bit-equal barycentres and feature means
    (fp32 sums in input order), majority labels with the smallest-label tie rule (random labels: ties do occur)."""
    from pointasnl_amd.utils.cpp_wrappers.cpp_subsampling import grid_subsampling as cpp_subsampling

    rng = np.random.default_rng(n + fdim)
    p = rng.random((n, 3)).astype(np.float32)
    if kind == "room":
        p = (p * np.array([6.0, 4.0, 2.5], dtype=np.float32)).astype(np.float32)
        p[::3, 2] = 0.0  # a floor: dense voxels with hundreds of members
    elif kind == "neg":
        p = (p * 4 - 2).astype(np.float32)
    f = rng.random((n, fdim)).astype(np.float32) if fdim else None
    c = rng.integers(0, 4, (n, ldim)).astype(np.int32) if ldim else None
    want = O.grid_subsample(p, f, c, dl)
    got = cpp_subsampling.compute(p, features=f, classes=c, sampleDl=dl, verbose=0)
    want = want if isinstance(want, tuple) else (want,)
    got = got if isinstance(got, tuple) else (got,)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)


@pytest.mark.parametrize("b,n,nq,k,kind", [(3, 700, 90, 16, "ball"), (2, 1500, 400, 8, "cube"), (1, 64, 200, 5, "ball"),
                                          (2, 5000, 64, 32, "ball"), (1, 300, 50, 12, "lattice")])
def test_knn_batch_distance_pick(b, n, nq, k, kind):
    """nearest_neighbors.knn_batch_distance_pick (knn.pyx:111-148): picks and neighbour lists equal the oracle's (itself
    equal to the reference's cpp_knn_batch_distance_pick with the seed pinned: tests/test_oracle_golden.py); numpy in ->
    numpy out like the reference binding; a fixed seed reproduces, another seed differs."""
    import pointasnl_amd as P

    x = clouds(900 + n, b, n, kind)
    want_i, want_q = O.knn_batch_distance_pick(x, nq, k, seed=77)
    got_i, got_q = P.nearest_neighbors.knn_batch_distance_pick(torch.from_numpy(x).cuda(), nq, k, seed=77)
    np.testing.assert_array_equal(got_i.cpu().numpy(), want_i)
    np.testing.assert_array_equal(got_q.cpu().numpy(), want_q)
    hi, hq = P.nearest_neighbors.knn_batch_distance_pick(x, nq, k, omp=True, seed=77)
    assert isinstance(hi, np.ndarray) and hi.dtype == np.int64 and (hi == want_i).all() and (hq == want_q).all()
    other, _ = P.nearest_neighbors.knn_batch_distance_pick(x, nq, k, seed=78)
    assert (other != want_i).any()
