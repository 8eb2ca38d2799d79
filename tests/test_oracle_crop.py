"""CPU: the crop oracle (oracle.ops.knn_crop / radius_crop) pinned to sklearn's KDTree -- the third-party library the
reference's crop_pc calls (semantic_kitti_dataset_grid.py:269-271; the trees are built over float32 scans, which sklearn
stores as float64) -- and the host flow of crop_pc against a literal replay of the reference's lines on an sklearn tree."""
import numpy as np
import pytest

from oracle import ops as O

KDTree = pytest.importorskip("sklearn.neighbors").KDTree


def scan(seed, n, snapped=False):
    """a lidar-like scan: ground disc with 1/r density + vertical walls, metres; `snapped`: coordinates on a 0.06 m lattice
    (distance ties by construction)"""
    rng = np.random.default_rng(seed)
    r = 2.0 + 38.0 * rng.random(n) ** 2
    th = rng.random(n) * 2 * np.pi
    p = np.stack([r * np.cos(th), r * np.sin(th), rng.standard_normal(n) * 0.02], 1)
    w = n // 6
    p[:w, 2] = rng.random(w) * 2.0
    if snapped:
        p = np.round(p / 0.06) * 0.06
    return p.astype(np.float32)


@pytest.mark.parametrize("seed,n,k", [(0, 50000, 13000), (1, 20000, 1), (2, 3000, 2999), (3, 3000, 3000), (4, 100000, 12801)])
def test_knn_crop_equals_sklearn_query(seed, n, k):
    p = scan(seed, n)
    c = p[np.random.default_rng(seed).integers(0, n)]
    dist, ind = KDTree(p).query(c.reshape(1, -1), k=k)
    sel, d2 = O.knn_crop(p, c, k)
    assert sel.shape == (k,) and np.all(np.diff(sel) > 0)
    np.testing.assert_array_equal(np.sort(ind[0]), sel)                      # the same SET
    np.testing.assert_array_equal(np.sqrt(np.sort(d2)), dist[0])             # the same distances, bit for bit (float64)
    order = np.argsort(d2, kind="stable")
    np.testing.assert_array_equal(sel[order], ind[0])                        # tie-free: sklearn's order is (distance)


@pytest.mark.parametrize("seed", range(3))
def test_knn_crop_ties_same_distances_boundary_by_index(seed):
    p = scan(10 + seed, 30000, snapped=True)
    c = p[7]
    k = 5000
    dist, ind = KDTree(p).query(c.reshape(1, -1), k=k)
    sel, d2 = O.knn_crop(p, c, k)
    np.testing.assert_array_equal(np.sqrt(np.sort(d2)), dist[0])             # the multiset of distances is sklearn's
    all_d2 = O.crop_d2(p, c)
    kth = np.sort(d2)[-1]
    inside = np.nonzero(all_d2 < kth)[0]
    assert set(inside) <= set(sel) and set(inside) <= set(ind[0])            # everything strictly inside: in both
    on = np.nonzero(all_d2 == kth)[0]
    need = k - len(inside)
    np.testing.assert_array_equal(np.setdiff1d(sel, inside), on[:need])      # the boundary tie: lowest indices


@pytest.mark.parametrize("seed,r", [(0, 3.0), (1, 0.5), (2, 12.0), (3, 0.06 * 25), (4, 1e-3)])
def test_radius_crop_equals_sklearn_query_radius(seed, r):
    p = scan(20 + seed, 40000, snapped=seed == 3)   # seed 3: points EXACTLY on the radius (inclusive comparison)
    c = p[11]
    want = np.sort(KDTree(p).query_radius(c.reshape(1, -1), r=r)[0])
    sel, d2 = O.radius_crop(p, c, r)
    np.testing.assert_array_equal(sel, want)
    assert len(sel) >= 1  # the centre itself


class _SkScan:
    """the reference's search_tree, for the flow test"""

    def __init__(self, p):
        self.t = KDTree(p)

    def query(self, X, k=1):
        return self.t.query(X, k=k)

    def query_radius(self, X, r):
        return self.t.query_radius(X, r=r)


def test_crop_pc_flow_is_the_references(monkeypatch):
    """pointasnl_amd's crop_pc on an sklearn tree == the reference's lines replayed literally (same numpy RNG stream)."""
    from pointasnl_amd.SemanticKITTI import semantic_kitti_dataset_grid as G

    p = scan(5, 30000)
    labels = (np.arange(len(p)) % 19).astype(np.uint8)
    num_point, num_buffer, pick = 10240, 2560, 1234

    def reference_lines(rs, in_radius):
        center_point = p[pick, :].reshape(1, -1)
        tree = KDTree(p)
        if in_radius > 0:
            select_idx = tree.query_radius(center_point, r=in_radius)[0]
        else:
            buffer = num_buffer + rs.randint(0, num_buffer // 4)
            select_idx = tree.query(center_point, k=num_point + buffer)[1][0]
        idx = np.arange(len(select_idx))
        rs.shuffle(idx)
        select_idx = select_idx[idx][:num_point]
        if len(select_idx) < num_point:
            num_in = len(select_idx)
            dup = rs.choice(num_in, num_point - num_in)
            select_idx = select_idx[list(range(num_in)) + list(dup)]
        return p[select_idx], labels[select_idx], select_idx

    for in_radius in (0.0, 6.0):
        want = reference_lines(np.random.RandomState(3), in_radius)
        got = G.crop_pc(p, labels, _SkScan(p), pick, num_point, num_buffer, in_radius, rng=np.random.RandomState(3))
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a, b)
        assert got[0].shape == (num_point, 3)
