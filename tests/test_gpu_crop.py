"""GPU: the crop stage (pasnl_knn_crop through the C ABI; SURVEY 8(f) rank 4, semantic_kitti_dataset_grid.py:265-286)
against the oracle (pinned to sklearn in tests/test_oracle_crop.py), against sklearn itself, and chained
grid_subsample -> crop -> pointasnl_sem_seg_res with no host KD-tree."""
import numpy as np
import pytest
import torch

from oracle import ops as O
from test_oracle_crop import scan

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    from pointasnl_amd.SemanticKITTI import semantic_kitti_dataset_grid as G

    return G


def check(G, p, centres, ks, kcap=None, radius=0.0, batched_points=False):
    """one call for all crops; every row against the oracle: the selected indices (ascending), their float64 distances, the count"""
    b = len(centres)
    kd = torch.tensor(ks, dtype=torch.int32).cuda() if ks is not None else None
    idx, d2, cnt = G.select_batch(torch.from_numpy(p).cuda(), torch.from_numpy(np.asarray(centres, np.float32)).cuda(), k=kd,
                                  kcap=kcap, radius=radius, want_d2=True)
    idx, d2, cnt = idx.cpu().numpy(), d2.cpu().numpy(), cnt.cpu().numpy()
    for c in range(b):
        pc = p[c] if batched_points else p
        want, wd2 = O.radius_crop(pc, centres[c], radius) if radius > 0 else O.knn_crop(pc, centres[c], min(ks[c], idx.shape[1]) if ks is not None else idx.shape[1])
        assert cnt[c] == len(want), (c, cnt[c], len(want))
        m = min(len(want), idx.shape[1])
        np.testing.assert_array_equal(idx[c, :m], want[:m])
        np.testing.assert_array_equal(d2[c, :m], wd2[:m])


@pytest.mark.parametrize("seed,n,snapped", [(0, 100000, False), (1, 2048, False), (2, 2049, False), (3, 7, False), (4, 1, False),
                                             (5, 60000, True), (6, 131072, True), (7, 300, True)])
def test_knn_crop_vs_oracle(G, seed, n, snapped):
    p = scan(100 + seed, n, snapped)
    rng = np.random.default_rng(seed)
    ks = sorted({1, 2, n // 7 + 1, n // 2 + 1, n - 1, n, min(n, 12800 + 17)} - {0})
    centres = [p[rng.integers(0, n)] for _ in ks]
    centres[0] = (p.mean(0) + 0.013).astype(np.float32)   # a centre that is not a point of the scan
    check(G, p, centres, ks, kcap=n)


def test_knn_crop_k_out_of_range_and_default(G):
    p = scan(31, 5000)
    check(G, p, [p[1], p[2], p[3]], [0, -5, 10 ** 6], kcap=5000)   # empty, empty, everything
    check(G, p, [p[9]], None, kcap=1300)                           # k = NULL: kcap nearest
    check(G, p, [p[9], p[10]], [2000, 700], kcap=1300)             # k clamped to the row length


def test_knn_crop_duplicates_and_nonfinite(G):
    p = scan(32, 4000)
    p[100:600] = p[100]                 # 500 copies of one point: a tie of 500 at distance 0 from it
    p[7] = np.inf                       # d2 = inf; p[8]: d2 = NaN (sorts behind inf, as the bits do)
    p[8, 0] = np.nan
    idx, d2, cnt = G.select_batch(torch.from_numpy(p).cuda(), torch.from_numpy(p[100:101].copy()).cuda(), k=250, want_d2=True)
    np.testing.assert_array_equal(idx[0].cpu().numpy(), np.arange(100, 350))   # the lowest indices of the tie
    idx, _, cnt = G.select_batch(torch.from_numpy(p).cuda(), torch.from_numpy(p[100:101].copy()).cuda(), k=3999)
    assert int(cnt[0]) == 3999 and 8 not in set(idx[0].cpu().numpy().tolist()) and 7 in set(idx[0].cpu().numpy().tolist())


def test_knn_crop_per_crop_scans(G):
    b, n = 5, 30000
    p = np.stack([scan(200 + i, n, snapped=i % 2 == 1) for i in range(b)])
    centres = [p[i, 17 * i + 3] for i in range(b)]
    check(G, p, centres, [12800, 1, 13433, 29999, 10240], kcap=n, batched_points=True)


@pytest.mark.parametrize("seed,r,snapped", [(0, 3.0, False), (1, 0.5, False), (2, 80.0, False), (3, 0.06 * 25, True), (4, 1e-4, False)])
def test_radius_crop_vs_oracle(G, seed, r, snapped):
    p = scan(300 + seed, 70000, snapped)
    check(G, p, [p[11], p[999], (p[5] + 0.01).astype(np.float32)], None, radius=r)


def test_radius_crop_truncated_row_reports_true_count(G):
    p = scan(41, 20000)
    want, _ = O.radius_crop(p, p[3], 10.0)
    idx, _, cnt = G.select_batch(torch.from_numpy(p).cuda(), torch.from_numpy(p[3:4].copy()).cuda(), kcap=100, radius=10.0)
    assert int(cnt[0]) == len(want) > 100
    np.testing.assert_array_equal(idx[0].cpu().numpy(), want[:100])


def test_device_scan_is_the_sklearn_tree(G):
    """DeviceScan.query / query_radius against sklearn.neighbors.KDTree on the same scan: indices AND distances, in sklearn's
    order (tie-free scan); on a voxel-snapped scan the distances and everything strictly inside the k-th distance."""
    KDTree = pytest.importorskip("sklearn.neighbors").KDTree
    p = scan(51, 90000)
    t, s = KDTree(p), G.DeviceScan(p)
    for pick, k in ((5, 12800), (77, 1), (1234, 13439)):
        c = p[pick].reshape(1, -1)
        d, i = t.query(c, k=k)
        gd, gi = s.query(c, k=k)
        np.testing.assert_array_equal(gi, i)
        np.testing.assert_array_equal(gd, d)
        assert gi.dtype == i.dtype and gd.dtype == d.dtype and gi.shape == i.shape
    for r in (0.3, 5.0):
        want = t.query_radius(p[9].reshape(1, -1), r=r)
        got = s.query_radius(p[9].reshape(1, -1), r=r)
        assert got.shape == want.shape and got.dtype == want.dtype
        np.testing.assert_array_equal(got[0], np.sort(want[0]))
    with pytest.raises(ValueError):
        s.query(p[:1], k=len(p) + 1)
    ps = scan(52, 50000, snapped=True)
    d, i = KDTree(ps).query(ps[3].reshape(1, -1), k=9000)
    gd, gi = G.DeviceScan(ps).query(ps[3].reshape(1, -1), k=9000)
    np.testing.assert_array_equal(gd, d)
    inside = i[0][d[0] < d[0][-1]]
    assert set(inside.tolist()) <= set(gi[0].tolist())


def test_crop_pc_equals_reference_lines_on_sklearn(G):
    """crop_pc on the DeviceScan == the reference's lines on an sklearn tree under the same numpy RNG stream: the same crop,
    order included (k form, tie-free scan); the radius form as a set (sklearn lists a radius query in tree order)."""
    KDTree = pytest.importorskip("sklearn.neighbors").KDTree
    p = scan(61, 80000)
    labels = (np.arange(len(p)) % 19).astype(np.uint8)
    tree, dev = KDTree(p), G.DeviceScan(p)
    num_point, num_buffer = 10240, 2560
    for pick in (0, 4321):
        rs = np.random.RandomState(pick)
        buffer = num_buffer + rs.randint(0, num_buffer // 4)
        sel = tree.query(p[pick].reshape(1, -1), k=num_point + buffer)[1][0]
        perm = np.arange(len(sel))
        rs.shuffle(perm)
        want = sel[perm][:num_point]
        gp, gl, gi = G.crop_pc(p, labels, dev, pick, num_point, num_buffer, rng=np.random.RandomState(pick))
        np.testing.assert_array_equal(gi, want)
        np.testing.assert_array_equal(gp, p[want])
        np.testing.assert_array_equal(gl, labels[want])
    gp, gl, gi = G.crop_pc(p, labels, dev, 50, num_point, num_buffer, in_radius=4.0, rng=np.random.RandomState(1))
    inside = tree.query_radius(p[50].reshape(1, -1), r=4.0)[0]
    assert gp.shape == (num_point, 3) and set(gi.tolist()) <= set(inside.tolist())
    assert len(inside) >= num_point or set(gi.tolist()) == set(inside.tolist())   # fewer than num_point inside: all of them, padded


def test_input_stage_chain_without_host_tree(G):
    """grid_subsampling -> crop -> pointasnl_sem_seg_res on the device: raw scan in, logits out, no KD-tree anywhere; the crop
    equals the oracle's on the subsampled scan."""
    from pointasnl_amd.models import pointasnl_sem_seg_res
    from pointasnl_amd.utils import tf_util
    from pointasnl_amd.utils.cpp_wrappers.cpp_subsampling import grid_subsampling

    rng = np.random.default_rng(9)
    raw = scan(71, 400000)
    sub = grid_subsampling.compute(raw, sampleDl=0.06)
    np.testing.assert_array_equal(sub, O.grid_subsample(raw, sampleDl=0.06))
    assert len(sub) > 20000
    dev = G.DeviceScan(sub)
    labels = np.zeros(len(sub), np.uint8)
    num_point = 10240
    batch = []
    for pick in rng.integers(0, len(sub), 2):
        pts, _, sel = G.crop_pc(sub, labels, dev, int(pick), num_point, 2560, rng=np.random.RandomState(int(pick)))
        want, _ = O.knn_crop(sub, sub[pick], num_point + 2560 + np.random.RandomState(int(pick)).randint(0, 640))
        assert set(sel.tolist()) <= set(want.tolist()) and len(set(sel.tolist())) == num_point
        batch.append(pts - sub[pick])
    x = torch.from_numpy(np.stack(batch).astype(np.float32)).cuda()
    tf_util.set_store(tf_util.VariableStore(seed=5))
    with torch.no_grad():
        out = pointasnl_sem_seg_res.get_model(x, False, 20, feature_channel=0)
    logits = out[0] if isinstance(out, (tuple, list)) else out
    assert logits.shape[:2] == (2, num_point) and bool(torch.isfinite(logits).all())
