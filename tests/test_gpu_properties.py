"""MI355X, BASELINE.json's FULL sizes (cls 64x1024, ScanNet 16x8192, SemanticKITTI 8x10240): size-independent properties
of every op of the path, where the CPU oracle would take too long to compare element by element -- plus an oracle
comparison on a random sample of rows."""
import numpy as np
import pytest
import torch

from conftest import clouds
from oracle import ops as O

pytestmark = pytest.mark.gpu

CONFIGS = [("cls", 64, 1024, 512, 32), ("scannet", 16, 8192, 1024, 32), ("kitti", 8, 10240, 1280, 32)]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def P():
    import pointasnl_amd

    return pointasnl_amd


@pytest.mark.parametrize("name,b,n,m,k", CONFIGS)
def test_fps_gather_properties(P, name, b, n, m, k):
    xyz = clouds(11, b, n)
    idx = P.tf_sampling.farthest_point_sample(m, dev(xyz)).cpu().numpy()
    assert idx.shape == (b, m) and (idx[:, 0] == 0).all() and idx.min() >= 0 and idx.max() < n
    assert all(len(set(row)) == m for row in idx)  # tie-free clouds: no point is picked twice
    new_xyz = P.tf_sampling.gather_point(dev(xyz), dev(idx)).cpu().numpy()
    np.testing.assert_array_equal(new_xyz, xyz[np.arange(b)[:, None], idx])
    # greedy property: every pick maximises the distance to the picks before it (checked at a few prefixes)
    for j in (1, 2, m // 2, m - 1):
        d = ((xyz[:, :, None, :] - new_xyz[:, None, :j, :]) ** 2).sum(-1).min(-1)  # (b, n) distance to the first j picks
        assert (np.abs(d[np.arange(b), idx[:, j]] - d.max(axis=1)) <= 1e-6 * d.max(axis=1)).all()
    # bit-exact against the oracle on two clouds
    np.testing.assert_array_equal(idx[:2], O.farthest_point_sample(m, xyz[:2]))


@pytest.mark.parametrize("name,b,n,m,k", CONFIGS)
def test_knn_properties(P, name, b, n, m, k):
    xyz = clouds(12, b, n)
    q = xyz[:, :m].copy()
    idx = P.nearest_neighbors.knn_batch(dev(xyz), dev(q), k, dtype=torch.int32).cpu().numpy()
    assert idx.shape == (b, m, k) and idx.min() >= 0 and idx.max() < n
    assert (idx[:, :, 0] == np.arange(m)[None, :]).all()  # queries are support points: self first (AdaptiveSampling relies on it)
    d = ((xyz[np.arange(b)[:, None, None], idx] - q[:, :, None, :]) ** 2).sum(-1)
    assert (np.diff(d, axis=-1) >= -1e-7).all()  # ascending distance
    assert all(len(set(r)) == k for r in idx.reshape(-1, k)[:: max(1, b * m // 4096)])
    # exactness on a sample of queries: the k-th distance equals the brute-force k-th distance
    rng = np.random.default_rng(0)
    for bi, j in zip(rng.integers(0, b, 64), rng.integers(0, m, 64)):
        full = np.sort(((xyz[bi] - q[bi, j]) ** 2).sum(-1))[:k]
        np.testing.assert_allclose(d[bi, j], full, rtol=1e-5, atol=1e-9)
    np.testing.assert_array_equal(idx[:1, :64], O.knn_batch(xyz[:1], q[:1, :64], k).astype(np.int32))


@pytest.mark.parametrize("name,b,n,m,k", CONFIGS)
def test_ball_group_properties(P, name, b, n, m, k):
    xyz = clouds(13, b, n)
    q = xyz[:, :m].copy()
    r = 0.2 if n == 1024 else 0.07
    idx, cnt = P.tf_grouping.query_ball_point(r, k, dev(xyz), dev(q))
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    assert idx.min() >= 0 and idx.max() < n and cnt.min() >= 1 and cnt.max() <= k  # every query contains itself
    d = np.sqrt(((xyz[np.arange(b)[:, None, None], idx] - q[:, :, None, :]) ** 2).sum(-1))
    assert (d < r * (1 + 1e-6)).all()
    valid = np.arange(k)[None, None, :] < cnt[..., None]
    assert (np.where(valid[..., 1:], np.diff(idx, axis=-1), 1) > 0).all()  # the hits are in ascending index order
    assert (np.where(valid, 0, idx - idx[..., :1]) == 0).all()  # padding repeats the first hit
    oi, oc = O.query_ball_point(r, k, xyz[:1], q[:1])
    np.testing.assert_array_equal(idx[:1], oi)
    np.testing.assert_array_equal(cnt[:1], oc)
    pts = np.random.default_rng(1).standard_normal((b, n, 16)).astype(np.float32)
    g = P.tf_grouping.group_point(dev(pts), dev(idx)).cpu().numpy()
    np.testing.assert_array_equal(g, pts[np.arange(b)[:, None, None], idx])


@pytest.mark.parametrize("name,b,n,m,k", CONFIGS[1:])
def test_three_nn_interpolate_properties(P, name, b, n, m, k):
    xyz1, xyz2 = clouds(14, b, n), clouds(15, b, m)
    dist, idx = P.tf_interpolate.three_nn(dev(xyz1), dev(xyz2))
    w = P.tf_interpolate.three_weights(dist)
    dist, idx, wn = dist.cpu().numpy(), idx.cpu().numpy(), w.cpu().numpy()
    assert idx.min() >= 0 and idx.max() < m and (np.diff(dist, axis=-1) >= 0).all()
    np.testing.assert_allclose(wn.sum(-1), 1.0, rtol=0, atol=2e-6)
    pts = np.random.default_rng(2).standard_normal((b, m, 32)).astype(np.float32)
    out = P.tf_interpolate.three_interpolate(dev(pts), dev(idx), w).cpu().numpy()
    gathered = pts[np.arange(b)[:, None, None], idx]  # (b, n, 3, c): a convex combination stays inside the hull
    assert (out <= gathered.max(2) + 1e-5).all() and (out >= gathered.min(2) - 1e-5).all()
    od, oi = O.three_nn(xyz1[:1, :512], xyz2[:1])
    np.testing.assert_array_equal(idx[:1, :512], oi)
    np.testing.assert_array_equal(dist[:1, :512], od)


@pytest.mark.parametrize("b,p,n,cb", [(64, 512, 1024, 32), (16, 1024, 8192, 32), (8, 1280, 10240, 32)])
def test_nl_attention_properties(b, p, n, cb):
    """softmax(QK^T/sqrt(cb)) V at the full layer-1 shapes (the (B,P,N) map would be 128 / 512 / 400 MiB): rows are convex
    combinations of V; adding a constant to every V row shifts the output by it (linearity + weights summing to one);
    a sample of queries against the fp64 formula."""
    from pointasnl_amd.utils import pointasnl_util as U

    rng = np.random.default_rng(p)
    q = rng.standard_normal((b, p, cb)).astype(np.float32)
    kv = rng.standard_normal((b, n, 2 * cb)).astype(np.float32)
    out = U.nl_attention(dev(q), dev(kv)).cpu().numpy()
    v = kv[..., cb:]
    assert (out <= v.max(1, keepdims=True) + 1e-5).all() and (out >= v.min(1, keepdims=True) - 1e-5).all()
    kv2 = kv.copy()
    kv2[..., cb:] += 3.0
    out2 = U.nl_attention(dev(q), dev(kv2)).cpu().numpy()
    np.testing.assert_allclose(out2 - out, 3.0, rtol=0, atol=2e-5)
    for bi, j in zip(rng.integers(0, b, 16), rng.integers(0, p, 16)):
        s = (kv[bi, :, :cb].astype(np.float64) @ q[bi, j].astype(np.float64)) / np.sqrt(cb)
        w = np.exp(s - s.max())
        np.testing.assert_allclose(out[bi, j], (w / w.sum()) @ v[bi].astype(np.float64), rtol=1e-5, atol=1e-5)
