"""CPU: the oracle against the reference's own C++ compiled into oracle/_ref (present where /root/reference was
available at build time; the .so files travel to the GPU box with the tree).  Random sweeps beyond the fixtures."""
import numpy as np
import pytest

from conftest import clouds
from oracle import ops as O
from oracle import ref

pytestmark = pytest.mark.skipif(not (ref.available("libref_knn.so") and ref.available("libref_interp.so")),
                                reason="oracle/_ref not built")


@pytest.mark.parametrize("seed", range(6))
def test_knn_random(seed):
    rng = np.random.default_rng(seed)
    b, n = int(rng.integers(1, 4)), int(rng.integers(40, 1500))
    m, k = int(rng.integers(1, n + 1)), int(rng.integers(1, 40))
    sup = clouds(1000 + seed, b, n, "ball")
    qry = clouds(2000 + seed, b, m, "ball")
    np.testing.assert_array_equal(O.knn_batch(sup, qry, k), ref.knn_batch(sup, qry, k, omp=bool(seed % 2)))


def test_knn_lattice_same_distances_possibly_different_tie_order():
    # exact duplicates / equal distances: nanoflann's order is traversal dependent (documented deviation)
    sup = clouds(77, 2, 600, "lattice")
    qry = sup[:, :100].copy()
    a, b = O.knn_batch(sup, qry, 16), ref.knn_batch(sup, qry, 16)
    da = ((sup[np.arange(2)[:, None, None], a] - qry[:, :, None, :]) ** 2).sum(-1)
    db = ((sup[np.arange(2)[:, None, None], b] - qry[:, :, None, :]) ** 2).sum(-1)
    np.testing.assert_array_equal(np.sort(da, -1), np.sort(db, -1))


@pytest.mark.parametrize("seed", range(4))
def test_three_nn_interpolate_random(seed):
    rng = np.random.default_rng(seed)
    b, n, m, c = 2, int(rng.integers(1, 3000)), int(rng.integers(3, 700)), int(rng.integers(1, 70))
    x1, x2 = clouds(seed, b, n, "cube"), clouds(seed + 9, b, m, "lattice" if seed % 2 else "cube")
    d, i = O.three_nn(x1, x2)
    rd, ri = ref.three_nn(x1, x2)
    np.testing.assert_array_equal(i, ri)
    np.testing.assert_array_equal(d, rd)
    pts = rng.random((b, m, c), dtype=np.float32)
    w = O.three_weights(d)
    np.testing.assert_array_equal(O.three_interpolate(pts, i, w), ref.three_interpolate(pts, i, w))


@pytest.mark.parametrize("n,dl,fdim,ldim", [(20000, 0.1, 3, 1), (5000, 0.04, 0, 0), (3000, 0.5, 6, 2), (1, 0.1, 2, 1)])
def test_grid_subsample_oracle_vs_reference(n, dl, fdim, ldim):
    """oracle_grid_subsample vs the reference's own grid_subsampling.cpp (oracle/_ref/libref_gridsub.so): identical
    voxel sets, bit-equal barycentres / feature means; labels are a function of the voxel here (no vote ties, whose
    outcome the reference leaves to its hash table)."""
    from oracle import ops, ref

    if not ref.available("libref_gridsub.so"):
        pytest.skip("oracle/_ref/libref_gridsub.so not built (needs /root/reference)")
    rng = np.random.default_rng(n)
    p = (rng.random((n, 3)) * np.array([3.0, 2.0, 1.0]) - 0.7).astype(np.float32)
    f = rng.random((n, fdim)).astype(np.float32) if fdim else None
    vox = np.floor(p / np.float32(dl)).astype(np.int64)
    c = np.stack([(vox[:, 0] * 7 + vox[:, 1] * 3 + vox[:, 2] + l) % 5 for l in range(ldim)], 1).astype(np.int32) if ldim else None
    a = ops.grid_subsample(p, f, c, dl)
    b = ref.grid_subsample(p, f, c, dl)
    a = a if isinstance(a, tuple) else (a,)
    b = b if isinstance(b, tuple) else (b,)
    assert a[0].shape == b[0].shape
    ka, kb = np.lexsort(a[0].T[::-1]), np.lexsort(b[0].T[::-1])
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x[ka], y[kb])
