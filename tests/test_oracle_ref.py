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
