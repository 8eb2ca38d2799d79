import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The oracle is compiled on demand; the HIP library must already be built (it travels with the tree)."""
    import oracle

    oracle.build()


def clouds(seed, b, n, kind="ball"):
    """Seeded synthetic clouds: 'ball' uniform in the unit ball, 'cube' uniform in [0,1)^3,
    'lattice' coordinates snapped to multiples of 1/8 (forces distance ties)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    if kind == "cube":
        return rng.random((b, n, 3), dtype=np.float32)
    if kind == "lattice":
        return (np.round(rng.random((b, n, 3)) * 8) / 8).astype(np.float32)
    v = rng.standard_normal((b, n, 3))
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    r = rng.random((b, n, 1)) ** (1 / 3)
    return (v * r).astype(np.float32)
