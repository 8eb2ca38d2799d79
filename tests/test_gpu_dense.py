"""csrc/dense.hip and the fused sampler+gather through the C ABI: the thin dense layers of the classifier head
(pasnl_dense_rows), the narrow projections of the first non-local cell (pasnl_narrow_project2), pooling into a wider
table (pasnl_max_pool_rows_strided) and pasnl_farthest_point_sample_gather.
Floating point: 1e-5 relative to the fp64 product (north_star tolerance); indices and gathers: bit-exact."""
import ctypes

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


def _dense_case(rows, k, n, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((rows, k)).astype(np.float32)
    w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, n).astype(np.float32)
    return x, w, b


@pytest.mark.parametrize("rows,k,n,relu", [
    (64, 1536, 512, True), (64, 512, 256, True), (64, 256, 40, False),   # fc1, fc2, fc3 of pointasnl_cls at B = 64
    (16, 1536, 512, True), (1, 8, 1, False), (33, 72, 31, True), (128, 2048, 100, True), (97, 4096, 512, False),
    (2, 16384, 256, True),                                                # a set-abstraction after_conv at B*P = 2
])
def test_dense_rows_matches_fp64(P, rows, k, n, relu):
    from pointasnl_amd.utils import tf_util
    x, w, b = _dense_case(rows, k, n, 100 + rows + n)
    want = x.astype(np.float64) @ w.astype(np.float64) + b
    if relu:
        want = np.maximum(want, 0)
    got = tf_util._dense_rows(dev(x), dev(w), dev(b), relu).cpu().numpy()
    assert got.shape == (rows, n) and got.dtype == np.float32
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 1e-5 * max(scale, 1.0)


@pytest.mark.parametrize("rows,k,n,relu", [
    (320, 16384, 512, True), (2560, 4096, 128, True), (640, 8192, 256, True), (4096, 384, 256, True),   # sem_seg(_res) deep layers
    (1024, 16480, 512, True), (130, 272, 132, False), (129, 64, 4, True), (257, 2048, 260, False), (3000, 256, 96, True),
])
def test_dense_splitk_matches_fp64(P, rows, k, n, relu):
    """pasnl_dense_splitk (128 x 128 tiles x K slices + ordered reduce): 1e-5 of the output scale against the fp64 product,
    ragged tiles in both directions, a row stride larger than K, one slice and many."""
    from pointasnl_amd.utils import tf_util
    x, w, b = _dense_case(rows, k, n, 300 + rows + n)
    want = x.astype(np.float64) @ w.astype(np.float64) + b
    if relu:
        want = np.maximum(want, 0)
    xd = dev(np.concatenate([x, np.full((rows, 8), 7.0, np.float32)], axis=1))[:, :k]   # lda = k + 8, not contiguous
    got = tf_util._dense_splitk(xd, dev(w), dev(b), relu)
    again = tf_util._dense_splitk(xd, dev(w), dev(b), relu)
    assert torch.equal(got, again), "slice-ordered sums are bit-reproducible"
    got = got.cpu().numpy()
    assert got.shape == (rows, n) and got.dtype == np.float32
    assert np.abs(got - want).max() <= 1e-5 * max(np.abs(want).max(), 1.0)


def test_dense_layer_takes_the_splitk_kernel_for_thin_long_products(P, monkeypatch):
    """tf_util._dense routes the products pasnl_dense_splitk was measured to win (one column block of tiles, thousands of rows,
    K >= 4096) to it and leaves the rest to the vendor library."""
    from pointasnl_amd.utils import tf_util
    from pointasnl_amd import _hip
    launched = []
    real = _hip.launch
    monkeypatch.setattr(_hip, "launch", lambda sym, *a: (launched.append(sym), real(sym, *a))[1])
    tf_util.set_store(tf_util.VariableStore(seed=3))
    for rows, k, n, expect in [(2560, 4096, 128, True), (320, 16384, 512, False), (131072, 128, 128, False), (640, 100, 64, False)]:
        del launched[:]
        x = torch.randn(rows, k, device="cuda")
        out = tf_util._dense(x, n, f"probe_{rows}_{k}", False, "relu")
        st = tf_util.store()
        with tf_util.variable_scope(f"probe_{rows}_{k}"):
            w, b = st.layer(k, n, False, None)
        want = torch.relu(torch.addmm(b.double(), x.double(), w.double()))
        assert ("pasnl_dense_splitk" in launched) == expect, (rows, k, n, launched)
        assert float((out.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("rows,k,n,relu,bias", [
    (32768, 512, 128, True, True), (4096, 2048, 256, True, True), (130, 96, 128, False, True), (129, 32, 384, True, False),
    (1000, 4192, 128, True, True), (257, 544, 256, False, False),
])
def test_dense_bf16x3_matches_fp64(P, rows, k, n, relu, bias):
    """pasnl_dense_bf16x3 (every fp32 operand as three bf16 terms, six products on the bf16 matrix pipe, fp32 accumulation):
    1e-5 of the output scale against the fp64 product -- the same contract as the fp32 chain, not the same bits --, ragged row
    tiles, a row stride larger than K, with and without bias / activation; the same bits on every call."""
    from pointasnl_amd.utils import tf_util
    tf_util.set_store(tf_util.VariableStore(seed=5))
    x, w, b = _dense_case(rows, k, n, 700 + rows + n)
    want = x.astype(np.float64) @ w.astype(np.float64) + (b if bias else 0.0)
    if relu:
        want = np.maximum(want, 0)
    xd = dev(np.concatenate([x, np.full((rows, 4), 7.0, np.float32)], axis=1))[:, :k]   # lda = k + 4
    wd, bd = dev(w), (dev(b) if bias else None)
    got = tf_util._dense_bf16x3(xd, wd, bd, relu)
    assert torch.equal(got, tf_util._dense_bf16x3(xd, wd, bd, relu))
    got = got.cpu().numpy()
    assert got.shape == (rows, n) and got.dtype == np.float32
    assert np.abs(got - want).max() <= 1e-5 * max(np.abs(want).max(), 1.0)


def test_dense_bf16x3_split_is_exact_to_24_bits(P):
    """The three bf16 planes of the weights add up to the fp32 value within 2^-24 relative (hi + mid + lo carries 24 bits),
    in the matrix-instruction operand order [plane][k / 8][n][k % 8]."""
    from pointasnl_amd import _hip
    k, n = 64, 128
    w = torch.randn(k, n, device="cuda") * torch.logspace(-6, 6, n, device="cuda")
    ws = torch.empty(int(_hip.lib().pasnl_bf16x3_weights_bytes(k, n)), dtype=torch.uint8, device="cuda")
    _hip.launch("pasnl_bf16x3_split_weights", "bf16x3_split", k, n, _hip.ptr(w), _hip.ptr(ws))
    planes = ws.view(torch.bfloat16).view(3, k // 8, n, 8).permute(0, 1, 3, 2).reshape(3, k, n).double()
    back = planes[0] + planes[1] + planes[2]
    assert float(((back - w.double()).abs() / w.double().abs().clamp_min(1e-30)).max()) <= 2.0 ** -23


def test_dense_layer_takes_bf16x3_only_when_switched_on(P, monkeypatch):
    """The product mode is opt-in (tf_util.DENSE_BF16X3) and only for products with a 128 x 128 tile per CU."""
    from pointasnl_amd.utils import tf_util
    from pointasnl_amd import _hip
    launched = []
    real = _hip.launch
    monkeypatch.setattr(_hip, "launch", lambda sym, *a: (launched.append(sym), real(sym, *a))[1])
    tf_util.set_store(tf_util.VariableStore(seed=4))
    x = torch.randn(32768, 1024, device="cuda")
    tf_util._dense(x, 128, "bx_off", False, "relu")
    assert "pasnl_dense_bf16x3" not in launched
    monkeypatch.setattr(tf_util, "DENSE_BF16X3", True)
    out = tf_util._dense(x, 128, "bx_on", False, "relu")
    assert "pasnl_dense_bf16x3" in launched
    with tf_util.variable_scope("bx_on"):
        w, b = tf_util.store().layer(1024, 128, False, None)
    want = torch.relu(torch.addmm(b.double(), x.double(), w.double()))
    assert float((out.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    del launched[:]
    tf_util._dense(torch.randn(4096, 1024, device="cuda"), 128, "bx_small", False, "relu")
    assert "pasnl_dense_bf16x3" not in launched, "32 tiles: the vendor library's product"


def test_dense_rows_is_reproducible_and_leaves_its_counters_clean(P):
    # the K slices are summed in slice order by whichever workgroup arrives last: every run gives the same bits, and the
    # counters are back at zero for the next launch (50 back-to-back launches on one workspace)
    from pointasnl_amd.utils import tf_util
    x, w, b = _dense_case(64, 1536, 512, 7)
    xd, wd, bd = dev(x), dev(w), dev(b)
    first = tf_util._dense_rows(xd, wd, bd, True).clone()
    for _ in range(50):
        again = tf_util._dense_rows(xd, wd, bd, True)
    torch.cuda.synchronize()
    assert torch.equal(first, again)


def test_dense_layer_routes_thin_products_to_the_hip_kernel(P):
    # tf_util.fully_connected on one row per cloud: same numbers with the kernel and with the vendor GEMM
    from pointasnl_amd.utils import tf_util
    x = dev(_dense_case(64, 1536, 512, 3)[0])
    outs = []
    for flag in (True, False):
        tf_util.set_store(tf_util.VariableStore(seed=11, randomize_bn=True))
        tf_util.DENSE_ROWS = flag
        try:
            outs.append(tf_util.fully_connected(x, 512, bn=True, is_training=False, scope='fc1'))
        finally:
            tf_util.DENSE_ROWS = True
    assert torch.allclose(outs[0], outs[1], rtol=1e-5, atol=1e-5)


def test_dense_rows_errors(P):
    from pointasnl_amd import _hip
    lib = _hip.lib()
    x, w, b = (dev(a) for a in _dense_case(8, 16, 32, 1))
    out = torch.empty((8, 32), device="cuda")
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    st = _hip.stream_ptr()
    args = lambda rows, k, n, wsb: (rows, k, n, _hip.ptr(x), _hip.ptr(w), _hip.ptr(b), 1, _hip.ptr(out), _hip.ptr(ws),
                                    ctypes.c_size_t(wsb), st)
    assert lib.pasnl_dense_rows(*args(8, 16, 32, ws.numel())) == 0
    assert lib.pasnl_dense_rows(*args(8, 12, 32, ws.numel())) == -5      # K not a multiple of 8
    assert lib.pasnl_dense_rows(*args(129, 16, 32, ws.numel())) == -5    # too many rows for this kernel
    assert lib.pasnl_dense_rows(*args(8, 16, 32, 16)) == -3              # workspace too small
    assert lib.pasnl_dense_rows(*args(-1, 16, 32, ws.numel())) == -1
    assert lib.pasnl_dense_rows(*args(0, 16, 32, 0)) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("rows0,k0,n0,rows1,k1,n1", [
    (64 * 1024, 3, 64, 64 * 512, 6, 32),   # conv_kv / conv_query of pointasnl_cls layer1 at B = 64
    (16 * 8192, 3, 64, 16 * 1024, 6, 32),  # ScanNet layer1
    (777, 16, 256, 5, 1, 128), (300, 9, 32, 0, 0, 0),
])
def test_narrow_project2(P, rows0, k0, n0, rows1, k1, n1):
    from pointasnl_amd import _hip
    rng = np.random.default_rng(rows0 + n0)
    jobs = []
    for rows, k, n in ((rows0, k0, n0), (rows1, k1, n1)):
        if rows == 0:
            jobs.append(None)
            continue
        x = rng.standard_normal((rows, k)).astype(np.float32)
        w = rng.standard_normal((k, n)).astype(np.float32)
        b = rng.uniform(-0.5, 0.5, n).astype(np.float32)
        jobs.append((x, w, b, dev(x), dev(w), dev(b), torch.empty((rows, n), device="cuda")))
    a = []
    for (rows, k, n), j in zip(((rows0, k0, n0), (rows1, k1, n1)), jobs):
        a += [ctypes.c_long(rows), k, n] + ([_hip.ptr(j[3]), _hip.ptr(j[4]), _hip.ptr(j[5]), _hip.ptr(j[6])] if j else
                                            [_hip.ptr(None)] * 4)
    _hip.launch("pasnl_narrow_project2", "narrow_project", *a)
    for j in jobs:
        if j is None:
            continue
        want = j[0].astype(np.float64) @ j[1].astype(np.float64) + j[2]
        got = j[6].cpu().numpy()
        assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_non_local_cell_with_narrow_projections_matches_the_gemm_path(P):
    from pointasnl_amd.utils import tf_util, pointasnl_util as U
    rng = np.random.default_rng(5)
    feature = dev(rng.standard_normal((3, 700, 3)).astype(np.float32))
    new_point = dev(rng.standard_normal((3, 1, 260, 6)).astype(np.float32))
    outs = []
    for flag in (True, False):
        tf_util.set_store(tf_util.VariableStore(seed=21, randomize_bn=True))
        U.NL_NARROW_PROJECT = flag
        try:
            outs.append(U.PointNonLocalCell(feature, new_point, [32, 128], False, None, None, 'layer1', bn=True))
        finally:
            U.NL_NARROW_PROJECT = True
    assert outs[0].shape == (3, 260, 128)
    assert torch.allclose(outs[0], outs[1], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("b,n,m,kind", [
    (3, 1024, 512, "ball"), (2, 512, 128, "ball"), (2, 600, 77, "cube"), (1, 37, 5, "lattice"), (2, 2048, 300, "lattice"),
    (1, 8192, 1024, "cube"), (1, 10240, 1280, "ball"), (2, 1, 1, "ball"),
])
def test_fps_gather_is_fps_then_gather(P, b, n, m, kind):
    xyz = clouds(31 + n, b, n, kind)
    want_idx = O.farthest_point_sample(m, xyz)
    idx, new_xyz = P.tf_sampling.farthest_point_sample_gather(m, dev(xyz))
    np.testing.assert_array_equal(idx.cpu().numpy(), want_idx)
    np.testing.assert_array_equal(new_xyz.cpu().numpy(), O.gather_point(xyz, want_idx))


def test_max_pool_into_a_wider_table(P):
    from pointasnl_amd.utils import pointnet_util as PU
    rng = np.random.default_rng(9)
    a = rng.standard_normal((5, 1, 77, 96)).astype(np.float32)
    c = rng.standard_normal((5, 1, 300, 40)).astype(np.float32)
    table = torch.full((5, 136), float("nan"), device="cuda")
    ra = PU.max_pool_points(dev(a), out=table[:, :96])
    rc = PU.max_pool_points(dev(c), out=table[:, 96:])
    want = np.concatenate([a.max(axis=2)[:, 0], c.max(axis=2)[:, 0]], axis=1)
    np.testing.assert_array_equal(table.cpu().numpy(), want)
    assert ra.shape == (5, 1, 1, 96) and rc.shape == (5, 1, 1, 40)
    np.testing.assert_array_equal(ra.cpu().numpy()[:, 0, 0], a.max(axis=2)[:, 0])
    with pytest.raises(ValueError):
        PU.max_pool_points(dev(a), out=table[:, :95])
