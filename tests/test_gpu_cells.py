"""AS / NL cells and the classification graph on the HIP path vs the numpy oracle (fp32, cross-checked in fp64).
Tolerance 1e-5 (BASELINE.json north_star) on the attention cores; the end-to-end logits accumulate ~10 layers
of fp32 GEMMs in a different summation order (vendor BLAS vs numpy) and are held to 1e-4 relative."""
import numpy as np
import pytest
import torch

from conftest import clouds
from oracle import cells

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("b,p,n,cb", [
    (4, 512, 1024, 32),   # cls layer1
    (4, 128, 512, 64),    # cls layer2
    (1, 1024, 8192, 32),  # ScanNet layer1 (512 MiB map if materialised at B=16)
    (2, 32, 64, 128),     # ScanNet layer4
    (2, 45, 77, 32),      # ragged: P, N not multiples of the tiles
    (1, 1, 1, 64),
    (2, 40, 80, 128),     # KITTI layer4_1
])
def test_nl_attention(b, p, n, cb, variant):
    from pointasnl_amd.utils import pointasnl_util as U

    if variant == 1 and cb == 128:
        pytest.skip("vector-FMA variant covers cb<=64")
    rng = np.random.default_rng(p * 7 + n)
    q = rng.standard_normal((b, p, cb)).astype(np.float32)
    kv = rng.standard_normal((b, n, 2 * cb)).astype(np.float32)
    want64 = cells.nl_attention_core(q.astype(np.float64), kv.astype(np.float64), cb)
    want32 = cells.nl_attention_core(q, kv, cb)
    got = U.nl_attention(dev(q), dev(kv), variant=variant).cpu().numpy()
    assert np.abs(want32 - want64).max() < 1e-5  # the fp32 oracle itself is within tolerance of fp64
    np.testing.assert_allclose(got, want64, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("b,p,n", [
    (16, 512, 1024),   # cls layer 1 at a batch that reaches the two-tile kernel (128 pairs of tiles)
    (3, 2800, 256),    # the last pair: a full tile and a tile of 16 queries
    (2, 4100, 96),     # the last pair: 4 queries and an empty tile; 3 key blocks for up to 8 waves (waves without a block)
    (130, 64, 32),     # one key block
    (4, 2048, 8192),   # long key loop
    (3, 2800, 250),    # a ragged key block: the dispatcher has to fall back to the one-tile kernel
])
def test_nl_attention_two_tiles_per_wave(b, p, n):
    """cb = 32 with >= 128 pairs of 32-query tiles runs nl_attention_pair_kernel (two tiles per wave, softmax pieces in the
    shadow of the other tile's products): the same 1e-5 contract, ragged query counts, waves that see no key block, and a key
    that dominates late (the rescale path of both tiles)."""
    from pointasnl_amd.utils import pointasnl_util as U

    cb = 32
    rng = np.random.default_rng(p + n)
    q = rng.standard_normal((b, p, cb)).astype(np.float32)
    kv = rng.standard_normal((b, n, 2 * cb)).astype(np.float32)
    kv[0, n - 3, :cb] = q[0, min(40, p - 1)] * 5.0   # a late spike for a query of the SECOND tile
    kv[b - 1, n // 2, :cb] = q[b - 1, 1] * 5.0
    sel = [0, b - 1] if b > 2 else list(range(b))    # the fp64 oracle on two clouds is enough (and keeps the test short)
    got = U.nl_attention(dev(q), dev(kv), variant=2).cpu().numpy()
    assert np.isfinite(got).all()
    want = cells.nl_attention_core(q[sel].astype(np.float64), kv[sel].astype(np.float64), cb)
    np.testing.assert_allclose(got[sel], want, rtol=1e-5, atol=1e-5)
    # every cloud against the vector-FMA kernel (an independent implementation on the same inputs)
    ref = U.nl_attention(dev(q), dev(kv), variant=1).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("b,p,n", [
    (8, 1280, 10240),  # SemanticKITTI layer 1_1: 160 workgroups for 256 CUs
    (4, 1024, 8192),   # a ScanNet layer 1 at a small batch
    (2, 1280, 10240),
    (1, 100, 4096),    # a ragged pair of query tiles, one cloud: 2 workgroups -> the largest number of parts
    (3, 70, 4160),     # 130 key blocks: parts of unequal length, the last one short
])
def test_nl_attention_keys_over_workgroups(b, p, n):
    """cb = 32 with too few query tiles for the chip: pasnl_nl_attention_ws splits the KEYS over workgroups too and combines the
    parts in ascending key order.  Same 1e-5 contract against fp64 (a late spike in the LAST part and one in the first: the
    cross-part rescale in both directions), bit-identical from call to call (fixed merge order), and the plain form agrees."""
    import ctypes

    from pointasnl_amd import _hip
    from pointasnl_amd.utils import pointasnl_util as U

    cb = 32
    rng = np.random.default_rng(p + n)
    q = rng.standard_normal((b, p, cb)).astype(np.float32)
    kv = rng.standard_normal((b, n, 2 * cb)).astype(np.float32)
    kv[0, n - 3, :cb] = q[0, min(40, p - 1)] * 5.0
    kv[b - 1, 2, :cb] = q[b - 1, 1] * 5.0
    assert U.NL_KEY_PARTS
    nbytes = int(_hip.lib().pasnl_nl_attention_workspace_bytes(b, p, n, cb))
    assert nbytes > 0, "these shapes are the ones the partitioned form exists for"
    got = U.nl_attention(dev(q), dev(kv))
    again = U.nl_attention(dev(q), dev(kv))
    assert torch.equal(got, again)
    sel = sorted({0, b - 1})
    want = cells.nl_attention_core(q[sel].astype(np.float64), kv[sel].astype(np.float64), cb)
    np.testing.assert_allclose(got.cpu().numpy()[sel], want, rtol=1e-5, atol=1e-5)
    plain = U.nl_attention(dev(q), dev(kv), variant=2)   # an explicit variant: the one-workgroup form
    np.testing.assert_allclose(got.cpu().numpy(), plain.cpu().numpy(), rtol=2e-5, atol=2e-5)
    # the C ABI refuses a workspace that is too small before anything is launched
    out = torch.empty((b, p, cb), dtype=torch.float32, device="cuda")
    ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
    with pytest.raises(_hip.PasnlError):
        _hip.launch("pasnl_nl_attention_ws", "nl", b, p, n, cb, _hip.ptr(dev(q)), _hip.ptr(dev(kv)), _hip.ptr(out), 0, _hip.ptr(ws),
                    ctypes.c_size_t(nbytes - 1))


def test_nl_attention_full_shapes_do_not_take_the_partitioned_form():
    from pointasnl_amd import _hip

    for b, p, n, cb in [(64, 512, 1024, 32), (16, 1024, 8192, 32), (8, 80, 320, 64), (8, 1280, 10250, 32), (8, 320, 1280, 32)]:
        assert int(_hip.lib().pasnl_nl_attention_workspace_bytes(b, p, n, cb)) == 0


@pytest.mark.parametrize("b,p,n,cb", [(2, 128, 512, 64), (2, 45, 77, 32)])
def test_nl_attention_lds_staged_kernel_still_agrees(b, p, n, cb):
    """cb <= 64 runs the kernel that takes its operands straight from global memory; the LDS-staged one (the only one
    for cb = 128) stays selectable (variant 3) and must give the same answer."""
    from pointasnl_amd.utils import pointasnl_util as U

    rng = np.random.default_rng(n)
    q = rng.standard_normal((b, p, cb)).astype(np.float32)
    kv = rng.standard_normal((b, n, 2 * cb)).astype(np.float32)
    want = cells.nl_attention_core(q.astype(np.float64), kv.astype(np.float64), cb)
    got = U.nl_attention(dev(q), dev(kv), variant=3).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_nl_attention_spike():
    # one key dominates from a late tile: forces the online-softmax rescale branch with a large max jump
    from pointasnl_amd.utils import pointasnl_util as U

    rng = np.random.default_rng(5)
    b, p, n, cb = 1, 64, 300, 32
    q = rng.standard_normal((b, p, cb)).astype(np.float32)
    kv = rng.standard_normal((b, n, 2 * cb)).astype(np.float32) * 0.1
    kv[0, 250, :cb] = q[0, 3] * 6.0
    want = cells.nl_attention_core(q.astype(np.float64), kv.astype(np.float64), cb)
    for variant in (1, 2):
        got = U.nl_attention(dev(q), dev(kv), variant=variant).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("g,as_,cb", [(2048, 12, 32), (512, 12, 65), (300, 8, 32), (100, 4, 32), (7, 16, 40), (3, 1, 33)])
def test_as_attention(g, as_, cb):
    from pointasnl_amd.utils import pointasnl_util as U

    rng = np.random.default_rng(g)
    q = rng.standard_normal((g, as_, cb)).astype(np.float32)
    kv = rng.standard_normal((g, as_, 2 * cb)).astype(np.float32)
    want = cells.nl_attention_core(q.astype(np.float64), kv.astype(np.float64), cb)
    got = U.as_attention(dev(q), dev(kv)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def _store(seed):
    from pointasnl_amd.utils import tf_util

    return tf_util.set_store(tf_util.VariableStore(seed=seed, randomize_bn=True))


@pytest.mark.parametrize("as_,c", [(12, 3), (12, 128), (8, 32), (4, 64)])
def test_adaptive_sampling_cell(as_, c):
    from pointasnl_amd.utils import pointasnl_util as U

    st = _store(as_ * 100 + c)
    rng = np.random.default_rng(c)
    b, p, k = 2, 96, 32
    gxyz = rng.standard_normal((b, p, k, 3)).astype(np.float32) * 0.2
    gfeat = np.concatenate([gxyz, rng.standard_normal((b, p, k, c)).astype(np.float32)], -1)
    nx, nf = U.AdaptiveSampling(dev(gxyz), dev(gfeat), as_, False, None, None, "layer1", True)
    params = st.export_numpy()
    wx, wf = cells.adaptive_sampling(gxyz.astype(np.float64), gfeat.astype(np.float64), as_, params, "layer1")
    np.testing.assert_allclose(nx.cpu().numpy(), wx, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(nf.cpu().numpy(), wf, rtol=1e-5, atol=1e-5)
    wx32, wf32 = cells.adaptive_sampling(gxyz, gfeat, as_, params, "layer1")
    np.testing.assert_allclose(wx32, wx, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,c,p,cq", [(1024, 3, 512, 6), (512, 128, 128, 131), (256, 256, 64, 259)])
def test_point_nonlocal_cell(n, c, p, cq):
    from pointasnl_amd.utils import pointasnl_util as U

    st = _store(n + c)
    rng = np.random.default_rng(n)
    b = 2
    feat = rng.standard_normal((b, n, c)).astype(np.float32)
    newf = rng.standard_normal((b, p, cq)).astype(np.float32)
    mlp = [max(32, c // 2), 2 * max(32, c // 2)]
    got = U.PointNonLocalCell(dev(feat), dev(newf).unsqueeze(1), mlp, False, None, None, "layerX", True)
    want = cells.point_nonlocal_cell(feat.astype(np.float64), newf.astype(np.float64), mlp, st.export_numpy(), "layerX")
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("adaptive", [False, True])
def test_cls_forward_matches_oracle(adaptive):
    from pointasnl_amd.models import pointasnl_cls

    st = _store(17)
    pc = clouds(99, 2, 1024)
    with torch.no_grad():
        logits, ep = pointasnl_cls.get_model(dev(pc), is_training=False, adaptive_sample=adaptive)
    params = st.export_numpy()
    want, wep = cells.cls_forward(pc, params, adaptive_sample=adaptive)
    want64, _ = cells.cls_forward(pc, params, adaptive_sample=adaptive, dtype=np.float64)
    # sampled coordinates: without AS they are gathered input points -> exact
    if not adaptive:
        np.testing.assert_array_equal(ep["l1_xyz"].cpu().numpy(), wep["l1_xyz"])
    else:
        np.testing.assert_allclose(ep["l1_xyz"].cpu().numpy(), wep["l1_xyz"], rtol=1e-5, atol=1e-5)
    scale = max(1.0, np.abs(want64).max())
    assert np.abs(logits.cpu().numpy() - want64).max() / scale < 1e-4
    assert np.abs(want - want64).max() / scale < 1e-4
    assert (logits.argmax(1).cpu().numpy() == want64.argmax(1)).all()


@pytest.mark.parametrize("g,k,c,c1", [(64, 32, 3, 64), (16, 64, 128, 128), (8, 32, 64, 32), (5, 96, 30, 64), (3, 32, 3, 128)])
def test_sa_local_cell(g, k, c, c1):
    """Fused conv0 -> conv1, weight net, H2^T.G (pointasnl_util.py:264-274) vs the numpy restatement in fp64."""
    from pointasnl_amd.utils import pointasnl_util as U

    st = _store(g * 7 + c)
    rng = np.random.default_rng(k + c)
    w = 6 + c
    x = rng.standard_normal((1, g, k, w)).astype(np.float32)
    x[..., :3] *= 0.2
    with st.scope("L"):
        got = U.sa_local_cell(dev(x), [c1, c1, 2 * c1], False, None, None, True).cpu().numpy()
    p = st.export_numpy()
    x64 = x.astype(np.float64)
    h = cells._layer(cells._layer(x64, p["L/conv0"], "relu"), p["L/conv1"], "relu")
    wn = cells._layer(x64[..., :3], p["L/weight_net/wconv0"], "relu")
    want = np.swapaxes(h, 2, 3) @ wn  # (1,g,c1,32)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() / scale < 1e-5
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5 * scale)


@pytest.mark.parametrize("b,n,c,m,k,c1", [
    (2, 1024, 3, 512, 32, 64), (2, 512, 128, 128, 64, 128), (1, 256, 64, 37, 32, 32),   # the models' shapes (8-step tail)
    (1, 100, 30, 5, 96, 64), (3, 64, 3, 64, 32, 128), (1, 40, 250, 3, 32, 64),           # scalar loads, 3 tiles, 9 chunks
    (19, 96, 24, 33, 32, 32),    # row width 32: no partial chunk; 19 clouds: the XCD map with a ragged last round
    (17, 64, 56, 16, 32, 64),    # row width 64: two full chunks, no partial chunk
    (2, 128, 16, 40, 64, 64),    # partial chunk with 16 live steps (16-byte loads)
    (1, 77, 11, 9, 32, 32),      # partial chunk with 10 live steps (scalar loads)
    (1, 50, 4, 1, 32, 128)])     # one group: fewer groups than waves
def test_sa_cell_gather_fused(b, n, c, m, k, c1):
    """pasnl_sa_cell = grouping + skip max + local cell in one kernel: `out` vs the fp64 restatement of
    pointasnl_util.py:63-74,248-249,264-274 on the gathered rows; `skip` bit-equal to the gathered maximum (:258)."""
    from pointasnl_amd.utils import pointasnl_util as U

    st = _store(b * 11 + c)
    rng = np.random.default_rng(n + c)
    xyz = clouds(5, b, n)
    feat = rng.standard_normal((b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, k)).astype(np.int32)
    new_xyz = clouds(6, b, m)
    with st.scope("L"):
        got, skip = U.sa_cell(dev(xyz), dev(feat), dev(idx), dev(new_xyz), [c1, c1, 2 * c1], False, None, None,
                              True)
    bi = np.arange(b)[:, None, None]
    gx = xyz[bi, idx]
    x = np.concatenate([gx - new_xyz[:, :, None, :], gx, feat[bi, idx]], axis=-1)  # (b,m,k,6+c) float32, exact
    np.testing.assert_array_equal(skip.cpu().numpy(), x.max(axis=2))
    p = st.export_numpy()
    x64 = x.astype(np.float64)
    h = cells._layer(cells._layer(x64, p["L/conv0"], "relu"), p["L/conv1"], "relu")
    wn = cells._layer(x64[..., :3], p["L/weight_net/wconv0"], "relu")
    want = np.swapaxes(h, 2, 3) @ wn  # (b,m,c1,32)
    scale = np.abs(want).max()
    assert np.abs(got.cpu().numpy() - want).max() / scale < 1e-5
    # and it is the same function as the two-kernel path
    with st.scope("L"):
        np_, skip2 = U.sa_group(dev(xyz), dev(feat), dev(idx), dev(new_xyz))
        two = U.sa_local_cell(np_, [c1, c1, 2 * c1], False, None, None, True)
    np.testing.assert_array_equal(skip.cpu().numpy(), skip2.cpu().numpy())
    np.testing.assert_allclose(got.cpu().numpy(), two.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)


@pytest.mark.parametrize("b,n,m", [(2, 1024, 1024), (3, 300, 77), (1, 64, 1), (8, 2048, 500)])
def test_sa_cell_16_channels_native_kernel(b, n, m, monkeypatch):
    """mlp [16, 16, 32] on xyz-only rows, 32 neighbours (pointasnl_sem_seg_res.py:32): the 16x16x4-MFMA kernel against the fp64
    restatement, its skip maxima bit-equal to the gathered maximum, and the same function as the zero-padded 32-channel kernel."""
    from pointasnl_amd.utils import pointasnl_util as U

    st = _store(b * 7 + m)
    rng = np.random.default_rng(n + m)
    xyz = clouds(15, b, n)
    idx = rng.integers(0, n, (b, m, 32)).astype(np.int32)
    new_xyz = clouds(16, b, m)
    with st.scope("L"):
        got, skip = U.sa_cell(dev(xyz), dev(xyz), dev(idx), dev(new_xyz), [16, 16, 32], False, None, None, True)
    assert got.shape == (b, m, 16, 32) and got.is_contiguous()  # no padded channels behind the view
    bi = np.arange(b)[:, None, None]
    gx = xyz[bi, idx]
    x = np.concatenate([gx - new_xyz[:, :, None, :], gx, gx], axis=-1)
    np.testing.assert_array_equal(skip.cpu().numpy(), x.max(axis=2))
    p = st.export_numpy()
    x64 = x.astype(np.float64)
    h = cells._layer(cells._layer(x64, p["L/conv0"], "relu"), p["L/conv1"], "relu")
    wn = cells._layer(x64[..., :3], p["L/weight_net/wconv0"], "relu")
    want = np.swapaxes(h, 2, 3) @ wn
    scale = np.abs(want).max()
    assert np.abs(got.cpu().numpy() - want).max() / scale < 1e-5
    monkeypatch.setattr(U, "SA_CELL16", False)
    with st.scope("L"):
        padded, skip2 = U.sa_cell(dev(xyz), dev(xyz), dev(idx), dev(new_xyz), [16, 16, 32], False, None, None, True)
    np.testing.assert_array_equal(skip.cpu().numpy(), skip2.cpu().numpy())
    np.testing.assert_allclose(got.cpu().numpy(), padded.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)


@pytest.mark.parametrize("b,n,c,m,c1,conv1", [(2, 300, 256, 40, 256, True), (1, 128, 256, 33, 256, False), (2, 96, 512, 20, 512, False),
                                               (1, 64, 128, 7, 256, True), (1, 80, 512, 3, 512, True),
                                               (2, 320, 128, 320, 128, False), (1, 77, 64, 9, 128, False),
                                               (2, 300, 128, 77, 128, True), (1, 96, 64, 640, 128, True)])  # 128 channels, few groups
def test_sa_cell_wide_layers(b, n, c, m, c1, conv1, monkeypatch):
    """The 256- / 512-channel layers (pointasnl_sem_seg.py:34, pointasnl_sem_seg_res.py:46-51) on pasnl_sa_cell: one workgroup per
    group, weights from L2; with conv1 (mlp [c, c, out]) and without (mlp [c, c]: the *_2 layers).  fp64 restatement to 1e-5,
    skip maxima bit-equal; the same bits with the weights packed in operand order (pasnl_sa_cell_packed, the default) and
    row-major, with a centre table and with neighbour 0 as the centre."""
    from pointasnl_amd.utils import pointasnl_util as U
    from pointasnl_amd import _hip

    rng = np.random.default_rng(n + c + m)
    xyz = clouds(25, b, n)
    feat = rng.standard_normal((b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, 32)).astype(np.int32)
    new_xyz = clouds(26, b, m)
    mlp = [c1, c1, 2 * c1] if conv1 else [c1, c1]
    runs = {}
    for packed in (False, True):
        monkeypatch.setattr(U, "SA_CELL_PACKED", packed)
        st = _store(b * 13 + c1 + m)
        launched = []
        real = _hip.launch
        monkeypatch.setattr(_hip, "launch", lambda sym, *a: (launched.append(sym), real(sym, *a))[1])
        with st.scope("L"):
            runs[packed] = U.sa_cell(dev(xyz), dev(feat), dev(idx), dev(new_xyz), mlp, False, None, None, True)
            centre0 = U.sa_cell(dev(xyz), dev(feat), dev(idx), None, mlp, False, None, None, True) if m <= n and (c <= 128 or c1 >= 256) else None
        monkeypatch.setattr(_hip, "launch", real)
        assert ("pasnl_sa_cell_packed" in launched) == (packed and c % 16 == 0 and c >= 32)
        runs[packed] = (runs[packed], centre0)
    for a, b_ in zip(runs[False][0], runs[True][0]):
        assert torch.equal(a, b_)
    if runs[False][1] is not None:
        for a, b_ in zip(runs[False][1], runs[True][1]):
            assert torch.equal(a, b_)
    got, skip = runs[True][0]
    assert got.shape == (b, m, c1, 32)
    bi = np.arange(b)[:, None, None]
    gx = xyz[bi, idx]
    x = np.concatenate([gx - new_xyz[:, :, None, :], gx, feat[bi, idx]], axis=-1)
    np.testing.assert_array_equal(skip.cpu().numpy(), x.max(axis=2))
    p = st.export_numpy()
    x64 = x.astype(np.float64)
    hcell = cells._layer(x64, p["L/conv0"], "relu")
    if conv1:
        hcell = cells._layer(hcell, p["L/conv1"], "relu")
    wn = cells._layer(x64[..., :3], p["L/weight_net/wconv0"], "relu")
    want = np.swapaxes(hcell, 2, 3) @ wn
    assert np.abs(got.cpu().numpy() - want).max() / np.abs(want).max() < 1e-5


@pytest.mark.parametrize("b,n,c,m,c1,centre0", [(2, 400, 32, 100, 64, True), (2, 400, 32, 100, 64, False), (1, 300, 64, 40, 128, True),
                                                (2, 256, 16, 256, 32, False), (1, 200, 32, 50, 32, True), (3, 128, 64, 128, 64, False),
                                                (1, 90, 128, 17, 128, False)])
def test_sa_cell_single_convolution_equals_identity_conv1(b, n, c, m, c1, centre0, monkeypatch):
    """mlp = [c, c] (one convolution, the *_2 layers of pointasnl_sem_seg_res.py:36,41): the kernels skip conv1.  (1) Against the
    ORACLE: the fp64 restatement of pointasnl_util.py:63-74,248-249,264-274 with one convolution, 1e-5 of the output scale and
    elementwise rtol 1e-4 / atol 1e-5 of the scale, skip maxima bit-equal to the gathered maximum -- for the persistent kernel's
    SINGLE form at 32 / 64 / 128 channels, with a centre table and with the groups' neighbour 0 as centres.  (2) With
    SA_CELL_SINGLE off the same layer runs an identity conv1 -- relu(h * 1 + 0) = h -- and the outputs have to be the same BITS."""
    from pointasnl_amd.utils import pointasnl_util as U

    rng = np.random.default_rng(c1 + m)
    xyz = clouds(27, b, n)
    feat = rng.standard_normal((b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, 32)).astype(np.int32)
    centres = clouds(28, b, m)
    new_xyz = None if centre0 else dev(centres)
    outs = []
    for single in (True, False):
        monkeypatch.setattr(U, "SA_CELL_SINGLE", single)
        st = _store(77)
        with st.scope("L"):
            r = U.sa_cell(dev(xyz), dev(feat), dev(idx), new_xyz, [c1, c1], False, None, None, True)
        outs.append([t.cpu().numpy() for t in r[:2]])
        if single:
            p = st.export_numpy()
    bi = np.arange(b)[:, None, None]
    gx = xyz[bi, idx]
    cen = gx[:, :, 0, :] if centre0 else centres
    x = np.concatenate([gx - cen[:, :, None, :], gx, feat[bi, idx]], axis=-1)  # float32, exact
    np.testing.assert_array_equal(outs[0][1], x.max(axis=2))
    x64 = x.astype(np.float64)
    h = cells._layer(x64, p["L/conv0"], "relu")
    wn = cells._layer(x64[..., :3], p["L/weight_net/wconv0"], "relu")
    want = np.swapaxes(h, 2, 3) @ wn
    scale = np.abs(want).max()
    assert outs[0][0].shape == want.shape
    assert np.abs(outs[0][0] - want).max() / scale < 1e-5
    np.testing.assert_allclose(outs[0][0], want, rtol=1e-4, atol=1e-5 * scale)
    if c1 == 128 and c >= 32:
        # 128 channels: one or both forms run on the WIDE kernel (its single-convolution form with a centre table; its
        # two-convolution form for the identity when the groups are few) -- another summation order than the persistent kernel's:
        # the same function, not the same bits
        np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6 * scale)
    else:
        np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("b,n,c,mlp", [(3, 512, 128, [128, 256, 512]), (2, 128, 256, [256, 512, 1024]), (2, 77, 128, [128, 256, 512]),
                                        (1, 32, 256, [256, 512, 1024]), (5, 200, 60, [256, 512, 1024])])
def test_group_all_module_in_one_kernel(b, n, c, mlp, monkeypatch):
    """pointnet_sa_module(group_all=True) (pointnet_util.py:87-137; models/pointasnl_cls.py:39-40) on csrc/mlp_pool.hip: three
    convolutions + the maximum over the cloud's points in one kernel -- against the fp64 restatement of the module (elementwise
    rtol 1e-4 / atol 1e-5 of the output scale), with and without the producer's [0 | xyz | points] rows and the pooled_out view,
    ragged last tiles, and the same function as the layer-by-layer path on the vendor GEMM."""
    from pointasnl_amd import _hip
    from pointasnl_amd.utils import pointnet_util as PU

    rng = np.random.default_rng(b * 100 + n)
    xyz = clouds(31, b, n)
    pts = rng.standard_normal((b, n, c)).astype(np.float32)
    x64 = np.concatenate([xyz, pts], axis=-1).astype(np.float64)
    outs = {}
    for mode in ("fused", "fused_cat", "layers"):
        monkeypatch.setattr(PU, "GROUP_ALL_FUSED", (128, 256) if mode != "layers" else ())
        st = _store(5)
        p_t = dev(pts)
        x_t = dev(xyz)
        if mode == "fused_cat":  # rows as PointASNLSetAbstraction(xyz_concat=True) leaves them: one zero column in front
            p_t.xyz_concat = (x_t, torch.cat([torch.zeros((b, n, 1), device="cuda"), x_t, p_t], dim=2).contiguous())
        wide = torch.zeros((b, mlp[-1] + 8), device="cuda")
        _hip.PROFILE = []
        try:
            with torch.no_grad():
                _, got, _ = PU.pointnet_sa_module(x_t, p_t, npoint=None, radius=None, nsample=None, mlp=mlp, mlp2=None,
                                                  group_all=True, is_training=False, bn_decay=None, scope="L",
                                                  pooled_out=wide[:, 8:] if mode != "layers" else None)
            launched = [sym for sym, _, _, _ in _hip.PROFILE]
        finally:
            _hip.PROFILE = None
        # the producer's 16-byte rows take the fused kernel; the reference's own 3 + c wide rows (not a multiple of 4 floats)
        # and the switch take the layers one by one
        assert ("pasnl_mlp3_max_pool" in launched) == (mode == "fused_cat"), (mode, launched)
        outs[mode] = got.cpu().numpy()
        assert got.shape == (b, 1, mlp[-1])
        if mode != "layers":
            np.testing.assert_array_equal(wide[:, 8:].cpu().numpy(), outs[mode][:, 0])  # written in place, nothing beside it
            assert float(wide[:, :8].abs().max()) == 0.0
        params = st.export_numpy()
    hcell = x64[:, None]
    for i in range(3):
        hcell = cells._layer(hcell, params[f"L/conv{i}"], "relu")
    want = hcell.max(axis=2)  # (b, 1, c3)
    scale = np.abs(want).max()
    for mode in ("fused", "fused_cat", "layers"):
        assert np.abs(outs[mode] - want).max() / scale < 1e-5, mode
        np.testing.assert_allclose(outs[mode], want, rtol=1e-4, atol=1e-5 * scale)


def test_sa_cell_unaligned_weights_take_the_scalar_staging_path():
    """The C-ABI takes any float pointers: weights that are not 16-byte aligned are staged with dword copies and give
    bit-identical results."""
    from pointasnl_amd import _hip

    b, n, c, m, k, c1 = 2, 200, 64, 24, 32, 64
    g = torch.Generator(device="cuda").manual_seed(3)
    r = lambda *sh: torch.randn(sh, device="cuda", generator=g)
    xyz, feat, new_xyz = r(b, n, 3), r(b, n, c), r(b, m, 3)
    idx = torch.randint(0, n, (b, m, k), device="cuda", dtype=torch.int32, generator=g)
    ws = [r(6 + c, c1) * 0.1, r(c1) * 0.1, r(c1, c1) * 0.1, r(c1) * 0.1, r(3, 32), r(32)]

    def run(weights):
        out = torch.empty((b, m, c1, 32), device="cuda")
        skip = torch.empty((b, m, 6 + c), device="cuda")
        _hip.launch("pasnl_sa_cell", "sa_cell", b, n, c, m, k, c1, c1, _hip.ptr(xyz), _hip.ptr(feat), _hip.ptr(idx),
                    _hip.ptr(new_xyz), *[_hip.ptr(t) for t in weights], _hip.ptr(out), _hip.ptr(skip))
        return out, skip

    def misaligned(t):  # the same values, 4 bytes past a 16-byte boundary
        buf = torch.empty(t.numel() + 4, device="cuda")
        v = buf[1:1 + t.numel()].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 == 4
        return v

    out_a, skip_a = run(ws)
    out_u, skip_u = run([misaligned(t) for t in ws])
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_u) and torch.equal(skip_a, skip_u)


@pytest.mark.parametrize("model", ["sem_seg", "sem_seg_res"])
def test_seg_forward_matches_oracle(model):
    """ScanNet / SemanticKITTI graphs (configs 4-5) at N=4096 (smallest size where every level still has >= 32
    points): encoder + three_nn / three_interpolate decoders vs the numpy oracle."""
    import importlib

    M = importlib.import_module(f"pointasnl_amd.models.pointasnl_{model}")
    st = _store(23)
    # every level must keep >= nsample = 32 points: N/128 for sem_seg (4096), N/256 for sem_seg_res (8192)
    npts = 4096 if model == "sem_seg" else 8192
    pc = clouds(77, 1, npts)
    with torch.no_grad():
        logits, _ = M.get_model(dev(pc), False, 13)
    params = st.export_numpy()
    fwd = cells.sem_seg_forward if model == "sem_seg" else cells.sem_seg_res_forward
    want = fwd(pc, params, 13, dtype=np.float64)
    got = logits.cpu().numpy()
    assert got.shape == (1, npts, 13)
    scale = max(1.0, np.abs(want).max())
    assert np.abs(got - want).max() / scale < 2e-4
    assert (got.argmax(-1) == want.argmax(-1)).mean() > 0.999


@pytest.mark.parametrize("n,radius", [(1024, 0.07), (1280, 0.2)])
def test_repulsion_loss(n, radius):
    """get_repulsion_loss (pointasnl_util.py:361-378), the only in-model consumer of query_ball_point / group_point:
    ScanNet shape (l1_xyz 1024 pts, r=0.07) and SemanticKITTI shape (1280 pts, r=0.2), ns=20."""
    from pointasnl_amd.utils import pointasnl_util as U

    pred = clouds(41, 3, n)
    got = float(U.get_repulsion_loss(dev(pred), nsample=20, radius=radius))
    want = cells.repulsion_loss(pred, nsample=20, radius=radius)
    assert abs(got - want) < 1e-6 * max(1.0, abs(want))


@pytest.mark.parametrize("b,n,c,k", [(2, 1024, 128, 16), (1, 300, 61, 16), (16, 64, 512, 16), (1, 40, 5, 32), (3, 17, 256, 16)])
def test_decode_cell(b, n, c, k):
    """pasnl_decode_cell = gathers + centring + weight net + F^T.G of PointASNLDecodingLayer (pointasnl_util.py:323-331)
    vs the fp64 restatement; also against the op-by-op chain of the product."""
    from pointasnl_amd.utils import pointasnl_util as U

    st = _store(n + c)
    rng = np.random.default_rng(n * 3 + c)
    xyz = clouds(8, b, n)
    feat = rng.standard_normal((b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, (b, n, k)).astype(np.int32)
    with st.scope("D"):
        out = U.decode_cell(dev(xyz), dev(feat), dev(idx))
    if hasattr(out, "row_order"):  # the tiled order (c % 32 == 0, k = 16): back to the reference's (channel, j) order
        flat = torch.empty_like(out).reshape(b, n, -1)
        flat[:, :, out.row_order[1]] = out.reshape(b, n, -1)
        out = flat.reshape(out.shape)
    got = out.cpu().numpy()
    p = st.export_numpy()
    bi = np.arange(b)[:, None, None]
    gx = xyz[bi, idx].astype(np.float64)
    F = np.concatenate([gx, feat[bi, idx].astype(np.float64)], axis=-1)                 # (b,n,k,3+c)
    G = cells._layer(gx - xyz[:, :, None, :].astype(np.float64), p["D/decode_weight_net/wconv0"], "relu")  # (b,n,k,32)
    want = np.swapaxes(F, 2, 3) @ G
    assert got.shape == want.shape == (b, n, 3 + c, 32)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() / scale < 1e-5


@pytest.mark.parametrize("b,n,c,m,k,as_", [(2, 1024, 3, 512, 32, 12), (2, 512, 128, 128, 64, 12), (1, 300, 3, 77, 32, 8),
                                          (1, 200, 64, 50, 32, 4), (1, 64, 32, 9, 16, 16),
                                          (1, 64, 3, 9, 16, 16), (1, 50, 2, 7, 8, 1),   # narrow rows: full / single neighbour
                                          (3, 4000, 3, 3000, 16, 12),                    # more groups than resident waves
                                          (1, 80, 9, 11, 16, 5)])                        # the widest narrow row (15 columns)
@pytest.mark.parametrize("narrow_cell,proj", [(True, True), (False, True), (False, False)])
def test_adaptive_sampling_fused(b, n, c, m, k, as_, narrow_cell, proj, monkeypatch):
    """AdaptiveSampling + SampleWeights (pointasnl_util.py:112-173) without grouped tensors (as_gather + one [K|V|Q]
    GEMM + strided micro attention + re-weighting) vs the fp64 restatement on explicitly gathered groups."""
    from pointasnl_amd.utils import pointasnl_util as U

    # narrow layers have three implementations of the same cell: one kernel / attention with fused projections / GEMM first
    monkeypatch.setattr(U, "AS_CELL_NARROW", narrow_cell)
    monkeypatch.setattr(U, "AS_PROJ_FUSED", proj)
    if c > 9:  # wide rows: one kernel after the projection GEMM / the few-kernel chain
        if not proj:
            pytest.skip("wide rows have two implementations")
        monkeypatch.setattr(U, "AS_CELL_WIDE", narrow_cell)
    st = _store(n + c + as_)
    rng = np.random.default_rng(c * 5 + as_)
    xyz = clouds(12, b, n)
    feat = rng.standard_normal((b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, k)).astype(np.int32)
    with st.scope("layerA"):
        new_xyz, new_feat = U.adaptive_sampling_fused(dev(xyz), dev(feat), dev(idx), as_, "layerA", True)
    bi = np.arange(b)[:, None, None]
    gx = xyz[bi, idx].astype(np.float64)
    gf = np.concatenate([gx, feat[bi, idx].astype(np.float64)], axis=-1)
    want_xyz, want_feat = cells.adaptive_sampling(gx, gf, as_, st.export_numpy(), "layerA", outer="layerA")
    np.testing.assert_allclose(new_xyz.cpu().numpy(), want_xyz, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(new_feat.cpu().numpy(), want_feat, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("g,as_,cb,w", [(4096, 12, 32, 9), (300, 8, 32, 9), (65, 4, 32, 15), (7, 16, 64, 1), (129, 12, 64, 13), (9000, 12, 32, 6)])
def test_as_attention_proj(g, as_, cb, w):
    """pasnl_as_attention_proj (projections built inside the attention kernel) == the fp64 attention on the projected
    rows, and == the GEMM + pasnl_as_attention_qkv path it replaces for narrow inputs."""
    from pointasnl_amd import _hip

    rng = np.random.default_rng(g + w)
    x = rng.standard_normal((g, as_, w)).astype(np.float32)
    wkvq = (rng.standard_normal((w, 3 * cb)) * 0.5).astype(np.float32)
    bkvq = (rng.standard_normal(3 * cb) * 0.1).astype(np.float32)
    kvq64 = x.astype(np.float64) @ wkvq.astype(np.float64) + bkvq
    want = cells.nl_attention_core(kvq64[..., 2 * cb:], kvq64[..., :2 * cb], cb)
    xd, wd, bd = dev(x), dev(wkvq), dev(bkvq)
    got = torch.empty((g, as_, cb), device="cuda")
    _hip.launch("pasnl_as_attention_proj", "as_attention", g, as_, cb, w, _hip.ptr(xd), _hip.ptr(wd), _hip.ptr(bd), _hip.ptr(got))
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    kvq = torch.addmm(bd, xd.reshape(-1, w), wd).contiguous()
    two = torch.empty_like(got)
    _hip.launch("pasnl_as_attention_qkv", "as_attention", g, as_, cb, _hip.ptr(kvq), _hip.ptr(two))
    np.testing.assert_allclose(got.cpu().numpy(), two.cpu().numpy(), rtol=1e-5, atol=1e-5)


def _np_ce(logits, labels):
    z = logits.astype(np.float64)
    z = z - z.max(axis=-1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=-1))
    return lse - np.take_along_axis(z, labels[..., None], axis=-1)[..., 0]


def _np_l2(a):
    return float((a.astype(np.float64) ** 2).sum() / 2)


def test_get_loss_cls():
    """models/pointasnl_cls.py:55-70: cross entropy + uniform_weight * repulsion loss + weights_decay * sum l2(weights)."""
    from pointasnl_amd.models import pointasnl_cls
    from pointasnl_amd.utils import tf_util

    st = tf_util.set_store(tf_util.VariableStore(seed=77))
    pc = clouds(3, 4, 1024)
    logits, end = pointasnl_cls.get_model(dev(pc), is_training=False)
    labels = np.array([3, 17, 0, 39])
    p = st.export_numpy()
    reg = 1e-4 * sum(_np_l2(v["w"]) for v in p.values() if "w" in v)
    ce = _np_ce(logits.cpu().numpy(), labels).mean()
    got0 = float(pointasnl_cls.get_loss(logits, torch.tensor(labels, device="cuda"), end))
    assert abs(got0 - (ce + reg)) < 1e-5 * max(1.0, abs(ce + reg))  # uniform_weight = 0: classify + 0 * classify + reg
    got1 = float(pointasnl_cls.get_loss(logits, torch.tensor(labels, device="cuda"), end, uniform_weight=0.5))
    uni = cells.repulsion_loss(end["l1_xyz"].cpu().numpy(), nsample=20, radius=0.07)
    want1 = ce + 0.5 * uni + reg
    assert abs(got1 - want1) < 1e-5 * max(1.0, abs(want1))


def test_get_loss_sem_seg():
    """models/pointasnl_sem_seg.py:53-68: weighted cross entropy (sum / non-zero weights) + the 'losses' collection
    (wd * l2 of every layer built with a weight_decay) + uniform_weight * repulsion + weights_decay * sum l2(weights)."""
    from pointasnl_amd.models import pointasnl_sem_seg
    from pointasnl_amd.utils import tf_util

    st = tf_util.set_store(tf_util.VariableStore(seed=78))
    rng = np.random.default_rng(5)
    pc = np.concatenate([clouds(4, 1, 4096), rng.random((1, 4096, 3)).astype(np.float32)], axis=-1)
    st = tf_util.set_store(tf_util.VariableStore(seed=78))
    logits, end = pointasnl_sem_seg.get_model(dev(pc), False, 20, weight_decay=0.02, feature_channel=3)
    labels = rng.integers(0, 20, (1, 4096))
    smpw = rng.random((1, 4096)).astype(np.float32)
    smpw[0, :100] = 0.0
    got = float(pointasnl_sem_seg.get_loss(logits, torch.tensor(labels, device="cuda"), end, smpw=torch.tensor(smpw, device="cuda")))
    p = st.export_numpy()
    ce = _np_ce(logits.cpu().numpy(), labels)
    classify = float((ce * smpw).sum() / np.count_nonzero(smpw))
    reg = 1e-4 * sum(_np_l2(v["w"]) for v in p.values() if "w" in v)
    # every pointasnl_util layer registers its decay, fc1 / fc2 too (pointasnl_sem_seg.py:43-47); none of them is missing
    assert set(st.decays) == {k + "/weights" for k, v in p.items() if "w" in v}
    coll = 0.02 * sum(_np_l2(v["w"]) for v in p.values() if "w" in v)
    uni = cells.repulsion_loss(end["l1_xyz"].cpu().numpy(), nsample=20, radius=0.07)
    # tf.losses.sparse_softmax_cross_entropy adds its result to the 'losses' collection too: classify counts twice
    # (pinned against the reference's Python in tests/test_gpu_reference_fixtures.py::test_get_loss_matches_reference_python)
    want = classify + (coll + classify) + 0.01 * uni + reg
    assert abs(got - want) < 1e-5 * max(1.0, abs(want))


@pytest.mark.parametrize("n,c,p", [(200, 80, 50), (96, 512, 24), (64, 20, 16)])
def test_point_nonlocal_cell_any_bottleneck_width(n, c, p):
    """Bottleneck widths the fused kernel does not cover (max(32, C//2) = 40, 256) take the op-by-op chain
    (pointasnl_util.py:196-212 on the vendor BLAS) instead of raising; 32 stays on the kernel."""
    from pointasnl_amd.utils import pointasnl_util as U

    st = _store(n + c)
    rng = np.random.default_rng(c)
    feat = rng.standard_normal((2, n, c)).astype(np.float32)
    newf = rng.standard_normal((2, p, 3 + c)).astype(np.float32)
    mlp = [max(32, c // 2), 64]
    got = U.PointNonLocalCell(dev(feat), dev(newf).unsqueeze(1), mlp, False, None, None, "layerX", True)
    want = cells.point_nonlocal_cell(feat.astype(np.float64), newf.astype(np.float64), mlp, st.export_numpy(), "layerX")
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("c,as_", [(100, 12), (122, 8), (200, 12), (256, 6), (280, 4), (29, 8), (177, 12), (237, 5)])
def test_adaptive_sampling_fused_wide_bottlenecks(c, as_):
    """as_cell_wide instantiations beyond the models' own (cb = (3+c)//2 = 51, 62, 101, 129, 141: CBLK 4, 7, 9; 16, 90, 120:
    CBLK 1, 6, 8 -- the kernel is instantiated per exact block count since only the last block clamps and masks) and, for
    c = 280 (cb = 141 <= 144 still fused; 1 + channel = 284 columns), the gate into the few-kernel chain."""
    from pointasnl_amd.utils import pointasnl_util as U

    b, n, m, k = 2, 150, 40, 16
    st = _store(c + as_)
    rng = np.random.default_rng(c)
    xyz = clouds(13, b, n)
    feat = rng.standard_normal((b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, k)).astype(np.int32)
    with st.scope("layerA"):
        new_xyz, new_feat = U.adaptive_sampling_fused(dev(xyz), dev(feat), dev(idx), as_, "layerA", True)
    bi = np.arange(b)[:, None, None]
    gx = xyz[bi, idx].astype(np.float64)
    gf = np.concatenate([gx, feat[bi, idx].astype(np.float64)], axis=-1)
    want_xyz, want_feat = cells.adaptive_sampling(gx, gf, as_, st.export_numpy(), "layerA", outer="layerA")
    np.testing.assert_allclose(new_xyz.cpu().numpy(), want_xyz, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(new_feat.cpu().numpy(), want_feat, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("c,mlp", [(153, [128, 128, 256]), (154, [128, 128, 256]), (300, [64, 64, 128])])
def test_set_abstraction_rows_too_wide_for_the_fused_cell_fall_back(c, mlp):
    """Rows whose conv0 weights do not fit the LDS next to W1 (C = 153, 154 with c1 = 128: the Python gate of round 1 said
    'fused', the launcher said PASNL_EUNSUPPORTED and the layer raised) now take the next path down and stay correct."""
    from pointasnl_amd.utils import pointasnl_util as U

    b, n, npoint, ns = 1, 96, 24, 32
    st = _store(c)
    rng = np.random.default_rng(c)
    xyz = clouds(14, b, n)
    feat = rng.standard_normal((b, n, c)).astype(np.float32)
    with torch.no_grad():
        nx, npts = U.PointASNLSetAbstraction(dev(xyz), dev(feat), npoint, ns, mlp, False, None, None, "layerW", as_neighbor=0, NL=False)
    wx, wpts = cells.set_abstraction(xyz.astype(np.float64), feat.astype(np.float64), npoint, ns, mlp, st.export_numpy(), "layerW",
                                     as_neighbor=0, NL=False)
    np.testing.assert_array_equal(nx.cpu().numpy(), wx.astype(np.float32))
    scale = np.abs(wpts).max()
    assert np.abs(npts.cpu().numpy() - wpts).max() / scale < 1e-4


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_launch_refuses_tensors_of_another_device():
    from pointasnl_amd import _hip, tf_sampling

    x = torch.rand(1, 64, 3, device="cuda:1")
    with torch.cuda.device(0), pytest.raises(_hip.PasnlError):
        tf_sampling.farthest_point_sample(8, x)
    with torch.cuda.device(1):
        assert tf_sampling.farthest_point_sample(8, x).device.index == 1


@pytest.mark.parametrize("b,n,c,m,k,c1", [
    (3, 300, 3, 70, 32, 64), (2, 256, 128, 40, 64, 128), (17, 128, 32, 50, 32, 32), (2, 200, 64, 200, 96, 64),
    (2, 300, 128, 40, 32, 128), (1, 200, 256, 24, 32, 256), (2, 100, 256, 60, 32, 512),  # the wide kernels (one workgroup per group)
])
def test_sa_cell_takes_its_centres_from_neighbour_0(b, n, c, m, k, c1):
    """pasnl_sa_cell with new_xyz = NULL (AdaptiveSampling with as_neighbor == 0, pointasnl_util.py:161-163): the centre of a
    group is xyz[idx[b,j,0]], read from the kernel's own first tile -- bit-identical to handing it the gathered centres."""
    from pointasnl_amd.utils import pointasnl_util as U

    st = _store(b * 13 + c)
    rng = np.random.default_rng(n + c + 1)
    xyz = clouds(8, b, n)
    feat = rng.standard_normal((b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, k)).astype(np.int32)
    centres = xyz[np.arange(b)[:, None], idx[:, :, 0]]
    with st.scope("L"):
        want, want_skip = U.sa_cell(dev(xyz), dev(feat), dev(idx), dev(centres), [c1, c1, 2 * c1], False, None, None, True)
        got, got_skip, cen, nf = U.sa_cell(dev(xyz), dev(feat), dev(idx), None, [c1, c1, 2 * c1], False, None, None, True)
    assert torch.equal(got, want) and torch.equal(got_skip, want_skip)
    # ... and it hands back what pasnl_take_neighbor0 computes: the centres and [centre | feature row of neighbour 0]
    np.testing.assert_array_equal(cen.cpu().numpy(), centres)
    np.testing.assert_array_equal(nf.cpu().numpy(), np.concatenate([centres, feat[np.arange(b)[:, None], idx[:, :, 0]]], axis=-1))


def test_set_abstraction_without_adaptive_sampling_is_the_same_with_and_without_centre0():
    from pointasnl_amd.utils import pointasnl_util as U, tf_util

    rng = np.random.default_rng(4)
    xyz = dev(clouds(9, 3, 512))
    feat = dev(rng.standard_normal((3, 512, 64)).astype(np.float32))
    outs = []
    for flag in (True, False):
        tf_util.set_store(tf_util.VariableStore(seed=31, randomize_bn=True))
        U.CENTRE0 = flag
        try:
            seen = []
            outs.append(U.PointASNLSetAbstraction(xyz, feat, 128, 32, [64, 64, 128], False, None, None, 'layer', as_neighbor=0,
                                                  after_sampling=lambda x: seen.append(x)) + (seen[0],))
        finally:
            U.CENTRE0 = True
    torch.cuda.synchronize()
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)


def test_set_abstraction_can_write_the_next_modules_concat():
    """xyz_concat=True: the layer's tail kernel also writes [0 | new_xyz | new_points] rows and the group_all module that
    follows (pointasnl_cls layer3_x) consumes them instead of concatenating -- same numbers as the plain path."""
    from pointasnl_amd.utils import pointasnl_util as U, pointnet_util as PU, tf_util

    rng = np.random.default_rng(14)
    xyz = dev(clouds(19, 40, 256))
    feat = dev(rng.standard_normal((40, 256, 32)).astype(np.float32))
    res = []
    for flag in (True, False):
        tf_util.set_store(tf_util.VariableStore(seed=77, randomize_bn=True))
        new_xyz, pts = U.PointASNLSetAbstraction(xyz, feat, 64, 32, [32, 32, 64], False, None, None, 'layer1', as_neighbor=0,
                                                 xyz_concat=flag)
        assert hasattr(pts, "xyz_concat") == flag
        if flag:
            cat = pts.xyz_concat[1]
            assert pts.xyz_concat[0] is new_xyz and cat.shape == (40, 64, 4 + 64)
            want = torch.cat([torch.zeros_like(new_xyz[..., :1]), new_xyz, pts], dim=-1)
            assert torch.equal(cat, want)
        _, pooled, _ = PU.pointnet_sa_module(new_xyz, pts, None, None, None, [64, 128], None, True, False, None, 'layer3')
        res.append((pts, pooled))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-5)


def test_adaptive_sampling_padded_projection_is_the_same_function():
    """The [K | V | Q] projection padded to a multiple of 32 columns (a rounder GEMM) against the exact width."""
    from pointasnl_amd.utils import pointasnl_util as U

    b, n, m, k, c, as_ = 3, 300, 64, 32, 128, 12
    rng = np.random.default_rng(77)
    xyz, feat = dev(clouds(17, b, n)), dev(rng.standard_normal((b, n, c)).astype(np.float32))
    idx = dev(rng.integers(0, n, (b, m, k)).astype(np.int32))
    outs = []
    for flag in (True, False):
        st = _store(99)
        U.AS_PAD_PROJECTION = flag
        try:
            with st.scope("layerA"):
                outs.append(U.adaptive_sampling_fused(xyz, feat, idx, as_, "layerA", True))
        finally:
            U.AS_PAD_PROJECTION = True
    for a, b_ in zip(*outs):
        assert torch.allclose(a, b_, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("b,n,c", [(2, 300, 128), (3, 100, 32), (1, 257, 256), (2, 64, 512), (2, 200, 96)])
def test_decode_cell_tiled_is_the_plain_cell_reordered(b, n, c):
    """pasnl_decode_cell_tiled: the same (3+c)*32 values per point, bit for bit, at the positions include/pasnl.h documents
    (the order decode_after_conv's weight rows are gathered in)."""
    from pointasnl_amd.utils import pointasnl_util as U, tf_util

    rng = np.random.default_rng(c + n)
    xyz, feat = dev(clouds(3, b, n)), dev(rng.standard_normal((b, n, c)).astype(np.float32))
    idx = dev(rng.integers(0, n, (b, n, 16)).astype(np.int32))
    outs = []
    for flag in (False, True):
        tf_util.set_store(tf_util.VariableStore(seed=8, randomize_bn=True))
        U.DECODE_CELL_TILED = flag
        try:
            outs.append(U.decode_cell(xyz, feat, idx))
        finally:
            U.DECODE_CELL_TILED = True
    plain, tiled = outs
    assert not hasattr(plain, "row_order") and hasattr(tiled, "row_order")
    order = tiled.row_order[1]
    assert torch.equal(tiled.reshape(b, n, -1), plain.reshape(b, n, -1)[:, :, order])
    assert torch.equal(torch.sort(order).values, torch.arange((3 + c) * 32, device=order.device))


def test_decoding_layer_tiled_matches_plain():
    from pointasnl_amd.utils import pointasnl_util as U, tf_util

    rng = np.random.default_rng(5)
    xyz1, xyz2 = dev(clouds(4, 2, 512)), dev(clouds(5, 2, 128))
    p1 = dev(rng.standard_normal((2, 512, 64)).astype(np.float32))
    p2 = dev(rng.standard_normal((2, 128, 128)).astype(np.float32))
    outs = []
    for flag in (False, True):
        tf_util.set_store(tf_util.VariableStore(seed=9, randomize_bn=True))
        U.DECODE_CELL_TILED = flag
        try:
            outs.append(U.PointASNLDecodingLayer(xyz1, xyz2, p1, p2, 16, [128, 64], False, None, None, 'fa'))
        finally:
            U.DECODE_CELL_TILED = True
    assert torch.allclose(outs[0], outs[1], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("rows,w,cb,c,cat,res", [
    (32768, 9, 32, 128, True, False), (8192, 134, 64, 256, False, False), (100, 9, 32, 128, False, True), (77, 134, 64, 256, True, True),
    (33, 70, 0, 32, False, False), (64, 16, 16, 512, False, True), (1, 1, 1, 32, True, False), (2049, 257, 33, 96, False, False),
])
def test_sa_tail_packed_weights_give_the_same_bits(rows, w, cb, c, cat, res):
    """pasnl_sa_tail_packed (weights in the matrix instruction's operand order, read in 16-byte pieces) == pasnl_sa_tail / _cat /
    _res on the row-major matrices, bit for bit: contraction lengths that are no multiple of a 32-index chunk (zero padding
    inside the packed matrices), with and without back-projection, concat rows and residual."""
    from pointasnl_amd import _hip
    rng = np.random.default_rng(7 * rows + c)
    f = lambda *sh: dev(rng.standard_normal(sh).astype(np.float32))
    A, S, N, R, xyz = f(rows, c), f(rows, w), f(rows, max(cb, 1)), f(rows, c), f(rows, 3)
    ws, bs, wb, bb, wagg, bagg = f(w, c) / np.sqrt(w), f(c), f(max(cb, 1), c), f(c), f(c, c) / np.sqrt(c), f(c)
    null = _hip.ptr(None)

    def packed(m):
        pk = torch.empty(int(_hip.lib().pasnl_sa_tail_packed_weights_bytes(m.shape[0], m.shape[1])) // 4, device="cuda")
        _hip.launch("pasnl_sa_tail_pack_weights", "pack", m.shape[0], m.shape[1], _hip.ptr(m), _hip.ptr(pk))
        return pk
    head = [rows, w, cb, c, _hip.ptr(A), _hip.ptr(S), _hip.ptr(N) if cb else null]
    plain = torch.full((rows, c), float("nan"), device="cuda")
    plain_cat = torch.full((rows, c + 4), float("nan"), device="cuda")
    tailw = [_hip.ptr(ws), _hip.ptr(bs), _hip.ptr(wb) if cb else null, _hip.ptr(bb) if cb else null, _hip.ptr(wagg), _hip.ptr(bagg)]
    if res:
        _hip.launch("pasnl_sa_tail_res", "sa_tail", *head, *tailw, _hip.ptr(R), _hip.ptr(plain))
        if cat:  # (no plain entry point does both: the concat rows of the packed form are checked against its own output)
            plain_cat = None
    elif cat:
        _hip.launch("pasnl_sa_tail_cat", "sa_tail", *head, *tailw, _hip.ptr(plain), _hip.ptr(xyz), _hip.ptr(plain_cat))
    else:
        _hip.launch("pasnl_sa_tail", "sa_tail", *head, *tailw, _hip.ptr(plain))
    pws, pwb, pwagg = packed(ws), packed(wb) if cb else None, packed(wagg)
    got = torch.full((rows, c), float("nan"), device="cuda")
    got_cat = torch.full((rows, c + 4), float("nan"), device="cuda")
    _hip.launch("pasnl_sa_tail_packed", "sa_tail", *head, _hip.ptr(pws), _hip.ptr(bs), _hip.ptr(pwb), _hip.ptr(bb) if cb else null,
                _hip.ptr(pwagg), _hip.ptr(bagg), _hip.ptr(R) if res else null, _hip.ptr(xyz) if cat else null,
                _hip.ptr(got_cat) if cat else null, _hip.ptr(got))
    assert torch.equal(got, plain)
    if cat:
        assert torch.equal(got_cat[:, 4:], got) and torch.equal(got_cat[:, 1:4], xyz) and bool((got_cat[:, 0] == 0).all())
        if plain_cat is not None:
            assert torch.equal(got_cat, plain_cat)


@pytest.mark.parametrize("rows,w,cb,c", [(10240, 35, 0, 64), (2560, 67, 0, 128), (77, 134, 64, 256), (64, 16, 0, 512), (2049, 257, 33, 96)])
def test_sa_tail_res_is_tail_plus_residual(rows, w, cb, c):
    """pasnl_sa_tail_res (the `_res` model's residual sum, pointasnl_sem_seg_res.py:37,42,47,52, in the tail's epilogue):
    pasnl_sa_tail's rows + the residual, bit for bit."""
    from pointasnl_amd import _hip
    rng = np.random.default_rng(rows + c)
    f = lambda *sh: dev(rng.standard_normal(sh).astype(np.float32))
    A, S, N, R = f(rows, c), f(rows, w), f(rows, max(cb, 1)), f(rows, c)
    ws, bs, wb, bb, wagg, bagg = f(w, c) / np.sqrt(w), f(c), f(max(cb, 1), c), f(c), f(c, c) / np.sqrt(c), f(c)
    null = _hip.ptr(None)
    head = [rows, w, cb, c, _hip.ptr(A), _hip.ptr(S), _hip.ptr(N) if cb else null, _hip.ptr(ws), _hip.ptr(bs),
            _hip.ptr(wb) if cb else null, _hip.ptr(bb) if cb else null, _hip.ptr(wagg), _hip.ptr(bagg)]
    plain = torch.full((rows, c), float("nan"), device="cuda")
    summed = torch.full((rows, c), float("nan"), device="cuda")
    _hip.launch("pasnl_sa_tail", "sa_tail", *head, _hip.ptr(plain))
    _hip.launch("pasnl_sa_tail_res", "sa_tail", *head, _hip.ptr(R), _hip.ptr(summed))
    assert torch.equal(summed, plain + R)


@pytest.mark.parametrize("rows,w,cb,c,cat", [
    (32768, 9, 32, 128, True),    # cls layer1 (narrow skip product: 9 columns), with the [0 | xyz | out] rows
    (8192, 134, 64, 256, False),  # cls layer2
    (100, 9, 32, 128, False),     # ragged last tile
    (77, 134, 64, 256, True),
    (33, 70, 0, 32, False),       # no back-projection (NL=False layers): att = NULL
    (64, 16, 16, 512, False),     # 16 waves per workgroup
    (1, 1, 1, 32, True),
    (2049, 257, 33, 96, False),   # nothing a multiple of a chunk
])
def test_sa_tail_matches_fp64(rows, w, cb, c, cat):
    """pasnl_sa_tail / pasnl_sa_tail_cat through the C ABI: out = relu((A + relu(S Ws + bs) + relu(N Wb + bb)) Wagg + bagg)
    (pointasnl_util.py:138-151 of the reference: skip convolution, back-projection, both adds, aggregation) against the
    fp64 product; operand chunks, tiles and row tiles that do not divide (clamped loads, zero-padded tiles)."""
    from pointasnl_amd import _hip
    rng = np.random.default_rng(rows + w + c)
    f = lambda *sh: rng.standard_normal(sh).astype(np.float32)
    A, S, N = f(rows, c), f(rows, w), f(rows, max(cb, 1))
    ws, bs, wb, bb = f(w, c) / np.float32(np.sqrt(w)), f(c), f(max(cb, 1), c) / np.float32(np.sqrt(max(cb, 1))), f(c)
    wagg, bagg, xyz = f(c, c) / np.float32(np.sqrt(c)), f(c), f(rows, 3)
    want = A.astype(np.float64) + np.maximum(S.astype(np.float64) @ ws + bs, 0)
    if cb:
        want = want + np.maximum(N.astype(np.float64) @ wb + bb, 0)
    want = np.maximum(want @ wagg.astype(np.float64) + bagg, 0)
    d = {k: dev(v) for k, v in dict(A=A, S=S, N=N, ws=ws, bs=bs, wb=wb, bb=bb, wagg=wagg, bagg=bagg, xyz=xyz).items()}
    out = torch.full((rows, c), float("nan"), device="cuda")
    null = _hip.ptr(None)
    args = [rows, w, cb, c, _hip.ptr(d["A"]), _hip.ptr(d["S"]), _hip.ptr(d["N"]) if cb else null, _hip.ptr(d["ws"]),
            _hip.ptr(d["bs"]), _hip.ptr(d["wb"]) if cb else null, _hip.ptr(d["bb"]) if cb else null, _hip.ptr(d["wagg"]),
            _hip.ptr(d["bagg"]), _hip.ptr(out)]
    if cat:
        out_cat = torch.full((rows, c + 4), float("nan"), device="cuda")
        _hip.launch("pasnl_sa_tail_cat", "sa_tail", *args, _hip.ptr(d["xyz"]), _hip.ptr(out_cat))
    else:
        _hip.launch("pasnl_sa_tail", "sa_tail", *args)
    got = out.cpu().numpy()
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    if cat:
        oc = out_cat.cpu().numpy()
        np.testing.assert_array_equal(oc[:, 0], 0)
        np.testing.assert_array_equal(oc[:, 1:4], xyz)
        np.testing.assert_array_equal(oc[:, 4:], got)
