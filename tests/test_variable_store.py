"""CPU: VariableStore.load takes values keyed by the reference's TF variable names (a trained checkpoint exported to npz),
flattens conv kernels, validates shapes, refuses missing names in strict mode, and every update drops the BN-folded caches
(assigning trained values after a forward was silently ignored)."""
import numpy as np
import pytest
import torch

from oracle import weights
from pointasnl_amd.utils import tf_util


def _names():
    return [("L/conv0/weights", [1, 1, 9, 16]), ("L/conv0/biases", [16]), ("L/conv0/bn/beta", [16]), ("L/conv0/bn/gamma", [16]),
            ("L/conv0/bn/moving_mean", [16]), ("L/conv0/bn/moving_variance", [16]),
            ("L/after_conv/weights", [1, 4, 8, 5]), ("L/after_conv/biases", [5])]


def test_load_by_tf_names_flattens_and_folds():
    vals = weights.make_all(11, _names())
    vals["L/conv0/weights:0"] = vals.pop("L/conv0/weights")          # ':0' suffix as tf.global_variables() prints it
    vals["L/conv0/weights/Adam_1"] = np.zeros((1, 1, 9, 16), np.float32)  # optimizer slot: ignored
    vals["global_step"] = np.array(7)
    st = tf_util.VariableStore(seed=0, device="cpu").load(vals, strict=True)
    with st.scope("L"), st.scope("conv0"):
        w, b = st.layer(9, 16, bn=True)
    w0 = vals["L/conv0/weights:0"].reshape(9, 16).astype(np.float64)
    s = vals["L/conv0/bn/gamma"] / np.sqrt(vals["L/conv0/bn/moving_variance"].astype(np.float64) + 1e-3)
    np.testing.assert_allclose(w.numpy(), w0 * s, rtol=1e-6)
    np.testing.assert_allclose(b.numpy(), (vals["L/conv0/biases"] - vals["L/conv0/bn/moving_mean"]) * s + vals["L/conv0/bn/beta"],
                               rtol=1e-5, atol=1e-7)
    with st.scope("L"), st.scope("after_conv"):  # [1,4,8,5] -> (32, 5), row-major over (W, C)
        w, _ = st.layer(32, 5, bn=False)
    np.testing.assert_array_equal(w.numpy(), vals["L/after_conv/weights"].reshape(32, 5))


def test_strict_load_refuses_unknown_variables_and_wrong_shapes():
    st = tf_util.VariableStore(seed=0, device="cpu").load(weights.make_all(11, _names()), strict=True)
    with pytest.raises(KeyError):
        with st.scope("L"), st.scope("conv1"):
            st.layer(16, 16, bn=True)
    with pytest.raises(ValueError):
        with st.scope("L"), st.scope("conv0"):
            st.layer(10, 16, bn=True)


def test_updates_invalidate_the_folded_cache():
    st = tf_util.VariableStore(seed=3, device="cpu")
    with st.scope("fc"):
        w_before, _ = st.layer(4, 2, bn=True)
        w_before = w_before.clone()
    st.assign("fc/bn/gamma", torch.full((2,), 3.0))
    with st.scope("fc"):
        w_after, _ = st.layer(4, 2, bn=True)
    assert not torch.equal(w_before, w_after)
    st.load({"fc/weights": np.ones((4, 2), np.float32)}, strict=False)
    with st.scope("fc"):
        w_loaded, _ = st.layer(4, 2, bn=True)
    np.testing.assert_allclose(w_loaded.numpy(), 3.0 / np.sqrt(st.vars["fc/bn/moving_variance"].numpy() + 1e-3) * np.ones((4, 1)), rtol=1e-6)


def test_dense_layer_with_padded_or_reordered_inputs_is_the_same_function():
    """tf_util.conv2d(input_pad=..., row_order=...): producers may hand a dense layer rows with alignment padding in front
    (PointASNLSetAbstraction(xyz_concat=True)) or with their values in another fixed order (the tiled decoder cell); the layer
    pads / gathers its weight rows once and caches them with the other folded weights -- and load() drops that cache."""
    st = tf_util.VariableStore(seed=3, device="cpu", randomize_bn=True)
    tf_util.set_store(st)
    x = torch.randn(4, 7, 1, 12)
    ref = tf_util.conv2d(x, 5, [1, 1], scope="L", bn=True, is_training=False)
    padded = torch.cat([torch.full((4, 7, 1, 2), 123.0), x], dim=-1)  # whatever sits in the padding meets zero weight rows
    got = tf_util.conv2d(padded, 5, [1, 1], scope="L", bn=True, is_training=False, input_pad=2)
    torch.testing.assert_close(got, ref)
    order = torch.randperm(12)
    wide = torch.randn(2, 3, 4, 3)  # (B, H, W, C): the [1, W] VALID kernel contracts the flattened (W, C) window
    ref2 = tf_util.conv2d(wide, 6, [1, 4], scope="M", padding="VALID", bn=False, is_training=False)
    shuffled = wide.reshape(2, 3, 12)[:, :, order].reshape(2, 3, 4, 3)
    got2 = tf_util.conv2d(shuffled, 6, [1, 4], scope="M", padding="VALID", bn=False, is_training=False, row_order=("perm", order))
    torch.testing.assert_close(got2, ref2)
    assert any("@pad2" in k for k in st._folded) and any("@perm" in k for k in st._folded)
    st.assign("L/weights", torch.zeros(12, 5))
    assert not st._folded  # every derived weight is rebuilt after an update


def test_decode_tiled_order_is_a_permutation_with_the_documented_layout():
    from pointasnl_amd.utils import pointasnl_util as U

    for c, v4 in [(128, True), (128, False), (32, False), (512, True), (96, False)]:
        order = U.decode_tiled_order(c, v4, torch.device("cpu"))
        assert order.shape == ((3 + c) * 32,)
        assert torch.equal(torch.sort(order).values, torch.arange((3 + c) * 32))
        assert torch.equal(order[:96], torch.arange(96))
        V = 4 if v4 else 1
        for T, g, h, m, i in [(0, 0, 0, 0, 0), (c // 32 - 1, 3, 1, 31, 3), (min(2, c // 32 - 1), 1, 0, 7, 2)]:
            q = 96 + T * 1024 + (2 * g + h) * 128 + 4 * m + i
            ch, j = 3 + 32 * V * (T // V) + V * m + T % V, 8 * g + 4 * h + i
            assert int(order[q]) == ch * 32 + j
