"""CPU: the C-ABI boundary.  The library loads, exports exactly what include/pasnl.h declares, the product has
no CPU fallback and never touches oracle/, and the reference-named modules import under the reference's idiom."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pasnl.h")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pasnl_[a-z0-9_]+)\s*\(", src)) - {"pasnl_stream_t"})


def test_library_exports_every_declared_symbol():
    from pointasnl_amd import _hip

    lib = _hip.lib()
    syms = header_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/pasnl.h but not exported"
    assert sorted(_hip.SYMBOLS) == syms
    nm = subprocess.run(["nm", "-D", "--defined-only", _hip.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (pasnl_\w+)", nm))
    assert exported == set(syms)


def test_version_and_strerror_need_no_gpu():
    from pointasnl_amd import _hip

    lib = _hip.lib()
    assert lib.pasnl_version() == 100
    assert b"invalid argument" in lib.pasnl_strerror(-1)
    assert lib.pasnl_strerror(0) == b"ok"


def test_argument_validation_happens_before_any_launch():
    # bad attributes are rejected on the host side of the ABI without touching a device
    from pointasnl_amd import _hip

    lib = _hip.lib()
    null = ctypes.c_void_p(0)
    assert lib.pasnl_farthest_point_sample(1, 16, 0, null, null, null) == -1       # npoint <= 0
    assert lib.pasnl_query_ball_point(1, 16, 4, ctypes.c_float(0.0), 4, null, null, null, null, null) == -1
    assert lib.pasnl_query_ball_point(1, 16, 4, ctypes.c_float(0.1), 0, null, null, null, null, null) == -1
    assert lib.pasnl_select_top_k(1, 8, 2, 0, null, null, null, null) == -1
    assert lib.pasnl_knn_batch(1, 8, 2, 9, null, null, null, 0, null, null) == -1  # k > n
    assert lib.pasnl_knn_batch(1, 8, 2, 4, null, null, null, 0, null, null) == -2  # null pointers
    assert lib.pasnl_nl_attention(1, 4, 4, 48, null, null, null, 0, null) == -5     # cb not in {32,64,128}
    assert lib.pasnl_farthest_point_sample(0, 16, 4, null, null, null) == 0         # empty batch is a no-op


def test_no_cpu_fallback():
    import torch

    import pointasnl_amd
    from pointasnl_amd import _hip

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_hip.PasnlError, match="no CPU fallback"):
        pointasnl_amd.tf_sampling.farthest_point_sample(4, torch.zeros(1, 8, 3))
    with pytest.raises(_hip.PasnlError):
        pointasnl_amd.nearest_neighbors.knn_batch(torch.zeros(1, 8, 3).numpy(), torch.zeros(1, 2, 3).numpy(), 2)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pointasnl_amd")
    bad = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                # real uses only (comments may cite oracle/ as the place a kernel's contract is restated)
                if re.search(r"^\s*(from|import)\s+oracle\b|liboracle|libref_|#include\s+\"[^\"]*oracle|"
                             r"(CDLL|dlopen|import_module|__import__)\([^)]*oracle", txt, flags=re.M):
                    bad.append(os.path.join(d, f))
    assert not bad, f"product code references the oracle: {bad}"
    nm = subprocess.run(["ldd", os.path.join(pkg, "csrc", "libpasnl_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in nm


def test_reference_import_idiom():
    import sys

    import pointasnl_amd

    pointasnl_amd.install_paths()
    import tf_grouping
    import tf_interpolate
    import tf_sampling

    for name in ("farthest_point_sample", "gather_point", "prob_sample"):
        assert hasattr(tf_sampling, name)
    for name in ("query_ball_point", "group_point", "select_top_k", "knn_point"):
        assert hasattr(tf_grouping, name)
    for name in ("three_nn", "three_interpolate"):
        assert hasattr(tf_interpolate, name)
    from pointasnl_amd.utils.nearest_neighbors.lib.python import nearest_neighbors

    assert hasattr(nearest_neighbors, "knn_batch") and hasattr(nearest_neighbors, "knn")
    assert sys.modules["tf_sampling"] is pointasnl_amd.tf_sampling


def test_variable_store_is_seeded_and_folds_bn():
    import numpy as np

    from pointasnl_amd.utils import tf_util

    a = tf_util.VariableStore(seed=5, device="cpu", randomize_bn=True)
    b = tf_util.VariableStore(seed=5, device="cpu", randomize_bn=True)
    with a.scope("l"), a.scope("c"):
        wa, ba = a.layer(7, 9, bn=True)
    with b.scope("l"), b.scope("c"):
        wb, bb = b.layer(7, 9, bn=True)
    assert (wa == wb).all() and (ba == bb).all()
    p = a.export_numpy()["l/c"]
    x = np.random.default_rng(0).random((4, 7), dtype=np.float32)
    want = ((x @ p["w"] + p["b"]) - p["mean"]) / np.sqrt(p["var"] + 1e-3) * p["gamma"] + p["beta"]
    np.testing.assert_allclose(x @ wa.numpy() + ba.numpy(), want, rtol=1e-5, atol=1e-6)


def test_product_library_reads_no_environment_variable():
    """include/pasnl.h promises "no global state": the A/B and probe switches of the kernels (PASNL_NL_SPLIT, PASNL_BALL_PROBE
    ...) exist only in the tuning build (-DPASNL_TUNING -> libpasnl_hip_tuning.so, which the package never loads)."""
    import re

    csrc = os.path.join(ROOT, "pointasnl_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".hpp")):
            continue
        text = open(os.path.join(csrc, name)).read()
        text = re.sub(r"#ifdef PASNL_TUNING.*?#else", "", text, flags=re.S)
        assert "getenv(" not in text, name
    assert "libpasnl_hip_tuning" not in open(os.path.join(ROOT, "pointasnl_amd", "_hip.py")).read()
