"""ctypes binding of libpasnl_hip.so (include/pasnl.h) -- the only native code the product path runs.

There is NO fallback: if the library is missing, or no HIP device is visible, every op raises.  Tensors are
torch CUDA(ROCm) tensors used as plain device buffers; launches go to torch's current stream, so the ops
compose with torch work and can be captured into a HIP graph (no allocation or sync happens inside the
library).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpasnl_hip.so")

# every symbol include/pasnl.h declares; tests/test_boundary.py checks the header and this list agree
SYMBOLS = [
    "pasnl_version", "pasnl_strerror", "pasnl_device_count",
    "pasnl_farthest_point_sample", "pasnl_farthest_point_sample_gather", "pasnl_gather_point", "pasnl_gather_point_grad", "pasnl_prob_sample",
    "pasnl_query_ball_point", "pasnl_sa_group", "pasnl_group_point", "pasnl_group_point_grad", "pasnl_select_top_k", "pasnl_knn_batch", "pasnl_knn_workspace_bytes", "pasnl_knn_batch_ws", "pasnl_knn_batch_ws_bg", "pasnl_knn_tree_workspace_bytes", "pasnl_knn_batch_tree", "pasnl_knn_batch_ref_workspace_bytes", "pasnl_knn_batch_ref", "pasnl_knn_distance_pick",
    "pasnl_three_nn", "pasnl_three_interpolate", "pasnl_three_interpolate_grad", "pasnl_three_weights", "pasnl_fp_interpolate_cat",
    "pasnl_nl_attention", "pasnl_nl_attention_workspace_bytes", "pasnl_nl_attention_ws", "pasnl_as_attention", "pasnl_as_reweight", "pasnl_sa_local_cell", "pasnl_sa_cell", "pasnl_sa_cell_centre0", "pasnl_sa_cell_packed", "pasnl_sa_tail", "pasnl_sa_tail_cat", "pasnl_sa_tail_res", "pasnl_sa_tail_packed_weights_bytes", "pasnl_sa_tail_pack_weights", "pasnl_sa_tail_packed", "pasnl_decode_cell", "pasnl_decode_cell_tiled", "pasnl_decode_cell_tiled_v4", "pasnl_max_pool_rows", "pasnl_max_pool_rows_strided", "pasnl_mlp3_max_pool_workspace_bytes", "pasnl_mlp3_max_pool", "pasnl_mlp3_packed_weights_bytes", "pasnl_mlp3_pack_weights", "pasnl_dense_rows_workspace_bytes", "pasnl_dense_rows", "pasnl_dense_splitk_workspace_bytes", "pasnl_dense_splitk", "pasnl_bf16x3_weights_bytes", "pasnl_bf16x3_split_weights", "pasnl_dense_bf16x3", "pasnl_narrow_project2", "pasnl_take_neighbor0", "pasnl_as_gather", "pasnl_as_attention_qkv", "pasnl_as_attention_proj", "pasnl_as_cell_narrow", "pasnl_as_cell_wide", "pasnl_as_cell_wide_ld", "pasnl_as_reweight_x", "pasnl_grid_subsample_workspace_bytes", "pasnl_grid_subsample", "pasnl_knn_crop_workspace_bytes", "pasnl_knn_crop",
    "pasnl_grad_workspace_bytes", "pasnl_gather_point_grad_det", "pasnl_group_point_grad_det", "pasnl_three_interpolate_grad_det",
]


class PasnlError(RuntimeError):
    pass


class PasnlUnsupported(PasnlError):
    """PASNL_EUNSUPPORTED: a valid request outside what the fused kernel covers (include/pasnl.h documents the limits).
    The cells catch it and take their op-by-op path (still HIP kernels + vendor GEMMs, never a CPU path)."""


_lib = None


def lib():
    """Load the C-ABI library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PasnlError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C pointasnl_amd/csrc`).  pointasnl_amd has no CPU or eager fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.pasnl_strerror.restype = ctypes.c_char_p
        _lib.pasnl_grad_workspace_bytes.restype = ctypes.c_size_t
        _lib.pasnl_knn_workspace_bytes.restype = ctypes.c_size_t
        _lib.pasnl_knn_tree_workspace_bytes.restype = ctypes.c_size_t
        _lib.pasnl_knn_batch_ref_workspace_bytes.restype = ctypes.c_size_t
        _lib.pasnl_dense_rows_workspace_bytes.restype = ctypes.c_size_t
        _lib.pasnl_bf16x3_weights_bytes.restype = ctypes.c_size_t
        _lib.pasnl_mlp3_max_pool_workspace_bytes.restype = ctypes.c_size_t
        _lib.pasnl_mlp3_packed_weights_bytes.restype = ctypes.c_size_t
        _lib.pasnl_sa_tail_packed_weights_bytes.restype = ctypes.c_size_t
        _lib.pasnl_dense_splitk_workspace_bytes.restype = ctypes.c_size_t
        _lib.pasnl_grid_subsample_workspace_bytes.restype = ctypes.c_size_t
        _lib.pasnl_knn_crop_workspace_bytes.restype = ctypes.c_size_t
        _lib.pasnl_nl_attention_workspace_bytes.restype = ctypes.c_size_t
        for s in SYMBOLS:
            getattr(_lib, s)  # AttributeError here == header / library mismatch
    return _lib


def require_device():
    if not torch.cuda.is_available():
        raise PasnlError("no HIP device visible: pointasnl_amd runs on MI355X (gfx950) only and has no CPU fallback")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor for a launch on torch's CURRENT stream: the tensor has to live on the current device
    (launching on cuda:0's stream with cuda:1's memory would fault or silently use the wrong GPU)."""
    if t is None:
        return ctypes.c_void_p(0)
    if t.device.index != torch.cuda.current_device():
        raise PasnlError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: wrap the call in "
                         "`with torch.cuda.device(tensor.device):` (launches go to the current device's stream)")
    return ctypes.c_void_p(t.data_ptr())


# Optional per-launch timing for bench.py: when PROFILE is a list, every C-ABI launch is bracketed by two
# HIP events recorded on the stream the kernel is enqueued on (torch's current stream) and
# (symbol, integer-arguments, start_event, end_event) is appended.  Off (None) in normal operation.
PROFILE = None


# Attribution of hardware-counter rows to launches (bench.py --traffic-pass under rocprofv3 --pmc): when MARK is a list, a
# marker kernel (torch.cuda._sleep -> "spin_kernel") is enqueued in front of every C-ABI launch and (symbol, integer-arguments)
# is appended; the counter rows between two markers then belong to one launch whatever number of kernels it starts.
MARK = None
MARK_SKIP = False  # marked, but not to be counted (first launches that also create workspaces)


TRACE = bool(os.environ.get("PASNL_TRACE"))  # debugging: print + synchronise around every launch


AFTER_LAUNCH = []  # callables run behind the next C-ABI launch (pointasnl_util.Forked starts its lazy forks there)


def launch(symbol, what, *args):
    """Call one pasnl_* entry point with the current stream appended; raise on a non-zero status."""
    if AFTER_LAUNCH:
        try:
            return _launch(symbol, what, *args)
        finally:
            for cb in list(AFTER_LAUNCH):
                cb()
    return _launch(symbol, what, *args)


def _launch(symbol, what, *args):
    fn = getattr(lib(), symbol)
    if TRACE:
        import sys
        import time
        ints = tuple(a for a in args if isinstance(a, (int, float)))
        print(f"[pasnl] {symbol}{ints} ...", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        check(fn(*args, stream_ptr()), what)
        torch.cuda.synchronize()
        print(f"[pasnl]   done {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr, flush=True)
        return
    if MARK is not None:
        torch.cuda._sleep(1)
        code = fn(*args, stream_ptr())
        # a launch that answered PASNL_EUNSUPPORTED started no kernel: its marker interval is empty and must not be read as a
        # zero-traffic measurement (profiles/pmc_to_traffic.py skips "_unsupported")
        MARK.append(["_unmeasured" if MARK_SKIP else (symbol if code == 0 else "_unsupported"),
                     [a if isinstance(a, int) else a.value for a in args if isinstance(a, (int, ctypes.c_long))]])
        check(code, what)
        return
    if PROFILE is None:
        check(fn(*args, stream_ptr()), what)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    code = fn(*args, stream_ptr())
    e1.record()
    check(code, what)
    PROFILE.append((symbol, tuple(a if isinstance(a, int) else a.value for a in args if isinstance(a, (int, ctypes.c_long))), e0, e1))


def check(code, what):
    if code != 0:
        msg = lib().pasnl_strerror(code).decode()
        if code == -1:
            raise ValueError(f"{what}: {msg}")
        if code == -5:
            raise PasnlUnsupported(f"{what}: {msg} (code {code})")
        raise PasnlError(f"{what}: {msg} (code {code})")


def as_dev(x, dtype):
    """Accept a torch tensor (must already live on the GPU) or a numpy array (host buffer: copied over PCIe)."""
    require_device()
    if isinstance(x, torch.Tensor):
        if not x.is_cuda:
            raise PasnlError("CPU torch tensor passed to a pointasnl_amd op: move it to the GPU (no CPU fallback)")
        if x.dtype != dtype:
            raise ValueError(f"expected dtype {dtype}, got {x.dtype}")
        return x.contiguous()
    import numpy as np

    return torch.from_numpy(np.ascontiguousarray(x)).to(device="cuda", dtype=dtype)


DETERMINISTIC_GRADS = True  # backward of gather_point / group_point / three_interpolate: ordered segmented sums (no fp atomics)


def grad_workspace(b, targets, contributions, device):
    """(device buffer, byte count as c_size_t) for the pasnl_*_grad_det entry points."""
    nbytes = int(lib().pasnl_grad_workspace_bytes(int(b), int(targets), ctypes.c_long(int(contributions))))
    ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=device)
    return ws, ctypes.c_size_t(nbytes)
