"""Timings of csrc/dense.hip against the vendor GEMMs they replace (cls head, first non-local cell's projections),
back-to-back launches on one stream, HIP events.   python tools/dense_probe.py [iters]"""
import ctypes
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import pointasnl_amd  # noqa: E402
from pointasnl_amd import _hip  # noqa: E402
from pointasnl_amd.utils import tf_util  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
import os  # noqa: E402
if os.environ.get("PASNL_PROBE_LIB") == "tuning":  # `make -C pointasnl_amd/csrc tuning`: PASNL_DENSE_KCHUNK is honoured
    _hip.LIB_PATH = _hip.LIB_PATH.replace("libpasnl_hip.so", "libpasnl_hip_tuning.so")


def timed(fn, per_graph=20):
    """average time of one call inside a replayed HIP graph of `per_graph` back-to-back calls (launch gaps included, host
    overhead excluded -- the way the forward runs them)"""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(per_graph):
                fn()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(1, iters // per_graph)
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * per_graph)


g = torch.Generator(device="cuda").manual_seed(0)
for rows, k, n, relu in [(64, 1536, 512, True), (64, 512, 256, True), (64, 256, 40, False), (16, 1536, 512, True)]:
    x = torch.randn(rows, k, device="cuda", generator=g)
    w = torch.randn(k, n, device="cuda", generator=g) / k ** 0.5
    b = torch.randn(n, device="cuda", generator=g)
    t_hip = timed(lambda: tf_util._dense_rows(x, w, b, relu))
    t_blas = timed((lambda: torch._addmm_activation(b, x, w)) if relu else (lambda: torch.addmm(b, x, w)))
    print(f"dense_rows {rows}x{k}x{n}: hip {t_hip:7.2f} us   vendor {t_blas:7.2f} us")

for rows0, k0, n0, rows1, k1, n1 in [(64 * 1024, 3, 64, 64 * 512, 6, 32), (16 * 8192, 3, 64, 16 * 1024, 6, 32)]:
    x0 = torch.randn(rows0, k0, device="cuda", generator=g); w0 = torch.randn(k0, n0, device="cuda", generator=g)
    x1 = torch.randn(rows1, k1, device="cuda", generator=g); w1 = torch.randn(k1, n1, device="cuda", generator=g)
    b0 = torch.randn(n0, device="cuda", generator=g); b1 = torch.randn(n1, device="cuda", generator=g)
    o0 = torch.empty(rows0, n0, device="cuda"); o1 = torch.empty(rows1, n1, device="cuda")

    def hip():
        _hip.launch("pasnl_narrow_project2", "narrow_project", ctypes.c_long(rows0), k0, n0, _hip.ptr(x0), _hip.ptr(w0),
                    _hip.ptr(b0), _hip.ptr(o0), ctypes.c_long(rows1), k1, n1, _hip.ptr(x1), _hip.ptr(w1), _hip.ptr(b1), _hip.ptr(o1))

    def blas():
        torch.addmm(b0, x0, w0, out=o0)
        torch.addmm(b1, x1, w1, out=o1)

    print(f"narrow_project2 {rows0}x{k0}x{n0} + {rows1}x{k1}x{n1}: hip {timed(hip):7.2f} us   vendor {timed(blas):7.2f} us "
          f"({(o0.numel() + o1.numel()) * 4 / 1e6:.1f} MB out)")

xyz = torch.rand(64, 1024, 3, device="cuda", generator=g)
t0 = timed(lambda: pointasnl_amd.tf_sampling.farthest_point_sample(512, xyz))
t1 = timed(lambda: pointasnl_amd.tf_sampling.farthest_point_sample_gather(512, xyz))
print(f"fps 64x1024->512: {t0:.1f} us, with the gather {t1:.1f} us")
