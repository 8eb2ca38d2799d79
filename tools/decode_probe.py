"""pasnl_decode_cell alone at the ScanNet decoder shapes, replayed from a HIP graph.
    [PASNL_PROBE_LIB=<suffix>] python tools/decode_probe.py"""
import os
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from pointasnl_amd import _hip  # noqa: E402

if os.environ.get("PASNL_PROBE_LIB"):
    _hip.LIB_PATH = _hip.LIB_PATH.replace("libpasnl_hip.so", f"libpasnl_hip_{os.environ['PASNL_PROBE_LIB']}.so")
g = torch.Generator(device="cuda").manual_seed(0)
ref = {}
for b, n, c, k in [(16, 8192, 128, 16), (16, 1024, 256, 16), (16, 256, 512, 16), (16, 64, 512, 16), (8, 10240, 32, 16)]:
    xyz = torch.rand(b, n, 3, device="cuda", generator=g)
    feat = torch.randn(b, n, c, device="cuda", generator=g)
    idx = torch.randint(0, n, (b, n, k), device="cuda", dtype=torch.int32, generator=g)
    ww, bw = torch.randn(3, 32, device="cuda", generator=g), torch.randn(32, device="cuda", generator=g)
    out = torch.empty(b, n, 3 + c, 32, device="cuda")
    sym = "pasnl_decode_cell_tiled" if os.environ.get("TILED") else "pasnl_decode_cell"
    run = lambda: _hip.launch(sym, "decode_cell", b, n, c, k, _hip.ptr(xyz), _hip.ptr(feat), _hip.ptr(idx),
                              _hip.ptr(ww), _hip.ptr(bw), _hip.ptr(out))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(4):
                run()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"decode_cell b={b} n={n} c={c} k={k}: {us:7.1f} us  {out.numel() * 4 / us / 1e6:5.2f} TB/s written   checksum {out.double().sum().item():.6e}", flush=True)
