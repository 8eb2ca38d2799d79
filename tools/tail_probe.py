"""pasnl_sa_tail alone, replayed from a HIP graph: time per launch at the two cls shapes (tuning build: PASNL_TAIL_ABL
ablations).   PASNL_TAIL_ABL=<mask> python tools/tail_probe.py"""
import os
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from pointasnl_amd import _hip  # noqa: E402

if os.environ.get("PASNL_PROBE_LIB") == "tuning":
    _hip.LIB_PATH = _hip.LIB_PATH.replace("libpasnl_hip.so", "libpasnl_hip_tuning.so")
elif os.environ.get("PASNL_PROBE_LIB", "").endswith(".so"):  # an alternative build to compare against
    _hip.LIB_PATH = os.environ["PASNL_PROBE_LIB"]

g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *sh: torch.randn(sh, device="cuda", generator=g)
for rows, w, cb, c in [(32768, 9, 32, 128), (8192, 134, 64, 256), (16384, 70, 32, 128)]:
    after, skip, att = r(rows, c), r(rows, w), r(rows, cb)
    ws, bs, wb, bb, wagg, bagg = r(w, c) * .1, r(c), r(cb, c) * .1, r(c), r(c, c) * .1, r(c)
    out = torch.empty(rows, c, device="cuda")
    big = torch.empty(64 << 20, device="cuda")  # 256 MB written between launches: the inputs come from HBM, as in the forward

    def run():
        _hip.launch("pasnl_sa_tail", "sa_tail", rows, w, cb, c, _hip.ptr(after), _hip.ptr(skip), _hip.ptr(att), _hip.ptr(ws),
                    _hip.ptr(bs), _hip.ptr(wb), _hip.ptr(bb), _hip.ptr(wagg), _hip.ptr(bagg), _hip.ptr(out))

    side = torch.cuda.Stream()
    res = []
    for flush in (False, True):
        with torch.cuda.stream(side):
            run()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(10):
                    if flush:
                        big.zero_()
                    run()
            graph.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / 100)
    with torch.cuda.stream(side):  # the flush alone
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(10):
                big.zero_()
        graph.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            graph.replay()
        e1.record(); torch.cuda.synchronize()
        zt = e0.elapsed_time(e1) * 1e3 / 100
    print(f"sa_tail rows={rows} w={w} cb={cb} c={c}: hot {res[0]:6.1f} us   cold {res[1] - zt:6.1f} us (flush {zt:.1f})", flush=True)
