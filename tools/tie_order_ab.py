"""cls B=64 step: the default (reference) neighbour order against the canonical one, alternating inside ONE process.
python tools/tie_order_ab.py [cfg]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from pointasnl_amd.utils import pointasnl_util as U
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
res = {"reference": [], "index": []}
for rnd in range(4):
    for order in ("reference", "index"):
        U.KNN_TIE_ORDER = order
        r = bench.run_config(cfg, dict(bench.WORKLOADS[cfg]), 20, 5, graph=True, kernel_pass=False, announce=False, pipeline="prefetch", extra_blocks=2)
        res[order].append(float(np.median(r["block_ms"])))
        print(rnd, order, [round(v, 4) for v in r["block_ms"]], flush=True)
U.KNN_TIE_ORDER = "reference"
a, b = np.median(res["reference"]), np.median(res["index"])
print(f"cfg{cfg}: reference {a:.4f} ms, canonical {b:.4f} ms, +{100 * (a / b - 1):.2f} %")
