"""Where the default neighbour order's cost on a step comes from (tuning build): canonical / flags only / full default.
python tools/tie_cost_split.py [cfg]  (the tuning build's probes make its kernels a little slower than the product library's:
tools/tie_order_ab.py is the A/B on the product library)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(ROOT, "pointasnl_amd", "csrc", "libpasnl_hip_tuning.so")
import numpy as np
import bench
from pointasnl_amd.utils import pointasnl_util as U
res = {}
CFG = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for rnd in range(3):
    for tag, order, env in (("canonical", "index", None), ("flags, no tree kernels", "reference", "PASNL_KNN_REF_NO_TREE"), ("tree stage for clouds <= 2048 points only", "reference", "PASNL_KNN_REF_NO_TREE_ABOVE=2048"), ("tree stage for clouds > 2048 points only", "reference", "PASNL_KNN_REF_NO_TREE_BELOW=2049"), ("default", "reference", None)):
        for e in ("PASNL_KNN_REF_NO_TREE", "PASNL_KNN_REF_NO_TIE_PATH", "PASNL_KNN_REF_TIE_PATH_ONLY", "PASNL_KNN_REF_NO_TREE_ABOVE", "PASNL_KNN_REF_NO_TREE_BELOW"): os.environ.pop(e, None)
        if env: os.environ[env.split("=")[0]] = env.split("=")[1] if "=" in env else "1"
        U.KNN_TIE_ORDER = order
        r = bench.run_config(CFG, dict(bench.WORKLOADS[CFG]), 20, 5, graph=True, kernel_pass=False, announce=False, pipeline="prefetch", extra_blocks=2)
        res.setdefault(tag, []).append(float(np.median(r["block_ms"])))
        print(rnd, tag, [round(v, 4) for v in r["block_ms"]], flush=True)
for k, v in res.items(): print("cfg", CFG, k, round(float(np.median(v)), 4))
