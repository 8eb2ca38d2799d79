"""python tools/step_stamps.py [cfg] [N]: where a step's searches on N-point clouds spend their time INSIDE the captured step (tuning build; one-thread stamp kernels on the
prefix's streams): sampler start / end, kNN start / end, tree kernel end.  python tools/step_stamps.py [index|reference]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = int(sys.argv[1]) if len(sys.argv) > 1 else 1
os.environ["PASNL_STAMP_N"] = sys.argv[2] if len(sys.argv) > 2 else "1024"
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(ROOT, "pointasnl_amd", "csrc", "libpasnl_hip_tuning.so")
import numpy as np
import bench
from pointasnl_amd.utils import pointasnl_util as U
for order, env in (("index", None), ("reference", "PASNL_KNN_REF_NO_TREE"), ("reference", None)):
    os.environ.pop("PASNL_KNN_REF_NO_TREE", None)
    if env: os.environ[env] = "1"
    U.KNN_TIE_ORDER = order
    rows = []
    for rep in range(3):
        r = bench.run_config(CFG, dict(bench.WORKLOADS[CFG]), 20, 5, graph=True, kernel_pass=False, announce=False, pipeline="prefetch", extra_blocks=0)
        buf = (ctypes.c_ulonglong * 16)()
        assert _hip.lib().pasnl_tuning_stamps_read(buf) == 0
        t = np.array(list(buf), dtype=np.float64) / 100.0  # us
        rows.append((r["ms_per_step"], t[4] - t[6], t[5] - t[4], t[7] - t[6], t[2] - t[7] if not env and order == "reference" else 0.0, t[5] - t[7]))
    for row in rows:
        print(order, env or "", "ms/step %.4f | sampler starts %+.1f us after the kNN kernel, runs %.1f | kNN %.1f | tree kernel %.1f | sampler ends %+.1f us after the kNN kernel" % row, flush=True)
