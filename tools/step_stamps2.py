"""kNN / tree kernel time of one level's search inside the captured cls step: python tools/step_stamps2.py N"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PASNL_STAMP_N"] = sys.argv[1]
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(ROOT, "pointasnl_amd", "csrc", "libpasnl_hip_tuning.so")
import numpy as np
import bench
from pointasnl_amd.utils import pointasnl_util as U
from pointasnl_amd.utils.nearest_neighbors.lib.python import nearest_neighbors as NN
seen = []
orig = NN._knn_ref_dev
def wrapped(pts, queries, K, i64, out, max_workgroups, stats):
    st = [] if stats is None else stats
    r = orig(pts, queries, K, i64, out, max_workgroups, st)
    seen.append((tuple(pts.shape), tuple(queries.shape), st[0]))
    return r
NN._knn_ref_dev = wrapped
for env in ("PASNL_KNN_REF_NO_TREE", None):
    os.environ.pop("PASNL_KNN_REF_NO_TREE", None)
    if env: os.environ[env] = "1"
    U.KNN_TIE_ORDER = "reference"
    for rep in range(2):
        r = bench.run_config(1, dict(bench.WORKLOADS[1]), 20, 5, graph=True, kernel_pass=False, announce=False, pipeline="prefetch", extra_blocks=0)
        buf = (ctypes.c_ulonglong * 16)()
        assert _hip.lib().pasnl_tuning_stamps_read(buf) == 0
        t = np.array(list(buf), dtype=np.float64) / 100.0
        print(env or "default", "ms/step %.4f | kNN %.1f us | tree kernel %.1f us" % (r["ms_per_step"], t[7] - t[6], (t[2] - t[7]) if not env else 0.0), flush=True)
import torch
torch.cuda.synchronize()
for sh, qs, nf in seen[-6:]:
    print("search", sh, qs, "listed", int(nf.sum()))
