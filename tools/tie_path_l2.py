"""The cls step's layer-2 search (512 points, 128 queries) with its listed queries, alone: which form takes them, how long.
python tools/tie_path_l2.py  (tuning build)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(ROOT, "pointasnl_amd", "csrc", "libpasnl_hip_tuning.so")
import numpy as np, torch
import bench
from pointasnl_amd.utils import pointasnl_util as U
from pointasnl_amd.utils.nearest_neighbors.lib.python import nearest_neighbors as NN
import pointasnl_amd as P
seen = []
orig = NN._knn_ref_dev
def wrapped(pts, queries, K, i64, out, max_workgroups, stats):
    st = [] if stats is None else stats
    r = orig(pts, queries, K, i64, out, max_workgroups, st)
    seen.append((pts.clone(), queries.clone(), K, st[0]))
    return r
NN._knn_ref_dev = wrapped
U.KNN_TIE_ORDER = "reference"
CFG = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bench.run_config(CFG, dict(bench.WORKLOADS[CFG]), 2, 1, graph=False, kernel_pass=False, announce=False, pipeline="serial", extra_blocks=0)
torch.cuda.synchronize()
NN._knn_ref_dev = orig
cases = [(p, q, K) for p, q, K, nf in seen]
print("searches of one forward:", [(tuple(p.shape[:2]), q.shape[1], K, int(nf.sum())) for p, q, K, nf in seen[:len(seen) // 2]])
done = set()
for p, q, K in cases:
    key = (tuple(p.shape), tuple(q.shape), K)
    if key in done: continue
    done.add(key)
    paths = (ctypes.c_int * 8)()
    _hip.lib().pasnl_tie_paths_read(paths, 1)
    stats = []
    for _ in range(3):
        stats = []
        P.nearest_neighbors.knn_batch(p, q, K, dtype=torch.int32, stats=stats)
    torch.cuda.synchronize()
    _hip.lib().pasnl_tie_paths_read(paths, 1)
    nf = stats[0].cpu().numpy()
    print("search", tuple(p.shape), tuple(q.shape), "K", K, "listed per cloud", nf[nf > 0].tolist(), "| clouds by form over 3 runs: sets", paths[0], "records moved", paths[1], "tree", paths[2], "(set form returned 1:", paths[3], "2:", paths[4], ")")
    buf = (ctypes.c_ulonglong * 32)()
    _hip.lib().pasnl_knn_small_probe_read(buf)
    t = np.array(list(buf), dtype=np.float64)
    u = lambda a, b: (t[b] - t[a]) / 100.0
    print("   first listed cloud: entry -> tied points listed %.1f" % u(0, 3), "| levels", [round(u(3 + i, 4 + i), 1) for i in range(12) if t[4 + i] > t[3 + i]], "| total %.1f (x100 cycles)" % u(0, 31), "| node 0: load %.1f, last pass %.1f, reductions %.1f, bookkeeping %.1f" % (u(3, 16), u(16, 17), u(17, 18), u(18, 4)))
    wg = (ctypes.c_ulonglong * 64)()
    if _hip.lib().pasnl_tie_path_wg_read(wg) == 0 and (p.shape[1] > 2048 or K > 64):
        nl = int((nf > 0).sum())
        print("   standalone tie-path kernel, per listed cloud: x100 cycles", [round(wg[i] / 100) for i in range(min(nl, 32))],
              "| 1000 tied + 10 distinct points + code", [int(wg[32 + i]) for i in range(min(nl, 32))])
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(50): P.nearest_neighbors.knn_batch(p, q, K, dtype=torch.int32)
    ev1.record(); torch.cuda.synchronize()
    t_def = ev0.elapsed_time(ev1) * 20
    ev0.record()
    for _ in range(50): P.nearest_neighbors.knn_batch(p, q, K, dtype=torch.int32, tie_order="index")
    ev1.record(); torch.cuda.synchronize()
    print("   %.1f us per search (50 back to back); canonical order %.1f" % (t_def, ev0.elapsed_time(ev1) * 20))
