"""Phase cycles of the two-pass kNN kernel (probe build: grouping.hip with -DPASNL_KNN_PROBE linked into
pointasnl_amd/csrc/libpasnl_hip_probe_knn.so).   python tools/knn_probe.py"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libpasnl_hip_probe_knn.so")
lib = _hip.lib()
read = lib.pasnl_knn_probe_read
read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
g = torch.Generator(device="cuda").manual_seed(1)
for (b, n, m, k, name) in [(64, 1024, 512, 32, "cls-L1"), (64, 512, 128, 64, "cls-L2"), (16, 8192, 1024, 32, "scannet-L1")]:
    sup = torch.rand((b, n, 3), device="cuda", generator=g)
    qry = sup[:, :m].contiguous()
    idx = torch.empty((b, m, k), dtype=torch.int32, device="cuda")
    run = lambda: _hip.launch("pasnl_knn_batch", "knn_batch", b, n, m, k, _hip.ptr(sup), _hip.ptr(qry), _hip.ptr(idx), 0, None)
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    read(buf)
    reps = 50
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    read(buf)
    v = [x / reps for x in buf]
    waves = max(1.0, v[4])
    print(json.dumps({"shape": name, "waves": round(waves), "pass1": round(v[0] / waves), "bound": round(v[1] / waves),
                      "pass2": round(v[2] / waves), "emit": round(v[3] / waves)}), flush=True)
