import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import pointasnl_amd as P
from pointasnl_amd.utils.nearest_neighbors.lib.python import nearest_neighbors as NN
torch.manual_seed(0)
x = torch.rand(2, 256, 3, device="cuda"); q = x[:, :64].contiguous()
e = NN.knn_batch(x, q, 16, dtype=torch.int32, tie_order="nanoflann")
torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
    o = NN.knn_batch(x, q, 16, dtype=torch.int32, tie_order="nanoflann")
print("captured flags", len(NN._DEFERRED_FLAGS))
for i in range(3):
    g.replay(); torch.cuda.synchronize()
    f = NN._DEFERRED_FLAGS[0]
    base = f.untyped_storage()
    hdr = torch.tensor([], dtype=torch.int32, device="cuda").set_(base, 0, (80,))
    print(i, "flag", [int(t.item()) for t in NN._DEFERRED_FLAGS], "equal", bool(torch.equal(o, e)), hdr[:8].tolist(), hdr[64:72].tolist())
