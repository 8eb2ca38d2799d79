"""Determinism of the forked searches: the same forward run N times eagerly and as a replayed graph must be bit-identical,
and identical to the forward with every kernel on one stream (OVERLAP off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import torch
import bench as B
from pointasnl_amd.utils import tf_util, pointasnl_util as U
model, bsz, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
M = importlib.import_module(f"pointasnl_amd.models.pointasnl_{model}")
x = torch.from_numpy(B.synth_clouds(5, bsz, n)).cuda()
tf_util.set_store(tf_util.VariableStore(seed=1))
def fwd():
    with torch.no_grad():
        if model == "cls":
            return M.get_model(x, is_training=False, adaptive_sample=True)[0]
        return M.get_model(x, False, 20)[0]
U.OVERLAP = False
ref = fwd().clone(); torch.cuda.synchronize()
U.OVERLAP = True
bad = 0
for i in range(10):
    out = fwd(); torch.cuda.synchronize()
    if not torch.equal(out, ref):
        bad += 1
        print("eager run", i, "differs: max abs", float((out - ref).abs().max()), "frac", float((out != ref).float().mean()))
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
    gout = fwd()
torch.cuda.current_stream().wait_stream(s)
for i in range(10):
    g.replay(); torch.cuda.synchronize()
    if not torch.equal(gout, ref):
        bad += 1
        print("graph replay", i, "differs: max abs", float((gout - ref).abs().max()), "frac", float((gout != ref).float().mean()))
print(model, "mismatches:", bad)
