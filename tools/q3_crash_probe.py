"""GPU_MAX_HW_QUEUES=3: does a captured segmentation forward replay?  (tests/test_gpu_overlap.py crashed in hipGraphLaunch)
python tools/q3_crash_probe.py <model> <bsz> <n> <prio>"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
from pointasnl_amd.utils import pointasnl_util as U, tf_util
model, bsz, n, prio = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
M = importlib.import_module(f"pointasnl_amd.models.pointasnl_{model}")
x = torch.from_numpy(B.synth_clouds(5, bsz, n)).cuda()
tf_util.set_store(tf_util.VariableStore(seed=1))
def fwd():
    with torch.no_grad():
        return M.get_model(x, is_training=False, adaptive_sample=True)[0] if model == "cls" else M.get_model(x, False, 20)[0]
for _ in range(3): fwd()
torch.cuda.synchronize()
s = torch.cuda.Stream(priority=prio); s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
    out = fwd()
torch.cuda.current_stream().wait_stream(s)
for i in range(5):
    g.replay(); torch.cuda.synchronize()
print("ok", model, bsz, n, prio, os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
