"""Diagnostics: per-point error of a model case of tests/golden/ref_models.npz on the GPU vs the oracle (fp64 / fp32)."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden import ref_cases as R
from oracle import cells, weights
import test_gpu_reference_fixtures as T

name = sys.argv[1]
case = next(c for c in R.MODEL_CASES if c["name"] == name)
gold = np.load(os.path.join(ROOT, "tests/golden/ref_models.npz"))
T.load_store(gold, name, R.model_seed(case))
pc = R.model_input(case)
logits, ep = T.run_model(case, pc)
got = logits.cpu().numpy()
params = cells.params_from_tf(weights.make_all(R.model_seed(case), json.loads(str(gold[f"{name}/vars"]))))
fwd = {"sem_seg": cells.sem_seg_forward, "sem_seg_res": cells.sem_seg_res_forward}[case["model"]]
for b in range(pc.shape[0]):
    w64 = fwd(pc[b:b + 1], params, case["kw"]["num_class"], dtype=np.float64, feature_channel=case["kw"]["feature_channel"])
    w32 = fwd(pc[b:b + 1], params, case["kw"]["num_class"], dtype=np.float32, feature_channel=case["kw"]["feature_channel"])
    sc = np.abs(w64).max()
    e = np.abs(got[b] - w64[0]).max(-1) / sc
    e32 = np.abs(w32[0] - w64[0]).max(-1) / sc
    print(f"cloud {b}: scale {sc:.1f}  gpu-vs-f64: max {e.max():.2e}, points >2e-4: {(e > 2e-4).sum()}, >1e-5: {(e > 1e-5).sum()}, median {np.median(e):.2e} | "
          f"oracle f32-vs-f64: max {e32.max():.2e}, >2e-4: {(e32 > 2e-4).sum()}, >1e-5: {(e32 > 1e-5).sum()}, median {np.median(e32):.2e}")
    worst = np.argsort(e)[-5:]
    print("  worst points", worst, e[worst], "|logit| at worst", np.abs(w64[0][worst]).max(-1), "xyz", pc[b, worst[-1]])
