"""Diagnostics: layer-wise teacher-forced errors of a model case (see tests/test_gpu_reference_fixtures.py)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden import ref_cases as R
from oracle import cells, weights
import test_gpu_reference_fixtures as T


class MP:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


name = sys.argv[1]
case = next(c for c in R.MODEL_CASES if c["name"] == name)
gold = np.load(os.path.join(ROOT, "tests/golden/ref_models.npz"))
T.load_store(gold, name, R.model_seed(case))
params = cells.params_from_tf(weights.make_all(R.model_seed(case), json.loads(str(gold[f"{name}/vars"]))))
pc = R.model_input(case)[:8]
errs, le = T.teacher_forced_errors(case, pc, params, MP())
for e in errs:
    print("%-12s %-10s %.2e" % e)
print("logits", le)
