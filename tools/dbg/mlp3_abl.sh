for a in "" 1 2 3 4 7; do
python - <<PY 2>&1 | grep -v amdgpu | grep fused
import os, sys
sys.path.insert(0, "/root/repo")
from pointasnl_amd import _hip
if "$a": _hip.LIB_PATH = os.path.abspath("pointasnl_amd/csrc/libpasnl_hip_mpabl$a.so")
print("ABL=[$a]", flush=True)
exec(open("tools/dbg/mlp3_time.py").read())
PY
done
