#!/bin/bash
# mlp3_pool with parts of it ablated (results are wrong, times are what is measured).  Build the diagnostic libraries first:
#   cd pointasnl_amd/csrc; for a in 1 2 3 4 7; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
#     -DPASNL_MLP3_ABL=$a -c mlp_pool.hip -o /tmp/mp_$a.o && hipcc --offload-arch=gfx950 -shared -fPIC \
#     $(ls *.o | grep -v "mlp_pool\|_tuning\|probe") /tmp/mp_$a.o -o libpasnl_hip_mpabl$a.so; done
# mask: 1 = no weight loads, 2 = no LDS operand reads, 4 = no tile load
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
