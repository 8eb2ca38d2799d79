#!/bin/bash
# round 5: ball query with the chunk's queries in cell order -- parity, A/B against the previous kernel on the same box, counters
O=gpurun_out/r05q; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_properties.py tests/test_gpu_ref_kernels.py -q -x -k "ball" 2>&1 | tail -5 | tee $O/pytest_ball.txt
for i in 1 2; do
timeout 100 python tools/ball_ab.py pointasnl_amd/csrc/libpasnl_hip_bgold.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ball_ab.txt
timeout 100 python tools/ball_ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ball_ab.txt
done
bash profiles/collect_ball_counters.sh > /dev/null 2>&1; python profiles/ball_counters_summary.py gpurun_out $O/ball_counters.json 2>&1 | tail -30 | tee $O/ball_counters.txt
