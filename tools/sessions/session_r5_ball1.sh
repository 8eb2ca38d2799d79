#!/bin/bash
# round 5, session 1: parity of the rewritten ball query (+ the other round-5 changes so far), A/B against the round-4 kernel
# on the same box, phase probe, counters
O=gpurun_out/r05a; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py tests/test_gpu_properties.py tests/test_gpu_ref_kernels.py -q -x -k "ball" 2>&1 | tail -5 | tee $O/pytest_ball.txt
python -m pytest tests/test_gpu_cells.py -q -x -k "single or 16_channels or gather_fused or wide" 2>&1 | tail -3 | tee $O/pytest_cells.txt
python -m pytest tests/test_gpu_reference_fixtures.py -q -x -k "tie_order" 2>&1 | tail -3 | tee $O/pytest_tie.txt
for i in 1 2; do
python tools/ball_ab.py pointasnl_amd/csrc/libpasnl_hip_r04ball.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ball_ab.txt
python tools/ball_ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ball_ab.txt
done
python tools/ballprobe.py 2>&1 | grep -v amdgpu.ids | tee $O/ballprobe.txt
bash profiles/collect_ball_counters.sh > /dev/null 2>&1; python profiles/ball_counters_summary.py gpurun_out $O/ball_counters.json 2>&1 | tail -30 | tee $O/ball_counters.txt
