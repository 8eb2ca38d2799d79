#!/bin/bash
out=gpurun_out/r04b; mkdir -p $out
python tools/ballprobe.py 2>&1 | grep -v amdgpu.ids | tee $out/ballprobe_sort.txt
PASNL_BALL_NOSORT=1 python tools/ballprobe.py 2>&1 | grep -v amdgpu.ids | tee $out/ballprobe_nosort.txt
