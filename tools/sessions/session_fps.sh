#!/bin/bash
O=gpurun_out/r04g; mkdir -p $O
timeout 600 python tools/fps_batch_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/fps_batch_sweep.txt
