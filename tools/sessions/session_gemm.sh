#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
for m in sem_seg sem_seg_res cls; do timeout 300 python tools/gemm_audit.py $m 2>&1 | grep -v amdgpu.ids | tee $O/gemm_audit_$m.txt | head -24; done
