#!/bin/bash
out=gpurun_out/s3; mkdir -p $out; export TMPDIR=/tmp
bash tools/gpu_session.sh s3 tests bench prof pmc
for v in "--pipeline 1" "--pipeline 2"; do
  timeout 120 python bench.py --model sem_seg --steps 5 --warmup 2 $v > "$out/sem_seg_${v// /_}.json" 2> "$out/sem_seg_${v// /_}.err"; echo "sem_seg $v rc=$?"; cut -c1-200 "$out/sem_seg_${v// /_}.json"; echo
done
