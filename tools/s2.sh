#!/bin/bash
out=gpurun_out/s2; mkdir -p $out; export TMPDIR=/tmp
PASNL_TRACE=1 timeout 150 python bench.py --model sem_seg --no-graph --steps 1 --warmup 1 > $out/trace_sem_seg.json 2> $out/trace_sem_seg.err; echo "sem_seg trace rc=$?"; tail -4 $out/trace_sem_seg.err
for p in 3 4; do timeout 300 python bench.py --pipeline $p --no-cpu-baseline > $out/bench_p$p.json 2> $out/bench_p$p.err; cut -c1-200 $out/bench_p$p.json; echo; done
bash tools/gpu_session.sh s2 pmc
