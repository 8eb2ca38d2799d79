#!/bin/bash
out=gpurun_out/s17; mkdir -p $out; export TMPDIR=/tmp
export PASNL_BENCH_WATCHDOG=150
timeout 200 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/e0; cut -c1-700 $out/bench.json | sed 's/"roofline.*//' ; echo
timeout 200 python bench.py --no-cpu-baseline --pipeline search 2>/dev/null | cut -c1-190; echo; timeout 200 python bench.py --no-cpu-baseline --pipeline serial 2>/dev/null | cut -c1-190; echo
timeout 200 python bench.py --no-cpu-baseline --AS 2>/dev/null | cut -c1-190; echo
timeout 200 python bench.py --no-cpu-baseline --force-dist 2>/dev/null | cut -c1-190; echo
timeout 200 python bench.py --model sem_seg --steps 20 --warmup 3 > $out/bench_sem_seg.json 2> $out/e1; cut -c1-190 $out/bench_sem_seg.json; echo; tail -3 $out/e1 | cut -c1-200
timeout 200 python bench.py --model sem_seg_res --steps 20 --warmup 3 > $out/bench_sem_seg_res.json 2> $out/e2; cut -c1-190 $out/bench_sem_seg_res.json; echo
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
