#!/bin/bash
out=gpurun_out/s14; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cells.py -m gpu -x -q -k repulsion 2>&1 | tail -2
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o seg -- python bench.py --model sem_seg --steps 5 --warmup 2 --no-cpu-baseline > $out/prof_bench.json 2> $out/prof.err
python profiles/summarize_rocpd.py $out/prof/seg_results.db $out/sem_seg_kernel_stats.csv; head -30 $out/sem_seg_kernel_stats.csv | cut -c1-150
