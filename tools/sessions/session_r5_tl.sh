#!/bin/bash
# kernel timeline of one replayed prefetch step: bash tools/sessions/session_r5_tl.sh <model> [tag]
M=${1:-cls}; TAG=${2:-r05}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/tl; PASNL_BENCH_TRACE_ONLY=1 timeout 300 rocprofv3 --kernel-trace -d $O/tl -o tl -f csv -- python bench.py --worker --model $M --steps 6 --warmup 2 --no-cpu-baseline --no-others > /dev/null 2>&1
MARK="sa_cell_kernel<64, 64, 8, false, true, true"; [ $M = sem_seg_res ] && MARK="sa_cell16_kernel"; [ $M = sem_seg ] && MARK="sa_cell_kernel<32, 32"
python tools/timeline.py $O/tl 2 "$MARK" > $O/timeline_${M}_prefetch.txt
rm -rf $O/tl
