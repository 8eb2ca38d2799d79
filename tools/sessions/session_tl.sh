#!/bin/bash
# kernel timeline of one replayed prefetch step: bash tools/sessions/session_tl.sh <model> <marker kernel substring>
M=${1:-cls}; MARK=${2:-"sa_cell_kernel<64, 64, 8, false, true, true>"}
O=gpurun_out/r04i; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/tl; PASNL_BENCH_TRACE_ONLY=1 timeout 300 rocprofv3 --kernel-trace -d $O/tl -o tl -f csv -- python bench.py --worker --model $M --steps 6 --warmup 2 --no-cpu-baseline --no-others > /dev/null 2>&1
python tools/timeline.py $O/tl 2 "$MARK" > $O/timeline_${M}_prefetch.txt; cat $O/timeline_${M}_prefetch.txt
rm -rf $O/tl
