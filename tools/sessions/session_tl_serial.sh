#!/bin/bash
# kernel timeline of one replayed SERIAL forward: bash tools/sessions/session_tl_serial.sh <model> <marker kernel substring>
M=${1:-sem_seg_res}; MARK=${2:-"fps_pruned_kernel"}
O=gpurun_out/r04q; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/tl; PASNL_BENCH_TRACE_ONLY=1 timeout 300 rocprofv3 --kernel-trace -d $O/tl -o tl -f csv -- python bench.py --worker --model $M --pipeline serial --steps 6 --warmup 2 --no-cpu-baseline --no-others > /dev/null 2>&1
python tools/timeline.py $O/tl 2 "$MARK" > $O/timeline_${M}_serial.txt; cat $O/timeline_${M}_serial.txt
rm -rf $O/tl
