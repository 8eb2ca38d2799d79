#!/bin/bash
O=gpurun_out/r04i; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/tl; timeout 300 rocprofv3 --kernel-trace -d $O/tl -o tl -f csv -- python bench.py --worker --steps 6 --warmup 2 --no-cpu-baseline --no-others --pipeline serial > /dev/null 2>&1
python tools/timeline.py $O/tl 2 > $O/timeline_cls_serial.txt; tail -70 $O/timeline_cls_serial.txt
rm -rf $O/tl
