#!/bin/bash
cd /root/repo
run() { timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=[x for x in d['kernels'] if x['kernel'].startswith('pasnl_sa_cell')];print(d['value'],d['ms_per_step'],[(x['dims'][5],x['avg_us']) for x in k][:3], len(d['config']['switches']))"; }
ALT="_hip.LIB_PATH='/root/repo/pointasnl_amd/csrc/libpasnl_hip_old.so'"
run; run --set "$ALT"; run; run --set "$ALT"
