mkdir -p gpurun_out/r3h
for rep in 1 2; do for mdl in "cls" "cls --AS" "sem_seg" "sem_seg_res"; do for v in "serial 3,4" "prefetch 3,4" "prefetch 0,1"; do
set -- $v
PASNL_BENCH_PREFETCH_SLOTS=$2 python bench.py --model $mdl --pipeline $1 --steps 20 --warmup 6 --no-cpu-baseline --no-others 2> gpurun_out/r3h/x.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep$rep $mdl $v', d['ms_per_step'], d['config']['outputs_agree'], d['config'].get('serial') and d['config']['serial']['ms_per_step'])"
done; done; done
