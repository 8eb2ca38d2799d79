# cls step with / without layer 2's search in the prefetched prefix (PASNL_BENCH_PREFIX_L2), and with other fork points
for rep in 1 2; do
for v in "0:cell2" "1:cell2" "1:conv2" "1:head"; do
  l2=${v%%:*}; at=${v#*:}
  PASNL_BENCH_PREFIX_L2=$l2 PASNL_BENCH_FORK_AT=$at python bench.py --worker --steps 20 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefix_l2=$l2 fork_at=$at', d['ms_per_step_blocks']['blocks'], d.get('outputs_agree'))"
done; done
