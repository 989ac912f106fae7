#!/bin/bash
# the slab run several times in a row (a process that follows another is where the lagged schedule's host stall showed):
# which schedule is kept, what the headline reads, whether the lagged headline had to be discarded
for c in auto auto auto lagged lagged; do
  if [ $c = auto ]; then unset RGBDR_BENCH_CHAIN; else export RGBDR_BENCH_CHAIN=$c; fi
  python bench.py --slab 1/4 --steps 40 --warmup 10 --no-legs 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config'].get('pre_chain_choice')
print('$c', j['ms_per_step'], j.get('host_enqueue_ms_per_step'), c)"
done
