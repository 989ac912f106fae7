#!/bin/bash
# GPU_MAX_HW_QUEUES (HIP runtime: hardware queues the process's streams are spread over, default 4) against the bench:
# the headline, and rank 1 of 4 of configs[3] under each chain schedule.
show='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j["ms_per_step"], j["value"], j["config"].get("pre_chain_choice"))'
for h in default 16 32 default 16; do
  if [ $h = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$h; fi
  echo "== GPU_MAX_HW_QUEUES=$h"
  python bench.py --steps 60 --warmup 10 --no-legs 2>/dev/null | python -c "$show"
  for c in sharded redundant lagged; do
    RGBDR_BENCH_CHAIN=$c python bench.py --slab 1/4 --steps 40 --warmup 10 --no-legs 2>/dev/null | python -c "$show"
  done
done
