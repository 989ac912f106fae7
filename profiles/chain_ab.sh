for c in sharded lagged sharded lagged; do
  RGBDR_BENCH_CHAIN=$c python bench.py --slab 1/4 --steps 40 --warmup 5 --no-legs 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', j['ms_per_step'], j['slab']['integrate_ms'], j['config']['pre_chain_choice'])"
done
