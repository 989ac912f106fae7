#!/bin/bash
# the LUT arena as a plain allocation (the library's candidate shopping) against a range of the fastest 1-GiB chunks (forced)
show='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(j["ms_per_step"], r["avg_launch_ms"], r["frac"], r.get("arena_placement_probe_ms"), r.get("arena_kept"), r.get("arena_chunks"), r.get("arena_chunks_replay_ms"))'
for args in "" "--sensors 8" "--sensors 8 --grid 1024 --steps 10"; do
  for mode in 0 force; do
    echo "== bench.py $args, RGBDR_ARENA_CHUNKS=$mode"
    RGBDR_ARENA_CHUNKS=$mode python bench.py --no-cpu-baseline --no-legs $args 2>/dev/null | python -c "$show"
  done
done
