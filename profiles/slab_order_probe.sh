#!/bin/bash
# Is "rank 2 of configs[3] / 4 is the slowest slab"  geometry or placement?  The four slabs in
# four FRESH processes each (one context per process), in two orders, and -- for comparison -- in one process (--slab-sweep 4,
# where slab r is the r-th context the process creates).   bash profiles/slab_order_probe.sh
OUT=$PWD/gpurun_out/slab_order; rm -rf $OUT; mkdir -p $OUT
for order in "0 1 2 3" "2 0 3 1"; do
  for r in $order; do
    python3 bench.py --slab $r/4 --steps 40 --warmup 5 --no-legs > $OUT/o$(echo $order | tr -d ' ')_r$r.json 2>/dev/null
  done
done
python3 bench.py --slab-sweep 4 --steps 40 --warmup 5 > $OUT/sweep.json 2>/dev/null
python3 - $OUT <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for order in ("0123", "2031"):
    row = []
    for r in order:
        j = json.loads(open("%s/o%s_r%s.json" % (out, order, r)).read().strip().splitlines()[-1])
        row.append("slab %s: %.4f ms/step, sweep %.4f, replay %.4f" % (r, j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["box_stream_replay_ms"]))
    print("fresh processes, order %s:  " % order + " | ".join(row))
j = json.loads(open(out + "/sweep.json").read().strip().splitlines()[-1])
rows = j.get("slabs") or j.get("per_slab") or []
print("one process (--slab-sweep 4):", json.dumps([{k: s.get(k) for k in ("rank", "ms_per_step", "integrate_ms")} for s in rows]) if rows else list(j.keys()))
PY
