#!/bin/bash
# Projection for bench.py's N > 1 default (the N = 1 workload at fixed work per GPU), all on ONE box in one call:
# the N = 1 frame and every rank of the 2- / 4- / 8-GPU weak-scaling grids run alone on that GPU (bench.py --weak
# --slab-sweep k: sharded pre_* chain, library-managed RCCL exchange with itself); efficiency = t1 / slowest rank's frame.
# A projection: no scaling curve was measured.
#   bash profiles/collect_weak_scaling.sh <tag>   -> gpurun_out/weak_<tag>/summary.json
TAG=${1:-r04}
OUT=$PWD/gpurun_out/weak_$TAG
rm -rf $OUT && mkdir -p $OUT
python3 bench.py --no-cpu-baseline --steps 30 > $OUT/t1.json 2>/dev/null
for k in 2 4 8; do
  python3 bench.py --weak --slab-sweep $k --steps 30 --warmup 5 > $OUT/weak_sweep$k.json 2>/dev/null
done
python3 - $OUT <<'PY'
import json, sys
out = sys.argv[1]
L = lambda n: json.load(open("%s/%s.json" % (out, n)))
t1 = L("t1")
res = {"note": "projection from single-GPU per-rank runs on one box -- no scaling curve was measured",
       "t1_ms_per_frame": t1["ms_per_step"], "t1_integrate_ms": t1["roofline"]["avg_launch_ms"], "t1_roofline_frac": t1["roofline"]["frac"]}
for k in (2, 4, 8):
    s = L("weak_sweep%d" % k)
    ranks = s["ranks"]
    worst = max(r["ms_per_step"] for r in ranks)
    res["weak%d" % k] = {"grid": s["config"]["grid"], "sensors": s["config"]["sensors"], "schedule": ranks[0].get("schedule"),
                         "rank_ms_per_frame": [r["ms_per_step"] for r in ranks], "rank_integrate_ms": [r["integrate_ms"] for r in ranks],
                         "rank_roofline_frac": [r["roofline_frac"] for r in ranks], "slowest_rank_ms": worst,
                         "projected_efficiency": round(t1["ms_per_step"] / worst, 4),
                         "projected_efficiency_mean_rank": round(t1["ms_per_step"] / (sum(r["ms_per_step"] for r in ranks) / len(ranks)), 4),
                         "projected_value_mvoxels_per_s": round(k * 512 ** 3 / (worst * 1e-3) / 1e6, 1)}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
