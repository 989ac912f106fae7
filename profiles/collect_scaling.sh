#!/bin/bash
# The strong-scaling projection DESIGN.md section 6 quotes, all on ONE box in one call: the 1-GPU denominators
# (8 sensors -> 512^3 and -> 1024^3 on one GPU) and every rank of configs[3] / configs[4] run alone on that GPU
# (bench.py --slab-sweep: sharded pre_* chain, library-managed RCCL exchange with itself), then
# efficiency = t1 / (k * slowest rank's frame).  A projection: no scaling curve was measured.
#   bash profiles/collect_scaling.sh <tag>   -> gpurun_out/scaling_<tag>/ and profiles-ready summary.json
TAG=${1:-r04}
OUT=$PWD/gpurun_out/scaling_$TAG
rm -rf $OUT && mkdir -p $OUT
python3 bench.py --sensors 8 --no-cpu-baseline --steps 30 > $OUT/t1_512.json 2>/dev/null
python3 bench.py --slab-sweep 4 --steps 30 --warmup 5 > $OUT/slab_sweep4.json 2>/dev/null
python3 bench.py --slab-sweep 4 --steps 30 --warmup 5 --no-shard --torch-collectives > $OUT/slab_sweep4_round3_schedule.json 2>/dev/null
python3 bench.py --sensors 8 --grid 1024 --no-cpu-baseline --steps 20 > $OUT/t1_1024.json 2>/dev/null
python3 bench.py --slab-sweep 8 --steps 30 --warmup 5 > $OUT/slab_sweep8.json 2>/dev/null
python3 - $OUT <<'PY'
import json, sys
out = sys.argv[1]
L = lambda n: json.load(open("%s/%s.json" % (out, n)))
res = {"note": "projection from single-GPU per-rank runs on one box -- no scaling curve was measured"}
for k, t1n, sw in ((4, "t1_512", "slab_sweep4"), (4, "t1_512", "slab_sweep4_round3_schedule"), (8, "t1_1024", "slab_sweep8")):
    t1, s = L(t1n), L(sw)
    ranks = s["ranks"]
    worst = max(r["ms_per_step"] for r in ranks)
    res[sw] = {"k": k, "t1_ms_per_frame": t1["ms_per_step"], "t1_integrate_ms": t1["roofline"]["avg_launch_ms"],
               "t1_roofline_frac": t1["roofline"]["frac"], "schedule": ranks[0].get("schedule"),
               "rank_ms_per_frame": [r["ms_per_step"] for r in ranks], "rank_integrate_ms": [r["integrate_ms"] for r in ranks],
               "rank_roofline_frac": [r["roofline_frac"] for r in ranks],
               "slowest_rank_ms": worst, "projected_efficiency": round(t1["ms_per_step"] / (k * worst), 4),
               "projected_efficiency_mean_rank": round(t1["ms_per_step"] / (k * sum(r["ms_per_step"] for r in ranks) / len(ranks)), 4),
               "sweep_only_efficiency": round(t1["roofline"]["avg_launch_ms"] / (k * max(r["integrate_ms"] for r in ranks)), 4)}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
