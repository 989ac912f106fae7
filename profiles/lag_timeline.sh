#!/bin/bash
# Kernel + copy timeline of one slab rank (1 of 4 of BASELINE configs[3]) under the two sharded chain schedules:
#   bash profiles/lag_timeline.sh   -> gpurun_out/lag_timeline/{sharded,lagged}.txt (profiles/slab_timeline.py's table)
# plain sharded: chain -> gather (RCCL kernel) -> sweep; lagged (dist.LaggedChain): import | chain of the NEXT frame | sweep,
# the gather's RCCL kernel on another queue UNDER the sweep.
OUT=$PWD/gpurun_out/lag_timeline${TAG:-}; rm -rf $OUT; mkdir -p $OUT; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for c in ${CHAINS:-sharded lagged}; do
  export RGBDR_BENCH_CHAIN=$c
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/$c -- python3 $ROOT/bench.py --slab 1/4 --steps 30 --warmup 5 --no-legs > $OUT/$c.json 2>/dev/null
  python3 $ROOT/profiles/slab_timeline.py $OUT/$c "true>" last > $OUT/$c.txt 2>&1
  rm -rf $OUT/$c
done
for c in ${CHAINS:-sharded lagged}; do echo "---- $c"; tail -${LINES_EACH:-34} $OUT/$c.txt; done
