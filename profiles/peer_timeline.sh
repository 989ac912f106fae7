#!/bin/bash
# Kernel + copy timeline of one slab rank (1 of 4 of BASELINE configs[3]) with the halo moved by RCCL and by the copy engine
# (bench.py --halo-transport peer; on one GPU the rank is its own neighbour):
#   bash profiles/peer_timeline.sh   -> gpurun_out/peer_timeline/{rccl,peer}.txt (profiles/slab_timeline.py's table)
OUT=$PWD/gpurun_out/peer_timeline; rm -rf $OUT; mkdir -p $OUT; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
export RGBDR_BENCH_CHAIN=sharded
for t in rccl peer; do
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/$t -- python3 $ROOT/bench.py --slab 1/4 --halo-transport $t --steps 30 --warmup 5 --no-legs > $OUT/$t.json 2>/dev/null
  python3 $ROOT/profiles/slab_timeline.py $OUT/$t "true>" last > $OUT/$t.txt 2>&1
  rm -rf $OUT/$t
done
for t in rccl peer; do echo "---- $t"; tail -${LINES_EACH:-30} $OUT/$t.txt; done
