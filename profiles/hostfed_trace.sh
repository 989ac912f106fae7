#!/bin/bash
# Kernel + memory-copy trace of the host-fed frame loop in each schedule (no PMC in these runs):
#   bash profiles/hostfed_trace.sh <tag>   -> gpurun_out/hostfed_<tag>/{sequential,pipelined,device}/
TAG=${1:-r04}
OUT=$PWD/gpurun_out/hostfed_$TAG
rm -rf $OUT && mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for MODE in device mapped-sequential mapped-pipelined pageable-sequential; do
  python3 $ROOT/profiles/hostfed_trace.py $MODE 60 > $OUT/$MODE.plain.json 2>$OUT/$MODE.plain.err
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/$MODE -- python3 $ROOT/profiles/hostfed_trace.py $MODE 30 > $OUT/$MODE.traced.json 2>$OUT/$MODE.err
done
python3 $ROOT/profiles/hostfed_timeline.py $OUT > $OUT/timeline.txt 2>&1
