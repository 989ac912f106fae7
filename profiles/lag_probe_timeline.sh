#!/bin/bash
# kernel timeline of profiles/lag_probe.py with the halo exchange on (its last run): gpurun_out/lag_probe_timeline.txt
OUT=$PWD/gpurun_out/lag_probe_tl; rm -rf $OUT; mkdir -p $OUT; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
export EXCH=1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -- python3 $ROOT/profiles/lag_probe.py > $OUT/probe.log 2>&1
python3 $ROOT/profiles/slab_timeline.py $OUT/t "true>" last > $ROOT/gpurun_out/lag_probe_timeline.txt 2>&1
rm -rf $OUT/t
grep -a "sweep" $OUT/probe.log | tail -12
tail -30 $ROOT/gpurun_out/lag_probe_timeline.txt
