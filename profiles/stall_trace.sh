#!/bin/bash
# The once-per-process host stall of the N > 1 paths: one slab rank of configs[3] for 400 steps under
# rocprofv3 --hip-trace --kernel-trace with NCCL_DEBUG=INFO; prints the longest HIP API calls of the host, what surrounds the
# longest one, and what RCCL logged around that time.   bash profiles/stall_trace.sh  ->  gpurun_out/stall_trace/
OUT=$PWD/gpurun_out/stall_trace; rm -rf $OUT; mkdir -p $OUT; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
# (the stall was first seen in a process that FOLLOWS another one on the box: an untraced run first)
python3 $ROOT/bench.py --slab 1/4 --steps 400 --warmup 5 --no-legs > $OUT/line_first.json 2> /dev/null
export NCCL_DEBUG=INFO NCCL_DEBUG_FILE=$OUT/rccl.%p.log NCCL_DEBUG_SUBSYS=INIT,COLL,P2P,ALLOC,PROXY
rocprofv3 --hip-trace --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --slab 1/4 --steps 400 --warmup 5 --no-legs > $OUT/line.json 2> $OUT/err.txt
python3 $ROOT/profiles/stall_trace_analyze.py $OUT
python3 - $OUT <<'PY'
import json, sys
for f in ('line_first.json', 'line.json'):
    j = json.loads(open(sys.argv[1] + '/' + f).read().strip().splitlines()[-1])
    print(f, 'ms_per_step', j.get('ms_per_step'), 'retimed', j.get('headline_retimed'), 'first measurement', j.get('ms_per_step_first_measurement'))
PY
rm -rf $OUT/trace
