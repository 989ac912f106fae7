#!/bin/bash
# Kernel durations of the ray-march of the default displayed frame (profiles/display_frame_only.py under rocprofv3
# --kernel-trace): the whole march, and -- through the slab entry points -- the march alone (mode 1) and the shading alone (mode 2).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export RGBDR_DISPLAY_SPLIT=1
for g in ${GRIDS:-ref 512}; do
OUT=$ROOT/gpurun_out/rmprof_$g; rm -rf $OUT; mkdir -p $OUT
export RGBDR_DISPLAY_GRID=$g
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o rm -- python3 $ROOT/profiles/display_frame_only.py > $OUT/probe.txt 2>&1
echo grid $g
python3 - $OUT <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1] + '/rm_kernel_stats.csv')):
    if 'raymarch' in r['Name'] or 'empty' in r['Name'] or 'peel' in r['Name']:
        print('%-40s calls %4s avg %8.2f us  min %7.2f  max %7.2f' % (r['Name'].split('(')[0].replace('rgbdr::', '')[:40], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
done
