#!/bin/bash
# Per-kernel durations of the hole filling of a 1280 x 720 frame (profiles/fill_probe.py under rocprofv3 --kernel-trace):
# prints the kernel stats of the k_fc_* kernels and the start / duration / gap of the last frame's launches.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/fillprof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fill -- python3 $ROOT/profiles/fill_probe.py > $OUT/probe.txt 2>&1
python3 - $OUT <<'PY'
import csv, sys
out = sys.argv[1]
for r in csv.DictReader(open(out + '/fill_kernel_stats.csv')):
    if 'fc_' in r['Name']:
        print('%-22s calls %4s avg %8.2f us  min %7.2f  max %7.2f' % (r['Name'].split('(')[0].replace('rgbdr::', ''), r['Calls'],
              float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
rows = sorted(csv.DictReader(open(out + '/fill_kernel_trace.csv')), key=lambda r: int(r['Start_Timestamp']))
fc = [r for r in rows if 'fc_' in r['Kernel_Name']]
n = next(k for k in range(1, len(fc)) if 'colorfill' in fc[k - 1]['Kernel_Name'])
seq, t0, prev = fc[-n:], int(fc[-n]['Start_Timestamp']), None
for r in seq:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('  %-20s grid %6s x %-4s start %7.2f us  dur %6.2f  gap %s' % (r['Kernel_Name'].split('(')[0].replace('rgbdr::', ''), r['Grid_Size_X'], r['Grid_Size_Y'],
          (s - t0) / 1e3, (e - s) / 1e3, '-' if prev is None else '%.2f' % ((s - prev) / 1e3)))
    prev = e
print('  last frame, first start to last end: %.2f us in %d launches' % ((prev - t0) / 1e3, n))
PY
