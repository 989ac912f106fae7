#!/bin/bash
# The kernels of one displayed frame (profiles/display_frame_only.py under rocprofv3 --kernel-trace): start, duration, gap to the
# kernel before it and the queue it ran on -- RGBDR_DISPLAY_PIPELINE=1 shows what overlaps what on a pipelined context.
#   bash profiles/frame_launches.sh ref|512
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/frameprof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RGBDR_DISPLAY_GRID=${1:-ref}
rocprofv3 --kernel-trace --output-format csv -d $OUT -o fr -- python3 $ROOT/profiles/display_frame_only.py > $OUT/probe.txt 2>&1
python3 - $OUT <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1] + '/fr_kernel_trace.csv')), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0].replace('rgbdr::', '').replace('void ', '') for r in rows]
ends = [i for i, n in enumerate(names) if 'colorfill' in n]
a, b = ends[-3] + 1, ends[-1] + 1        # the last two frames
t0 = int(rows[a]['Start_Timestamp']); prev = None
for r, n in zip(rows[a:b], names[a:b]):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('  %-40s queue %-3s start %7.1f us  dur %6.1f  end %7.1f  gap %s' % (n[:40], r.get('Queue_Id', '?'), (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3,
                                                                              '-' if prev is None else '%.1f' % ((s - prev) / 1e3)))
    prev = e
print('  two frames: %d launches, first start to last end %.1f us' % (b - a, (prev - t0) / 1e3))
PY
