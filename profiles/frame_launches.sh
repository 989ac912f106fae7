ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/frameprof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RGBDR_DISPLAY_GRID=${1:-ref}
rocprofv3 --kernel-trace --output-format csv -d $OUT -o fr -- python3 $ROOT/profiles/display_frame_only.py > $OUT/probe.txt 2>&1
python3 - $OUT <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1] + '/fr_kernel_trace.csv')), key=lambda r: int(r['Start_Timestamp']))
# last frame: from the last k_upload* / first kernel after the last colorfill but one
names = [r['Kernel_Name'].split('(')[0].replace('rgbdr::', '').replace('void ', '') for r in rows]
ends = [i for i, n in enumerate(names) if 'colorfill' in n]
a, b = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[a]['Start_Timestamp']); prev = None; busy = 0
for r, n in zip(rows[a:b], names[a:b]):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('  %-44s start %7.1f us  dur %6.1f  gap %s' % (n[:44], (s - t0) / 1e3, (e - s) / 1e3, '-' if prev is None else '%.1f' % ((s - prev) / 1e3)))
    busy += e - s; prev = e
print('  frame: %d launches, first start to last end %.1f us, kernels busy %.1f us, gaps %.1f us' % (b - a, (prev - t0) / 1e3, busy / 1e3, (prev - t0 - busy) / 1e3))
PY
