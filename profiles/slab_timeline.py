#!/usr/bin/env python3
"""Timeline of the last frames of a `rocprofv3 --kernel-trace [--memory-copy-trace]` run of bench.py --slab r/k:
    python3 profiles/slab_timeline.py <trace dir>
kernels and copies between consecutive staging sweeps, start / end in us relative to the first sweep shown."""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0].replace("rgbdr::", "")[:56], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
for f in glob.glob(d + "/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "copy")[:56], "", ""))
ev.sort()
want = sys.argv[2] if len(sys.argv) > 2 else "true>"     # the staging variant <N, 4, true, false, true>
sweeps = [i for i, e in enumerate(ev) if "k_integrate_tiled<" in e[3] and e[3].rstrip().endswith(want)]
if len(sweeps) < 8:
    sweeps = [i for i, e in enumerate(ev) if "k_integrate_tiled<" in e[3]]
# argv[3] = "last": four of the last sweeps (the timed steps of a run whose first dozens of steps try other schedules)
sel = sweeps[-7:-3] if (len(sys.argv) > 3 and sys.argv[3] == "last" and len(sweeps) > 8) else (sweeps[10:14] if len(sweeps) > 14 else sweeps[-4:])
t0 = ev[sel[0]][0]
for s, e, k, n, q, st in ev[sel[0] - 8:sel[-1] + 1]:
    print("%10.1f %10.1f %8.1f us  %s %-56s q%s s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, k, n, q, st))
