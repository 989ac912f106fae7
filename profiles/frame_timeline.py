#!/usr/bin/env python3
"""Per-frame timeline of any traced frame loop (rocprofv3 --kernel-trace [--memory-copy-trace] --output-format csv -d <dir>):
    python3 profiles/frame_timeline.py <dir> <name of the frame's last kernel (substring)> [frames to print]
prints the kernels / copies of the last frames with start, end, duration and the idle gap in front of each, and the mean
busy and idle time per frame."""
import collections, csv, glob, sys
d, last = sys.argv[1], sys.argv[2]
nshow = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ev = []
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("rgbdr::", "").replace("void ", "")[:52]))
for f in glob.glob(d + "/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "copy")))
ev.sort()
ends = [i for i, e in enumerate(ev) if last in e[2]]
frames = [(ends[i - 1] + 1, ends[i]) for i in range(1, len(ends))]
frames = frames[len(frames) // 2:]                     # steady state
busy, idle, per = collections.defaultdict(float), 0.0, 0.0
for a, b in frames:
    per += ev[b][1] - ev[a - 1][1]
    prev = ev[a - 1][1]
    for s, e, n in ev[a:b + 1]:
        busy[n] += e - s
        idle += max(0, s - prev)
        prev = max(prev, e)
n = len(frames)
print("frames %d  period %.1f us  idle %.1f us per frame" % (n, per / n / 1e3, idle / n / 1e3))
for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print("   %-56s %7.1f us" % (k, v / n / 1e3))
for a, b in frames[-nshow:]:
    t0 = ev[a - 1][1]
    prev = t0
    for s, e, nme in ev[a:b + 1]:
        print("%9.1f %9.1f %7.1f us  gap %5.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, nme))
        prev = max(prev, e)
    print()
