#!/bin/bash
# HIP's own call log (AMD_LOG_LEVEL=3) around the host stall of profiles/lag_stall_probe.py: the largest gaps between log lines
OUT=$PWD/gpurun_out/stall_log; rm -rf $OUT; mkdir -p $OUT
PLAIN_FIRST=own BARRIER_EVERY=50 SYNC_EVERY=0 MODE=sendrecv AMD_LOG_LEVEL=3 timeout 280 python profiles/lag_stall_probe.py > $OUT/out.txt 2> $OUT/hip.log
grep "pushes in" $OUT/out.txt $OUT/hip.log | tail -2
python3 - $OUT/hip.log $OUT/out.txt <<'PY'
import re, sys, collections
win = None
for ln in open(sys.argv[2]):
    if ln.startswith("STALL_WINDOW"):
        win = tuple(int(x) for x in ln.split()[1:3])
print("stall window (us):", win, "length", (win[1] - win[0]) if win else None)
calls, first, last = collections.Counter(), [], collections.deque(maxlen=12)
inwin = []
pat = re.compile(r":\s*(\d{9,})\s*us:\s*(.*)")
with open(sys.argv[1], errors="replace") as f:
    for ln in f:
        m = pat.search(ln)
        if not m or not win:
            continue
        t = int(m.group(1))
        if win[0] - 2000 <= t <= win[1] + 2000:
            body = re.sub(r"\x1b\[[0-9;]*m", "", m.group(2)).strip()
            name = body.split("(")[0].split(":")[0].strip()[:48]
            calls[name] += 1
            inwin.append((t - win[0], body[:200]))
            if len(first) < 60:
                first.append("%d  %s" % (t - win[0], body[:170]))
            last.append("%d  %s" % (t - win[0], body[:170]))
print("calls logged inside the window:", calls.most_common(12))
gaps = sorted(((inwin[i][0] - inwin[i - 1][0], i) for i in range(1, len(inwin))), reverse=True)[:3]
for g, i in gaps:
    print("==== silence of %d us inside the window; the lines before and after (us relative to the start of the stalled push):" % g)
    for k in range(max(0, i - 30), min(len(inwin), i + 12)):
        print("  %8d  %s" % inwin[k])
PY
rm -f $OUT/hip.log
