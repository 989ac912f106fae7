#!/usr/bin/env python3
"""Reads the traces of profiles/hostfed_trace.sh and prints, per schedule, the last frames as a timeline (start / end in us
relative to the first event shown; kernels and memory copies interleaved) plus per-kind totals per frame."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]
for mode in sorted(os.listdir(out)):
    d = os.path.join(out, mode)
    if not os.path.isdir(d):
        continue
    ev = []
    for f in glob.glob(d + "/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0].replace("rgbdr::", "")[:44], r.get("Queue_Id", "")))
    for f in glob.glob(d + "/*/*memory_copy_trace.csv"):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Kind", "copy"))[:44], ""))
    ev.sort()
    if not ev:
        continue
    print("=====", mode, open(os.path.join(out, mode + ".plain.json")).read().strip() if os.path.exists(os.path.join(out, mode + ".plain.json")) else "")
    sweeps = [i for i, e in enumerate(ev) if "k_integrate" in e[3]]
    lo = sweeps[-4] if len(sweeps) >= 4 else 0
    t0 = ev[lo][0]
    for s, e, k, n, q in ev[lo - 6 if lo >= 6 else 0:]:
        print("%10.1f %10.1f %8.1f us  %s %-44s q%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, k, n, q))
    if len(sweeps) > 6:
        per = (ev[sweeps[-1]][0] - ev[sweeps[5]][0]) / (len(sweeps) - 6) / 1e3
        tot = collections.defaultdict(float)
        for s, e, k, n, q in ev[sweeps[5]:sweeps[-1]]:
            tot[k + " " + n] += (e - s) / 1e3 / (len(sweeps) - 6)
        print("period %.1f us per frame; busy per frame: %s" % (per, json.dumps({k: round(v, 1) for k, v in sorted(tot.items())})))
