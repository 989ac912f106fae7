#!/usr/bin/env python3
"""Longest host-side HIP calls and gaps INSIDE the step loops of a bench run traced by profiles/stall_trace.sh (after the tenth
sweep launch; launches of a kernel never launched before are listed apart: their time is the lazy load of its code object)."""
import csv
import glob
import os
import sys

out = sys.argv[1]
api = sorted(glob.glob(out + '/trace/*/*hip_api_trace.csv'), key=os.path.getsize)[-1]
kt = sorted(glob.glob(out + '/trace/*/*kernel_trace.csv'), key=os.path.getsize)[-1]
kern = {r['Correlation_Id']: r['Kernel_Name'].split('(')[0][-50:] for r in csv.DictReader(open(kt))}
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Function'], r['Thread_Id'], r['Correlation_Id']) for r in csv.DictReader(open(api)))
main = max(set(r[3] for r in rows), key=lambda t: sum(1 for r in rows if r[3] == t))
mine = [r for r in rows if r[3] == main]
sweeps = [k for k, r in enumerate(mine) if 'k_integrate_tiled' in kern.get(r[4], '')]
start = sweeps[10] if len(sweeps) > 10 else 0
seen, first_use, steady = set(), [], []
for k, r in enumerate(mine):
    name = kern.get(r[4])
    new = name is not None and name not in seen
    if name:
        seen.add(name)
    if k < start:
        continue
    (first_use if new else steady).append((r[1] - r[0], k, r))
t0 = mine[start][0]
print('%d sweeps traced; after the 10th:' % len(sweeps))
print(' first launches of a kernel (lazy code-object load):')
for d, k, r in sorted(first_use, reverse=True)[:6]:
    print('   %8.3f ms  %-22s %s  (sweep #%d)' % (d / 1e6, r[2], kern.get(r[4], ''), sum(1 for s in sweeps if s < k)))
print(' longest other calls:')
for d, k, r in sorted(steady, reverse=True)[:8]:
    print('   %8.3f ms  %-22s %s  at t = %.1f ms (sweep #%d)' % (d / 1e6, r[2], kern.get(r[4], ''), (r[0] - t0) / 1e6, sum(1 for s in sweeps if s < k)))
gaps = sorted(((mine[k + 1][0] - mine[k][1], k) for k in range(start, len(mine) - 1)), reverse=True)[:6]
print(' longest gaps between calls (host outside HIP):')
for g, k in gaps:
    print('   %8.3f ms  after %-24s before %-24s (sweep #%d)' % (g / 1e6, mine[k][2], mine[k + 1][2], sum(1 for s in sweeps if s < k)))
