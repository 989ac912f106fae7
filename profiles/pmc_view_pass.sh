#!/bin/bash
# The consumers of the reference's default displayed frame -- k_depth_peels (+ k_peel_near), k_raymarch, k_fc_* -- on the
# reference's box and on the 512^3 grid: time, and what bounds each (VALU issue, vector-L1 accesses, HBM bytes, launches).
#   bash profiles/pmc_view_pass.sh <tag>   ->  gpurun_out/pmc_view_<tag>/summary.json   (copied to profiles/<tag>_pmc_view_pass.json)
TAG=${1:-r06}
OUT=$PWD/gpurun_out/pmc_view_$TAG
rm -rf $OUT && mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
W="python3 $ROOT/profiles/display_frame_only.py"
for G in ref 512; do
  export RGBDR_DISPLAY_GRID=$G
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$G/stats -- $W > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/$G/sq1 -- $W > /dev/null 2>&1
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/$G/tcp -- $W > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$G/fetch -- $W > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/$G/write -- $W > /dev/null 2>&1
  rm -f $OUT/$G/*/*/*kernel_trace.csv $OUT/$G/*/*/*_agent_info.csv
done
python3 - $OUT <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
CUS, SIMDS, XCDS = 256, 1024, 8
ISSUE = 2.6       # cycles per wavefront VALU instruction a SIMD sustains at 4-8 wavefronts (profiles/r05_valu_issue_probe.txt)
summary = {"bounds": {"valu_issue": "SQ_INSTS_VALU x %.1f cycles / (1024 SIMDs x kernel cycles): %.1f = what a SIMD sustains, "
                                    "profiles/r05_valu_issue_probe.txt" % (ISSUE, ISSUE),
                      "l1_accesses": "TCP_TOTAL_CACHE_ACCESSES / (256 CUs x kernel cycles): one access per CU and cycle",
                      "hbm_bytes": "(FETCH_SIZE x 2 + WRITE_SIZE) KB / time against 8 TB/s (the guide's gfx950 correction)",
                      "kernel cycles": "GRBM_GUI_ACTIVE / 8 XCDs"}}
for G in ('ref', '512'):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('%s/%s/*/*/*counter_collection.csv' % (out, G)):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('rgbdr::', '').replace('void ', '')
            if any(s in k for s in ('k_raymarch', 'k_depth_peels', 'k_peel_near', 'k_fc_', 'k_decode_dxt')):
                res[k][r['Counter_Name']].append(float(r['Counter_Value']))
    s = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
    for f in glob.glob('%s/%s/stats/*/*kernel_stats.csv' % (out, G)):
        for r in csv.DictReader(open(f)):
            n = r['Name'].split('(')[0].replace('rgbdr::', '').replace('void ', '')
            if n in s:
                s[n]['avg_ns'], s[n]['calls'], s[n]['launches_per_frame'] = float(r['AverageNs']), int(r['Calls']), int(r['Calls']) / 30.0
    for k, v in s.items():
        cyc = v.get('GRBM_GUI_ACTIVE', 0) / XCDS
        if cyc > 0:
            v['kernel_cycles'] = cyc
            v['frac_valu_issue'] = round(v.get('SQ_INSTS_VALU', 0) * ISSUE / (SIMDS * cyc), 3)
            v['frac_l1_accesses'] = round(v.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0) / (CUS * cyc), 3)
        if 'avg_ns' in v:
            b = (v.get('FETCH_SIZE', 0) * 2 + v.get('WRITE_SIZE', 0)) * 1024
            v['hbm_bytes'] = b
            v['frac_hbm_bytes'] = round(b / (v['avg_ns'] * 1e-9) / 8e12, 3)
            v['ms_per_frame'] = round(v['avg_ns'] * v['launches_per_frame'] * 1e-6, 4)
        fr = {n: v.get(n, 0) for n in ('frac_valu_issue', 'frac_l1_accesses', 'frac_hbm_bytes')}
        v['bound'] = max(fr, key=fr.get)
        v['frac_of_bound'] = fr[v['bound']]
    summary[G] = s
json.dump(summary, open(out + '/summary.json', 'w'), indent=1, sort_keys=True)
for G in ('ref', '512'):
    for k, v in sorted(summary[G].items()):
        print('%-4s %-28s %6.1f us x %4.1f/frame  valu %.2f  l1 %.2f  hbm %.2f  -> %s' % (
            G, k, v.get('avg_ns', 0) / 1e3, v.get('launches_per_frame', 0), v.get('frac_valu_issue', 0), v.get('frac_l1_accesses', 0),
            v.get('frac_hbm_bytes', 0), v.get('bound')))
PY
