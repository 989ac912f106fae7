#!/bin/bash
# Is the headline kernel data dependent?  PMC counters of k_integrate_tiled<4> on the ring scene (35 % valid pixels) and on
# the dense scene (every pixel valid and inside the box), separate --pmc passes per the guide:
#   bash profiles/pmc_integrate_scenes.sh <tag>   ->  gpurun_out/pmc_scenes_<tag>/summary.json
TAG=${1:-r06}
OUT=$PWD/gpurun_out/pmc_scenes_$TAG
rm -rf $OUT && mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for L in ring dense; do
  export RGBDR_PRE_LAYOUT=$L
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$L/stats -- python3 $ROOT/profiles/integrate_only.py > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/$L/sq1 -- python3 $ROOT/profiles/integrate_only.py > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/$L/sq2 -- python3 $ROOT/profiles/integrate_only.py > /dev/null 2>&1
  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH --output-format csv -d $OUT/$L/sq3 -- python3 $ROOT/profiles/integrate_only.py > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/$L/fetch -- python3 $ROOT/profiles/integrate_only.py > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/$L/tcc -- python3 $ROOT/profiles/integrate_only.py > /dev/null 2>&1
  rm -f $OUT/$L/*/*/*kernel_trace.csv $OUT/$L/*/*/*_agent_info.csv
done
python3 - $OUT <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
summary = {}
for L in ('ring', 'dense'):
    res = collections.defaultdict(list)
    for f in glob.glob('%s/%s/*/*/*counter_collection.csv' % (out, L)):
        for r in csv.DictReader(open(f)):
            if 'k_integrate_tiled<' in r['Kernel_Name']:
                res[r['Counter_Name']].append(float(r['Counter_Value']))
    s = {c: sum(v) / len(v) for c, v in res.items()}
    for f in glob.glob('%s/%s/stats/*/*kernel_stats.csv' % (out, L)):
        for r in csv.DictReader(open(f)):
            if 'k_integrate_tiled<' in r['Name']:
                s['avg_ns'], s['calls'] = float(r['AverageNs']), int(r['Calls'])
    if 'FETCH_SIZE' in s:
        s['hbm_read_bytes'] = s['FETCH_SIZE'] * 1024 * 2   # KB, and the guide's gfx950 correction (x 2)
    summary[L] = s
summary['dense_over_ring'] = {c: round(summary['dense'][c] / summary['ring'][c], 4) for c in summary['ring']
                              if c in summary['dense'] and summary['ring'][c]}
json.dump(summary, open(out + '/summary.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
PY
