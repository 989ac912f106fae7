#!/bin/bash
# Counters of k_raymarch on the benchmark volume at 1280 x 720 (peels_probe.py: with and without space skipping): bash profiles/pmc_raymarch.sh <tag>
TAG=${1:-r04}
OUT=$PWD/gpurun_out/pmc_raymarch_skip_$TAG
rm -rf $OUT && mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/profiles/peels_probe.py > $OUT/probe.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq1 -- python3 $ROOT/profiles/peels_probe.py > /dev/null 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum --output-format csv -d $OUT/tcp -- python3 $ROOT/profiles/peels_probe.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch -- python3 $ROOT/profiles/peels_probe.py > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 $ROOT/profiles/peels_probe.py > /dev/null 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'k_raymarch' in k or 'k_depth_peels' in k:
            res[k][r['Counter_Name']].append(float(r['Counter_Value']))
summary = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
for f in glob.glob(out + '/stats/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Name'].split('(')[0]
        if n in summary:
            summary[n]['avg_ns'] = float(r['AverageNs']); summary[n]['calls'] = int(r['Calls'])
json.dump(summary, open(out + '/pmc_summary.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
print(open(out + '/probe.txt').read()[-600:])
PY
