#!/bin/bash
# PMC counters of the pre_* kernels for one variant of the 13x13 passes: bash profiles/pmc_pre.sh <variant> <tag>
V=${1:-2}; TAG=${2:-r02}
OUT=$PWD/gpurun_out/pmc_pre_${TAG}_v$V
mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
# ($V only tags the output directory; set library knobs such as RGBDR_SEPARATE_PASSES in the environment)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/profiles/pre_only.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq1 -- python3 $ROOT/profiles/pre_only.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq2 -- python3 $ROOT/profiles/pre_only.py > /dev/null 2>&1
# lane utilisation: SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) -- both count quad-cycles per SIMD
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq3 -- python3 $ROOT/profiles/pre_only.py > /dev/null 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/sq*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'rgbdr::k_' in k:
            res[k][r['Counter_Name']].append(float(r['Counter_Value']))
summary = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
for k, v in summary.items():
    if v.get('SQ_THREAD_CYCLES_VALU') and v.get('SQ_ACTIVE_INST_VALU'):
        v['valu_lane_utilisation'] = v['SQ_THREAD_CYCLES_VALU'] / (64.0 * v['SQ_ACTIVE_INST_VALU'])
for f in glob.glob(out + '/stats/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Name'].split('(')[0]
        if n in summary:
            summary[n]['avg_ns'] = float(r['AverageNs']); summary[n]['calls'] = int(r['Calls'])
json.dump(summary, open(out + '/pmc_summary.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
PY
