#!/bin/bash
# HBM bytes of the sweep with and without RGBDR_FLAG_NO_RESAMPLE (profiles/generic_price_probe.py): separate --pmc passes
#   bash profiles/pmc_generic.sh <tag>  ->  gpurun_out/pmc_generic_<tag>/summary.json
TAG=${1:-r06}
OUT=$PWD/gpurun_out/pmc_generic_$TAG
rm -rf $OUT && mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
W="python3 $ROOT/profiles/generic_price_probe.py"
$W > $OUT/times.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $W > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch -- $W > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/write -- $W > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/tcc -- $W > /dev/null 2>&1
rm -f $OUT/*/*/*kernel_trace.csv $OUT/*/*/*_agent_info.csv
python3 - $OUT <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('rgbdr::', '').replace('void ', '')
        if 'k_integrate' in k:
            # the two operating points differ in grid size: keep them apart
            res['%s, %d tiles' % (k, int(r['Grid_Size']) // int(r['Workgroup_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
s = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
for k, v in s.items():
    if 'FETCH_SIZE' in v:
        v['hbm_bytes'] = (v['FETCH_SIZE'] * 2 + v.get('WRITE_SIZE', 0)) * 1024
json.dump({'kernels': s, 'times': open(out + '/times.txt').read().splitlines()}, open(out + '/summary.json', 'w'), indent=1, sort_keys=True)
print(open(out + '/times.txt').read())
for k, v in sorted(s.items()):
    print('%-60s HBM %.3f GB  VALU %.0f M  VMEM_RD %.1f M' % (k[:60], v.get('hbm_bytes', 0) / 1e9, v.get('SQ_INSTS_VALU', 0) / 1e6, v.get('SQ_INSTS_VMEM_RD', 0) / 1e6))
PY
