#!/bin/bash
# Per-rank single-GPU runs of BASELINE configs[3] (8 sensors, 512^3 / 4) and configs[4] (8 sensors, 1024^3 / 8), and the
# rocprofv3 evidence for the 8-sensor sweep kernel with staging.  Run on the GPU box from the repo root:
#   bash profiles/collect_slabs.sh <tag>
# Kernel timing (--kernel-trace --stats) and every PMC group are separate runs; the program follows `--` directly.
TAG=${1:-r03}
OUT=$PWD/gpurun_out/slabs_$TAG
rm -rf $OUT && mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --slab-sweep 4 --steps 30 --warmup 5 > $OUT/slab_sweep4.json 2> $OUT/slab_sweep4.err
python3 $ROOT/bench.py --slab-sweep 8 --steps 30 --warmup 5 > $OUT/slab_sweep8.json 2> $OUT/slab_sweep8.err
SLAB="python3 $ROOT/bench.py --slab 1/4 --steps 30 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $SLAB > $OUT/slab_1of4_under_rocprofv3.json 2>/dev/null
SHORT="python3 $ROOT/bench.py --slab 1/4 --steps 5 --warmup 2"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq1 -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $OUT/write -- $SHORT > /dev/null 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_integrate' in k:
            res[k][r['Counter_Name']].append(float(r['Counter_Value']))
summary = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
for f in glob.glob(out + '/stats/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if r['Name'] in summary:
            summary[r['Name']]['avg_ns'] = float(r['AverageNs'])
            summary[r['Name']]['calls'] = int(r['Calls'])
for k, v in summary.items():
    if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        v['hbm_bytes_per_launch'] = int(round(v['FETCH_SIZE'] * 2048 + v['WRITE_SIZE'] * 1024))   # FETCH_SIZE x2 on gfx950
json.dump(summary, open(out + '/pmc_summary.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
PY
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
