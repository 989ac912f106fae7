#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline block refers to.  Run on the
# GPU box from the repo root:  bash profiles/collect_pmc.sh <tag>
# Kernel timing (--kernel-trace --stats) and every PMC group are separate runs
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE takes 3 of 4 TCC slots,
# WRITE_SIZE 2, SQ has 8).
TAG=${1:-r01}
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT   # a tag is one run: never mix counter files of two runs
ROOT=$PWD
export RGBDR_BENCH_EXTRA=$OUT/bench_extra_of_the_profiled_runs.json   # (what each run's line moved out of itself; not the round's record)
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
# the timing pass uses the default step count so that the bench's own average (HIP events over the
# timed launches) and the profiler's average (all launches of the run) are both well populated
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_stats.json 2>/dev/null
# ... and the headline alone (--no-legs): every launch of the integrate kernel in this pass belongs to the headline's context and
# schedule, so the profiler's average is comparable with the bench's own (the legs also run the kernel on a second context whose
# arena is not probed, and under the pipelined schedule)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -- python3 $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-legs > $OUT/bench_stats_headline.json 2>/dev/null
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq1 -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $OUT/write -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- $BENCH > /dev/null 2>&1
rm -f $OUT/*/*/*kernel_trace.csv $OUT/*/*/*_agent_info.csv   # (tens of MiB of per-dispatch rows nobody reads; gpurun_out is capped at 64 MiB)
# ... and the counter rows of the displayed-frame leg's kernels (hundreds of frames x 16 launches; profiles/pmc_view_pass.sh is their pass)
for f in $OUT/*/*/*counter_collection.csv; do
  grep -v -e "k_fc_" -e "k_raymarch" -e "k_depth_peels" -e "k_peel_near" -e "k_empty_tiles" -e "k_upload_morph" "$f" > "$f.keep" && mv "$f.keep" "$f"
done
python3 - $OUT <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_integrate' in k or 'k_pre_depth' in k or 'k_quality' in k or 'k_normal' in k:
            res[k][r['Counter_Name']].append(float(r['Counter_Value']))
summary = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
for f in glob.glob(out + '/stats/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if r['Name'] in summary:
            summary[r['Name']]['avg_ns'] = float(r['AverageNs'])
            summary[r['Name']]['calls'] = int(r['Calls'])
json.dump(summary, open(out + '/pmc_summary.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
PY
