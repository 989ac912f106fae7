#!/bin/bash
# developer probe: kernel time + SQ counters of k_depth_peels, as shipped and with every super-cell marked "near"
# (RGBDR_PEEL_ALLNEAR=1: the walk then takes the full iteration everywhere, like the round-4 kernel)
OUT=$PWD/gpurun_out/pmc_peels; rm -rf $OUT; mkdir -p $OUT; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for mode in shipped allnear; do
  [ $mode = allnear ] && export RGBDR_PEEL_ALLNEAR=1 || unset RGBDR_PEEL_ALLNEAR
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${mode}_stats -- python3 $ROOT/profiles/peels_probe.py > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/${mode}_sq -- python3 $ROOT/profiles/peels_probe.py > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${mode}_sq2 -- python3 $ROOT/profiles/peels_probe.py > /dev/null 2>&1
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for mode in ("shipped", "allnear"):
    res = collections.defaultdict(list)
    for f in glob.glob(out + "/%s_sq*/*/*counter_collection.csv" % mode):
        for r in csv.DictReader(open(f)):
            if "k_depth_peels" in r["Kernel_Name"]:
                res[r["Counter_Name"]].append(float(r["Counter_Value"]))
    t = None
    for f in glob.glob(out + "/%s_stats/*/*kernel_stats.csv" % mode):
        for r in csv.DictReader(open(f)):
            if "k_depth_peels" in r["Name"]:
                t = float(r["AverageNs"])
    print(mode, "avg_ns", t, {k: round(sum(v) / len(v)) for k, v in sorted(res.items())})
PY
