#!/bin/bash
# profiles/probes_src/valu_issue_probe.hip alone and under rocprofv3 --pmc: what SQ_ACTIVE_INST_VALU counts per instruction
OUT=$PWD/gpurun_out/valu_probe; rm -rf $OUT; mkdir -p $OUT; ROOT=$PWD
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue_probe $ROOT/profiles/probes_src/valu_issue_probe.hip 2>/dev/null
/tmp/valu_issue_probe | tee $OUT/plain.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc -- /tmp/valu_issue_probe > /dev/null 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
rows = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        rows.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"][:40], r["Grid_Size"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, k, g), c in sorted(rows.items()):
    print(d, k, "grid", g, {n: round(v) for n, v in c.items()},
          "ACTIVE_INST_VALU per instruction %.2f quad-cycles" % (c["SQ_ACTIVE_INST_VALU"] / max(c["SQ_INSTS_VALU"], 1)),
          "| per SIMD: active quads x 4 / GUI cycles per XCD = %.2f" % (c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["GRBM_GUI_ACTIVE"] / 8)))
PY
rm -rf $OUT/pmc
