#!/bin/bash
# A/B of k_window_background builds (profiles/probes_src/librgbdr_hip_<variant>.so) under rocprofv3 --kernel-trace.
# Run on the GPU box from the repo root, the whole script under `timeout`.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  out=gpurun_out/wb_$v
  RGBDR_PROBE_LIB=profiles/probes_src/librgbdr_hip_$v.so timeout 150 rocprofv3 --kernel-trace --stats --output-format csv \
      -d $out -- python3 profiles/skip_probe.py < /dev/null 2>&1 | grep "^skip"
  for f in $(find $out -name "*kernel_stats.csv" < /dev/null); do
    echo "== $v"; grep -h "window_background\|skip_classify\|tiled_listed" "$f" < /dev/null | cut -c1-220
  done
done
