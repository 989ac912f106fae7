#!/bin/bash
# A/B of builds of the library on ONE box in ONE session (boxes differ by several per cent, sessions on one box by a few):
# the displayed frame of profiles/display_frame_only.py with each of the given libraries in turn, ROUNDS times over.
#   bash profiles/ab_display.sh profiles/probes_src/lib_a.so profiles/probes_src/lib_b.so ...      (GRIDS="ref 512", ROUNDS=3)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for g in ${GRIDS:-ref 512}; do
  for r in $(seq ${ROUNDS:-3}); do
    for lib in "$@"; do
      printf "%-28s " $(basename $lib)
      RGBDR_PROBE_LIB=$ROOT/$lib RGBDR_DISPLAY_GRID=$g python3 $ROOT/profiles/display_frame_only.py | sed 's/ per displayed frame.*raymarch/ raymarch/; s/, .holefill.*//'
    done
  done
done
