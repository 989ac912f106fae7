#!/bin/bash
# developer probe: which clock / power / busy files the GPU's sysfs directory offers on the box (bench_legs.py leg_box reads them)
for d in /sys/class/drm/card*/device; do
  [ -e $d/pp_dpm_mclk ] || continue
  echo "== $d"; ls $d | tr '\n' ' '; echo
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk gpu_busy_percent mem_busy_percent power_dpm_force_performance_level; do echo "-- $f"; cat $d/$f 2>&1 | head -12; done
  for h in $d/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo; for f in freq1_input freq1_label freq2_input freq2_label power1_average power1_input power1_label; do [ -e $h/$f ] && echo "$f: $(cat $h/$f 2>&1)"; done; done
done
