#!/bin/bash
# Samples shader clock and socket power from sysfs every 0.25 s for $1 samples (default 120).
n=${1:-120}
dev=$(ls -d /sys/class/drm/card*/device 2>/dev/null | head -n 1)
hw=$(ls -d $dev/hwmon/hwmon* 2>/dev/null | head -n 1)
echo "device $dev hwmon $hw"
for i in $(seq $n); do
  sclk=$(grep '\*' $dev/pp_dpm_sclk 2>/dev/null | tr '\n' ' ')
  pw=$(cat $hw/power1_average 2>/dev/null || cat $hw/power1_input 2>/dev/null)
  f=$(cat $hw/freq1_input 2>/dev/null)
  echo "$(date +%s.%N | cut -c1-13) sclk[$sclk] freq1_input=$f power_uW=$pw"
  sleep 0.25
done
