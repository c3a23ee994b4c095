#!/bin/bash
# Samples package power / shader clock / junction temperature (rocm-smi) twice a second while a workload runs, and
# prints the samples taken while the GPU was busy.   usage: tools/power_probe.sh [one_degree|nano]
which=${1:-one_degree}
python tools/class_times_by_feature_mode.py $which > /dev/null 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Current Socket|sclk|Sensor junction" | sed 's/GPU\[0\]\t*: //' | tr '\n' ' '; echo
  sleep 0.5
done | grep -E "\(([5-9][0-9]{2}|[1-9][0-9]{3})Mhz\)" | tail -40
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -1
