#!/bin/bash
# usage (GPU box): bash tools/watch_clocks.sh <seconds> -- <command...>
# Runs the command and samples rocm-smi (sclk, mclk, power, temperature) every 0.5 s beside it: is the chip
# running at its power cap (low shader clock) while the decode kernels run?
secs=$1; shift; shift
"$@" &
pid=$!
for i in $(seq 1 $((secs * 2))); do
  sleep 0.5
  kill -0 $pid 2>/dev/null || break
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor junction\)" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/  */ /g'
  echo
done
wait $pid
