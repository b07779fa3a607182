#!/bin/bash
# Samples GPU clock / power with rocm-smi while a workload runs (is the chip clock- or power-limited in situ?).
# usage: tools/power_probe.sh <out.log> <command...>
out="$1"; shift
"$@" > "${out%.log}.run.log" 2>&1 &
pid=$!
: > "$out"
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | tr -s ' ' | tr '\n' '|' >> "$out"
  echo >> "$out"
  sleep 0.4
done
wait $pid
