#!/bin/bash
# Samples the GPU's clock and power twice a second while a command runs:  tools/clock_watch.sh out.txt -- <command>
# (is the pipelined schedule's convolution slow-down a clock / power-cap effect?  see DESIGN 4.3)
out=$1; shift; shift
( while true; do
    printf "%s " "$(date +%s.%N)" >> "$out"
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | sed -E 's/GPU\[0\]\s*:\s*//' | tr '\n' ';' >> "$out"
    echo >> "$out"
    sleep 0.5
  done ) &
watcher=$!
"$@"
rc=$?
kill $watcher
exit $rc
