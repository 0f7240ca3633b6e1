#!/bin/bash
# samples sclk / power with rocm-smi while a command runs: tools/clock_watch.sh <cmd...>
"$@" > /tmp/cw_out.txt 2>/dev/null &
PID=$!
sleep ${CW_DELAY:-25}
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|Power \(W\)|Average Graphics|Current Socket" | tr '\n' ' '; echo; sleep 0.3; done
wait $PID
tail -2 /tmp/cw_out.txt
