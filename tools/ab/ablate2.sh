#!/bin/bash
# usage: tools/ablate2.sh <workload> "VAR1=a VAR2=b" "VAR1=c VAR2=d" ...
WL=$1; shift
for cfg in "$@"; do
  out=$(env $cfg python bench.py --workload $WL --no-cpu-baseline --steps 300 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_kernel_us'],1), 'us', round(d['value']/1e6,2), 'M/s')")
  echo "$cfg -> $out"
done
