#!/bin/bash
# usage: tools/ablate.sh <workload> <ENVVAR> <values...>  -- prints avg raster kernel us per debug value
WL=$1; VAR=$2; shift 2
for d in "$@"; do
  export $VAR=$d
  python bench.py --workload $WL --no-cpu-baseline --steps 300 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$d', round(d['roofline']['avg_kernel_us'],1), 'us', round(d['value']/1e6,2), 'M/s')"
done
