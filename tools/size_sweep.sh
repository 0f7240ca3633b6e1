#!/bin/bash
# env-steps/s vs envs per GPU (runs on the GPU box)
for wl in raw fused84 car; do
  for n in 256 1024 4096 16384 65536; do
    if [ $wl = car ] && [ $n -gt 16384 ]; then continue; fi
    python bench.py --workload $wl --envs $n --no-cpu-baseline --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', $n, round(d['value']/1e6,3), 'M/s', round(d['ms_per_step']*1e3,1), 'us/step')"
  done
done
