#!/bin/bash
# average duration of every car_* kernel under the timing ablations of car_sensor_kernel (CRL_CAR_SENSOR_SERIAL bits 2 / 4 / 8)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/car_sensor_ablate
mkdir -p $OUT
for v in ${SENSOR_SETTINGS:-0 2 6 8 14}; do
  export CRL_CAR_SENSOR_SERIAL=$v
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $OUT/t$v -- python3 $REPO/bench.py --workload car --steps 30 --warmup 5 --no-cpu-baseline > $OUT/b$v.json 2> $OUT/err$v)
  python3 - $OUT/t$v $v <<'PY'
import sys, glob, csv, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "crl::car_" in r["Kernel_Name"]:
        d[r["Kernel_Name"][5:22]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("setting", sys.argv[2], {k: round(sum(v[len(v)//2:]) / len(v[len(v)//2:]), 1) for k, v in d.items()})
PY
done
