#!/bin/bash
# kernel timeline of a few CarRacing steps (start/end per kernel, relative us)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/car_timeline
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $REPO/bench.py --workload ${CAR_WORKLOAD:-car} --steps ${CAR_STEPS:-30} --warmup ${CAR_WARMUP:-5} --no-cpu-baseline > $OUT/b.json 2> $OUT/err
cd $REPO
python3 - $OUT <<'PY'
import sys, glob, csv
f = sorted(glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True), key=lambda p: __import__("os").path.getmtime(p))[-1]
rows = [r for r in csv.DictReader(open(f)) if "crl::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:10.1f} {(int(r["End_Timestamp"])-t0)/1e3:10.1f}  q{r.get("Queue_Id","?")} {r["Kernel_Name"][5:32]}')
PY
