#!/bin/bash
# SQ instruction-mix / stall counters for one bench workload (runs on the GPU box).
# Usage: tools/pmc_sq.sh <workload> <tag>   (CRL_*_DEBUG env vars are inherited)
set -u
WL=${1:-fused84}; TAG=${2:-sq_$WL}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -- python3 $REPO/bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/b -- python3 $REPO/bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
cd $REPO
python3 - $OUT <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for sub in "ab":
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "raster" not in k and "car_" not in k and "dynamics" not in k: continue
        print(k)
        for c, v in sorted(d.items()):
            print(f"   {c:28s} mean/launch {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
find $OUT -name '*.csv' -size +2M -delete
