#!/bin/bash
# (the switches used here exist only in the profiling variant: python -m competitive_rl_amd.build --variant abl -DCRL_ABLATION)
export CRL_LIB_VARIANT=abl
# VALU / SALU / LDS wave-instructions of car_raster_kernel per (env, viewer) tile under the CRL_CAR_DEBUG ablations
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/car_raster_valu
mkdir -p $OUT
for v in ${RASTER_SETTINGS:-0 2 32 16 8 5}; do
  export CRL_CAR_DEBUG=$v
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/t$v -- python3 $REPO/bench.py --workload car --steps 6 --warmup 2 --no-cpu-baseline > $OUT/b$v.json 2> $OUT/err$v)
  python3 - $OUT/t$v $v <<'PY'
import sys, glob, csv, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "car_raster_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
tiles = tot["SQ_WAVES"] / 4
print("CRL_CAR_DEBUG", sys.argv[2], "per tile:", {k: round(v / tiles, 1) for k, v in tot.items() if k != "SQ_WAVES"}, "tiles", int(tiles))
PY
  find $OUT/t$v -name '*.csv' -size +2M -delete
done
