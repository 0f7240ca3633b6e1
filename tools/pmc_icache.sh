#!/bin/bash
# Instruction-cache behaviour of the CarRacing kernels (runs on the GPU box; PMC collection serialises the kernels).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_icache
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -i "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u > $OUT/avail.txt
cat $OUT/avail.txt | tr '\n' ' '; echo
CRL_BENCH_CAR_PREROLL=200 timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/a -- python3 $REPO/bench.py --workload car --steps 6 --warmup 2 --no-cpu-baseline > $OUT/a.json 2> $OUT/a.err < /dev/null
cd $REPO
python3 - $OUT <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "car_" not in k: continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    req, miss = m.get("SQC_ICACHE_REQ", 0), m.get("SQC_ICACHE_MISSES", 0)
    print(f"{k:48s} n={len(next(iter(d.values()))):4d} icache req {req:12.0f} miss {miss:10.0f} ({100*miss/max(req,1):5.1f} %)  valu {m.get('SQ_INSTS_VALU',0):12.0f} waves {m.get('SQ_WAVES',0):8.0f} wait_inst/wave_cycles {m.get('SQ_WAIT_INST_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1):.3f}")
PY
tail -3 $OUT/a.err | cut -c1-200
find $OUT -name '*.csv' -size +2M -delete
