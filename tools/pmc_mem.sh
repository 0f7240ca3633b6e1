#!/bin/bash
# memory-pipeline stall counters for one bench workload (runs on the GPU box)
WL=${1:-fused84}; TAG=${2:-mem_$WL}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { timeout 240 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 $REPO/bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline > $OUT/$1.json 2> $OUT/$1.err; }
run a "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_DATA_FIFO_FULL SQ_WAVE_CYCLES"
run b "TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ"
run c "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_TAG_STALL"
# (derived *_avr counters hung rocprofv3 on this pool: not collected)
cd $REPO
python3 - $OUT <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for sub in "abc":
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "raster" not in k: continue
        for c, v in sorted(d.items()):
            print(f"{k[:40]:40s} {c:36s} {sum(v)/len(v):18.1f}")
PY
find $OUT -name '*.csv' -size +2M -delete
