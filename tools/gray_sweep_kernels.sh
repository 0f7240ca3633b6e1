#!/bin/bash
# (the switches used here exist only in the profiling variant: python -m competitive_rl_amd.build --variant abl -DCRL_ABLATION)
export CRL_LIB_VARIANT=abl
# per-kernel average duration of the fused-84 step with the address-linear writer (CRL_GRAY_SWEEP=1) under its debug bits
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/gray_sweep_kernels
mkdir -p $OUT
export CRL_GRAY_SWEEP=${CRL_GRAY_SWEEP:-1}
for v in ${SWEEP_SETTINGS:-0 1 5}; do
  export CRL_GRAY_SWEEP_DEBUG=$v
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $OUT/t$v -- python3 $REPO/bench.py --workload fused84 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/b$v.json 2> $OUT/err$v)
  python3 - $OUT/t$v $v <<'PY'
import sys, glob, csv, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "crl::" in r["Kernel_Name"]:
        d[r["Kernel_Name"].split("(")[0][-40:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("sweep debug", sys.argv[2], {k: round(sum(v[len(v)//2:]) / len(v[len(v)//2:]), 1) for k, v in d.items()})
PY
done
