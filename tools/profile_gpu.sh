#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for
# one bench workload.  Usage: tools/profile_gpu.sh <workload> <tag>
# Outputs under gpurun_out/prof_<tag>/ ; summaries are later copied into profiles/ (tools/collect_profiles.py).
# rocprofv3 is given the program itself (python3 bench.py ...), never a shell wrapper; --pmc passes are separate
# from the kernel-trace/stats pass and use --kernel-trace only.
set -u
WL=${1:-raw}; TAG=${2:-r03_$WL}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT  # one run, one set of files: nothing of an earlier run may be picked up
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --workload $WL --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
# CarRacing: the two traffic passes run the step on ONE stream (CRL_CAR_NO_OVERLAP=1), i.e. with one raster launch over all envs, so that
# "bytes per launch" is the full frame batch (the pipelined step draws three env classes in separate launches)
# (and from reset, without the 1000 un-timed pre-roll steps of the steady-state bench: the frame kernel's bytes per launch do not depend on it)
if [ "$WL" = "car" ] || [ "$WL" = "car_fma" ]; then export CRL_CAR_NO_OVERLAP=1 CRL_BENCH_CAR_PREROLL=0; fi
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --workload $WL --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --workload $WL --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
unset CRL_CAR_NO_OVERLAP
if [ "$WL" = "car" ]; then
  # (pre-roll still off)
  # counted f32 / f64 vector FLOP of a CarRacing step (bench.py roofline_valu): wave-instruction counts per class
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_flops -- python3 $REPO/bench.py --workload $WL --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_pmc_flops.json 2> $OUT/pmc_flops.err
fi
unset CRL_BENCH_CAR_PREROLL
cd $REPO
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
tail -25 $OUT/summary.txt
# keep only small files for the merge back
find $OUT -name '*.csv' -size +4M -delete
