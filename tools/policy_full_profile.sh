#!/bin/bash
# On the GPU box: per-kernel times of the full-size ActorCritic opponent at N envs (default 65 536).  gpurun -- 'bash tools/policy_full_profile.sh'
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/pf_prof; N=${1:-65536}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $REPO/tools/policy_full_time.py $N 20 > $OUT/time.txt 2> $OUT/err < /dev/null
cat $OUT/time.txt
for f in $(find $OUT/t -name '*kernel_stats.csv'); do head -12 "$f"; cp "$f" $OUT/kernel_stats.csv; done
