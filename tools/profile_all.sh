#!/bin/bash
# One script, one HEAD: every judged artefact under profiles/ comes from this run (tools/collect_profiles.py copies them).
#   gpurun --timeout 2400 -- 'bash tools/profile_all.sh r04'
RND=${1:-r04}
for wl in raw fused84 fused84_f32 car tournament; do
  echo "=== $wl ==="
  bash tools/profile_gpu.sh $wl ${RND}_$wl 2>&1 | tail -6
done
# the default bench line (all configs + CPU baselines), as the driver runs it
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${RND}_bench_all.json 2> gpurun_out/${RND}_bench_all.err
tail -c 600 gpurun_out/${RND}_bench_all.json
# sustained CarRacing rate (ADVICE: the walk-ahead queue must not grow): 6 000 steps at 16 384 envs
python3 bench.py --workload car --steps 6000 --warmup 5 --no-cpu-baseline > gpurun_out/${RND}_bench_car_long.json 2> gpurun_out/${RND}_bench_car_long.err
tail -c 400 gpurun_out/${RND}_bench_car_long.json
