#!/bin/bash
# One script, one HEAD: every judged artefact under profiles/ comes from this run (tools/collect_profiles.py copies them).
#   gpurun --timeout 3000 -- 'bash tools/profile_all.sh r06'
RND=${1:-r06}
for wl in raw fused84 fused84_newest fused84_f32 fused84_f32_ref car car_fma tournament protocol; do
  echo "=== $wl ==="
  bash tools/profile_gpu.sh $wl ${RND}_$wl 2>&1 | tail -6
done
# kernel timelines of two steady-state steps, both solver arithmetics (start / end per kernel and queue)
for wl in car car_fma; do
  CAR_WORKLOAD=$wl bash tools/car_timeline.sh > gpurun_out/${RND}_${wl}_timeline_tail.txt 2>&1
  python3 tools/car_timeline_steps.py > gpurun_out/${RND}_${wl}_timeline.txt 2>&1
  python3 tools/car_timeline_summary.py > gpurun_out/${RND}_${wl}_timeline_summary.txt 2>&1
  rm -rf gpurun_out/car_timeline
done
# the default bench line (all configs + CPU baselines), as the driver runs it
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${RND}_bench_all.json 2> gpurun_out/${RND}_bench_all.err
tail -c 600 gpurun_out/${RND}_bench_all.json
# sustained CarRacing rate (the walk-ahead queue must not grow): 6 000 steps at 16 384 envs, both arithmetics
python3 bench.py --workload car --steps 6000 --warmup 5 --no-cpu-baseline > gpurun_out/${RND}_bench_car_long.json 2> gpurun_out/${RND}_bench_car_long.err
python3 bench.py --workload car_fma --steps 6000 --warmup 5 --no-cpu-baseline > gpurun_out/${RND}_bench_car_fma_long.json 2> gpurun_out/${RND}_bench_car_fma_long.err
tail -c 300 gpurun_out/${RND}_bench_car_long.json; tail -c 300 gpurun_out/${RND}_bench_car_fma_long.json
# phase stamps of the touching solve (profiling build), both arithmetics
for sv in box2d fma; do
  CRL_LIB_VARIANT=abl CRL_CAR_STAMPS=1 QUICK_SOLVER=$sv PYTHONPATH=. timeout 100 python3 tools/car_quick.py 16384 1500 500 > gpurun_out/${RND}_car_stamps_$sv.txt 2>&1
done
[ -x tools/solve_chain_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include tools/solve_chain_probe.hip -o tools/solve_chain_probe  # (git-ignored binary: built on first use)
timeout 60 ./tools/solve_chain_probe > gpurun_out/${RND}_solve_chain_probe.txt 2>&1
