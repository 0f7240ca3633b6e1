set -x
for wl in car_fma car; do
CAR_WORKLOAD=$wl bash tools/car_timeline.sh > gpurun_out/tl_$wl.txt 2>&1
python3 tools/car_timeline_steps.py > gpurun_out/tl_${wl}_steps.txt 2>&1
python3 tools/car_timeline_summary.py > gpurun_out/tl_${wl}_summ.txt 2>&1
cp gpurun_out/car_timeline/b.json gpurun_out/tl_${wl}_bench.json
rm -rf gpurun_out/car_timeline
done
