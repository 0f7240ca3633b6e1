for wl in car_fma car; do
CAR_WORKLOAD=$wl bash tools/car_timeline.sh > /dev/null 2>&1
python3 tools/car_timeline_steps.py > gpurun_out/tl2_${wl}_steps.txt 2>&1
python3 tools/car_timeline_summary.py > gpurun_out/tl2_${wl}_summ.txt 2>&1
rm -rf gpurun_out/car_timeline
done
