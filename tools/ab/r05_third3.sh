for rep in 1 2 3 4; do
for wl in car car_fma; do
for v in "X=1" "CRL_CAR_OBS_WHOLE_TILES=1"; do
env $v python bench.py --workload $wl --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl $v', round(d['ms_per_step'],4), round(d['roofline']['avg_kernel_us'],1))"
done
done
done
