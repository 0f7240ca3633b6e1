for wl in car car_fma; do
CRL_BENCH_DEBUG=1 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep "host enqueue"
CRL_BENCH_DEBUG=1 python bench.py --workload $wl --steps 200 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep "host enqueue"
done
