# A/B (round 5): which hardware pipe the walk-ahead's (low-priority, always busy) queue shares -- CRL_CAR_STREAM_ORDER deals the queues round the four pipes
run() { lbl=$1; shift; env "$@" PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 1500 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"; }
for rep in 1 2; do
run "s2oDg   (default: walk beside the bulk)  " X=1
run "s2oDDg  (walk beside side2)              " CRL_CAR_STREAM_ORDER=s2oDDg
run "s2oDDDg (walk beside the high-prio queue)" CRL_CAR_STREAM_ORDER=s2oDDDg
run "s2ogD   (walk on the caller's pipe)      " CRL_CAR_STREAM_ORDER=s2ogD
run "so2Dg   (bulk p1, one p2, side2 p3)      " CRL_CAR_STREAM_ORDER=so2Dg
run "2soDg   (side2 p1, bulk p2, walk beside side2)" CRL_CAR_STREAM_ORDER=2soDg
done
