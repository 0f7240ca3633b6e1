# the driver's round-end sequence: GPU tests, smoke, bench
(time timeout 3000 python -m pytest tests -x -q -m gpu) > gpurun_out/r05_gputest.log 2>&1; echo "rc=$?" >> gpurun_out/r05_gputest.log
(time python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/r05_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/r05_smoke.log
tail -6 gpurun_out/r05_gputest.log; tail -8 gpurun_out/r05_smoke.log
