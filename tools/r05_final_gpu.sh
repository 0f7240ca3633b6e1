set -x
(time timeout 2400 python -m pytest tests -x -q -m gpu) > gpurun_out/r05_gputest.log 2>&1; echo "rc=$?" >> gpurun_out/r05_gputest.log
tail -4 gpurun_out/r05_gputest.log
