for v in "X=1" "CRL_CAR_OBS_WHOLE_TILES=1"; do
env $v CRL_LIB_VARIANT=abl python bench.py --workload car --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['avg_kernel_us'], d['roofline']['frac'])"
done
# the frame kernel on its own: a full reset()+render of all envs, no pipeline
python - <<'PY'
import os, time, torch
import competitive_rl_amd as crl
for whole in (0, 1):
    pass
PY
