#!/bin/bash
# Round 6 A/B: envs per wavefront of the K = 1 gray launch (CRL_GRAY_EPWV=1: one, as before; default: four).  Variant library:
#   python tools/gray_variant.py ep pong_raster_gray.hip -DCRL_ABLATION
cd ${GRAFT_REPO_ROOT:-.}
export CRL_LIB_VARIANT=ep
run() { python bench.py --workload $1 --steps 100 --warmup 10 --no-cpu-baseline $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['avg_kernel_us'],1))"; }
for rep in 1 2 3; do
  echo "fused84_newest (65 536 x (2,1,84,84) u8): one env per wavefront $(CRL_GRAY_EPWV=1 run fused84_newest) | four $(run fused84_newest)"
done
echo "tournament (42 x 42): one $(CRL_GRAY_EPWV=1 run tournament) | four $(run tournament)"
echo "fused84 (K = 4: unaffected): $(CRL_GRAY_EPWV=1 run fused84) | $(run fused84)"
python -m pytest tests/test_hip_pong_parity.py tests/test_hip_round2.py tests/test_hip_edge_sizes.py tests/test_hip_full_size_sampled.py -m gpu -q -x -k "not car and not config4" 2>&1 | tail -3
