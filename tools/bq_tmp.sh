export CRL_GRAY_SWEEP_NB=2 CRL_GRAY_SWEEP_DEBUG=5
bash tools/pmc_sq.sh fused84 sq_sweep 2>&1 | grep -A20 "sweep_kernel" | head -40
